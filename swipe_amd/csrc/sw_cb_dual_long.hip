// Bound build of the two-query kernel on 8-lane chains of 33..62 rows per lane (pairs of 257..496-row queries): half the
// hand-overs and half the skew of 16 lanes x 17..31 rows; 512-thread blocks, one per CU, around the 101..135 KB profile.
#include "sw_common.cuh"
#include "sw_cb_dual_kernel.inc"

extern "C" hipError_t swa_launch_dual_bound_long(int K, const swa_mp_params* p, int cus, hipStream_t st)
{
#define SWA_DBK(KK) case KK: return launch_dual_bound<KK, 8, 512>(*p, cus, st);
  switch (K) {
    SWA_DBK(33) SWA_DBK(34) SWA_DBK(35) SWA_DBK(36) SWA_DBK(37) SWA_DBK(38) SWA_DBK(39) SWA_DBK(40) SWA_DBK(41) SWA_DBK(42) SWA_DBK(43) SWA_DBK(44) SWA_DBK(45) SWA_DBK(46) SWA_DBK(47) SWA_DBK(48)
    SWA_DBK(49) SWA_DBK(50) SWA_DBK(51) SWA_DBK(52) SWA_DBK(53) SWA_DBK(54) SWA_DBK(55) SWA_DBK(56) SWA_DBK(57) SWA_DBK(58) SWA_DBK(59) SWA_DBK(60) SWA_DBK(61) SWA_DBK(62)
    default: return hipErrorInvalidValue;
  }
#undef SWA_DBK
}
