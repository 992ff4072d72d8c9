// Bound build of the two-query kernel on 4-lane chains: pairs of 65..248-row queries, K = 17..62 rows per lane (33 and
// more: 512-thread blocks around the larger profile; see sw_cb_dual_kernel.inc).
#include "sw_common.cuh"
#include "sw_cb_dual_kernel.inc"

extern "C" hipError_t swa_launch_dual_bound_g4(int K, const swa_mp_params* p, int cus, hipStream_t st)
{
#define SWA_DBK(KK) case KK: return launch_dual_bound<KK, 4, (KK > 32 ? 512 : 256)>(*p, cus, st);
  switch (K) {
    SWA_DBK(17) SWA_DBK(18) SWA_DBK(19) SWA_DBK(20) SWA_DBK(21) SWA_DBK(22) SWA_DBK(23) SWA_DBK(24) SWA_DBK(25) SWA_DBK(26) SWA_DBK(27) SWA_DBK(28) SWA_DBK(29) SWA_DBK(30) SWA_DBK(31) SWA_DBK(32)
    SWA_DBK(33) SWA_DBK(34) SWA_DBK(35) SWA_DBK(36) SWA_DBK(37) SWA_DBK(38) SWA_DBK(39) SWA_DBK(40) SWA_DBK(41) SWA_DBK(42) SWA_DBK(43) SWA_DBK(44) SWA_DBK(45) SWA_DBK(46) SWA_DBK(47) SWA_DBK(48)
    SWA_DBK(49) SWA_DBK(50) SWA_DBK(51) SWA_DBK(52) SWA_DBK(53) SWA_DBK(54) SWA_DBK(55) SWA_DBK(56) SWA_DBK(57) SWA_DBK(58) SWA_DBK(59) SWA_DBK(60) SWA_DBK(61) SWA_DBK(62)
    default: return hipErrorInvalidValue;
  }
#undef SWA_DBK
}
