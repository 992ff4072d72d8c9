// Single-pass two-query kernel, nucleotide alphabet, 4-lane chains of 33..60 rows per lane: both strands of a query of
// 129..240 rows on a quarter of the lanes (the 17 x 16 x 256 B table and 3 K + 30 registers are those of the 16-lane build of 63 rows).
#include "sw_common.cuh"
#define SWA_MP_TEMPLATES_ONLY
#include "sw_mp_kernel.inc"

extern "C" hipError_t swa_launch_dual_long4(int K, const swa_mp_params* p, int cus, hipStream_t st)
{
#define SWA_DL(KK) case KK: return launch_dual<KK, 16, 4>(*p, cus, st);
  switch (K) {
    SWA_DL(33) SWA_DL(34) SWA_DL(35) SWA_DL(36) SWA_DL(37) SWA_DL(38) SWA_DL(39) SWA_DL(40) SWA_DL(41) SWA_DL(42) SWA_DL(43) SWA_DL(44) SWA_DL(45) SWA_DL(46) SWA_DL(47) SWA_DL(48) SWA_DL(49) SWA_DL(50) SWA_DL(51) SWA_DL(52) SWA_DL(53) SWA_DL(54) SWA_DL(55) SWA_DL(56) SWA_DL(57) SWA_DL(58) SWA_DL(59) SWA_DL(60)
    default: return hipErrorInvalidValue;
  }
#undef SWA_DL
}
