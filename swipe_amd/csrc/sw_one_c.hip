// Bound build, one lane per sequence pair, K = 1..24 rows (see sw_one_kernel.inc).
#include "sw_common.cuh"
#include "sw_profile.cuh"
#include "sw_one_kernel.inc"

extern "C" hipError_t swa_launch_one_bound_c(int K, const swa_narrow_params* p, int blocks, hipStream_t st)
{
#define SWA_ONEB(KK) case KK: return launch_one_bound<KK>(*p, blocks, st);
  switch (K) {
    SWA_ONEB(1) SWA_ONEB(2) SWA_ONEB(3) SWA_ONEB(4) SWA_ONEB(5) SWA_ONEB(6) SWA_ONEB(7) SWA_ONEB(8) SWA_ONEB(9) SWA_ONEB(10) SWA_ONEB(11) SWA_ONEB(12) SWA_ONEB(13) SWA_ONEB(14) SWA_ONEB(15) SWA_ONEB(16) SWA_ONEB(17) SWA_ONEB(18) SWA_ONEB(19) SWA_ONEB(20) SWA_ONEB(21) SWA_ONEB(22) SWA_ONEB(23) SWA_ONEB(24)
    default: return hipErrorInvalidValue;
  }
#undef SWA_ONEB
}
