// Bound build, one lane per sequence pair, K = 25..48 rows (see sw_one_kernel.inc).
#include "sw_common.cuh"
#include "sw_profile.cuh"
#include "sw_one_kernel.inc"

extern "C" hipError_t swa_launch_one_bound_d(int K, const swa_narrow_params* p, int blocks, hipStream_t st)
{
#define SWA_ONEB(KK) case KK: return launch_one_bound<KK>(*p, blocks, st);
  switch (K) {
    SWA_ONEB(25) SWA_ONEB(26) SWA_ONEB(27) SWA_ONEB(28) SWA_ONEB(29) SWA_ONEB(30) SWA_ONEB(31) SWA_ONEB(32) SWA_ONEB(33) SWA_ONEB(34) SWA_ONEB(35) SWA_ONEB(36) SWA_ONEB(37) SWA_ONEB(38) SWA_ONEB(39) SWA_ONEB(40) SWA_ONEB(41) SWA_ONEB(42) SWA_ONEB(43) SWA_ONEB(44) SWA_ONEB(45) SWA_ONEB(46) SWA_ONEB(47) SWA_ONEB(48)
    default: return hipErrorInvalidValue;
  }
#undef SWA_ONEB
}
