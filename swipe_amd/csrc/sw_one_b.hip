// One lane per sequence pair, K = 25..48 rows (see sw_one_kernel.inc).
#include "sw_common.cuh"
#include "sw_profile.cuh"
#include "sw_one_kernel.inc"

extern "C" hipError_t swa_launch_narrow_one_b(int K, const swa_narrow_params* p, int blocks, hipStream_t st)
{
#define SWA_ONE(KK) case KK: return launch_one<KK>(*p, blocks, st);
  switch (K) {
    SWA_ONE(25) SWA_ONE(26) SWA_ONE(27) SWA_ONE(28) SWA_ONE(29) SWA_ONE(30) SWA_ONE(31) SWA_ONE(32) SWA_ONE(33) SWA_ONE(34) SWA_ONE(35) SWA_ONE(36) SWA_ONE(37) SWA_ONE(38) SWA_ONE(39) SWA_ONE(40) SWA_ONE(41) SWA_ONE(42) SWA_ONE(43) SWA_ONE(44) SWA_ONE(45) SWA_ONE(46) SWA_ONE(47) SWA_ONE(48)
    default: return hipErrorInvalidValue;
  }
#undef SWA_ONE
}
