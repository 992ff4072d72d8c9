// One lane per sequence pair, K = 1..24 rows (see sw_one_kernel.inc).
#include "sw_common.cuh"
#include "sw_profile.cuh"
#include "sw_one_kernel.inc"

extern "C" hipError_t swa_launch_narrow_one_a(int K, const swa_narrow_params* p, int blocks, hipStream_t st)
{
#define SWA_ONE(KK) case KK: return launch_one<KK>(*p, blocks, st);
  switch (K) {
    SWA_ONE(1) SWA_ONE(2) SWA_ONE(3) SWA_ONE(4) SWA_ONE(5) SWA_ONE(6) SWA_ONE(7) SWA_ONE(8) SWA_ONE(9) SWA_ONE(10) SWA_ONE(11) SWA_ONE(12) SWA_ONE(13) SWA_ONE(14) SWA_ONE(15) SWA_ONE(16) SWA_ONE(17) SWA_ONE(18) SWA_ONE(19) SWA_ONE(20) SWA_ONE(21) SWA_ONE(22) SWA_ONE(23) SWA_ONE(24)
    default: return hipErrorInvalidValue;
  }
#undef SWA_ONE
}
