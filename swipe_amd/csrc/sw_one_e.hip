// Bound build, one lane per sequence pair, K = 49..60 rows (see sw_one_kernel.inc): 2 K state registers leave room
// for twelve more rows than the exact build has at two waves per SIMD; one column at a time.
#include "sw_common.cuh"
#include "sw_profile.cuh"
#include "sw_one_kernel.inc"

template <int K>
static hipError_t launch_one_bound_long(const swa_narrow_params& p, int blocks, hipStream_t st)
{
  const size_t lds = (size_t)32 * ((K + 7) / 8) * 256;
  auto kern = swa_one_bound_kernel<K, 2, false>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, st, p);
  return hipGetLastError();
}

extern "C" hipError_t swa_launch_one_bound_e(int K, const swa_narrow_params* p, int blocks, hipStream_t st)
{
#define SWA_ONEB(KK) case KK: return launch_one_bound_long<KK>(*p, blocks, st);
  switch (K) {
    SWA_ONEB(49) SWA_ONEB(50) SWA_ONEB(51) SWA_ONEB(52) SWA_ONEB(53) SWA_ONEB(54) SWA_ONEB(55) SWA_ONEB(56) SWA_ONEB(57) SWA_ONEB(58) SWA_ONEB(59) SWA_ONEB(60)
    default: return hipErrorInvalidValue;
  }
#undef SWA_ONEB
}
