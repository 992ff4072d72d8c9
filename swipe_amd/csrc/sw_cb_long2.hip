// Bound builds of the row-shifted kernel for chains of 2 lanes with 49..62 rows per lane (see sw_cb_kernel.inc): two state
// registers per row leave room for them at two waves per SIMD where the exact build stops at 48.
#include "sw_common.cuh"
#include "sw_cb_kernel.inc"

extern "C" hipError_t swa_launch_narrow_bound_long2(int K, const swa_narrow_params* p, int blocks, hipStream_t st)
{
#define SWA_CBK(KK) case KK: return launch_bound<KK, 2>(*p, blocks, st);
  switch (K) {
    SWA_CBK(49) SWA_CBK(50) SWA_CBK(51) SWA_CBK(52) SWA_CBK(53) SWA_CBK(54) SWA_CBK(55) SWA_CBK(56) SWA_CBK(57) SWA_CBK(58) SWA_CBK(59) SWA_CBK(60) SWA_CBK(61) SWA_CBK(62)
    default: return hipErrorInvalidValue;
  }
#undef SWA_CBK
}
