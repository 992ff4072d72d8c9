// Bound builds of the row-shifted kernel for chains of 2 lanes (see sw_cb_kernel.inc).
#include "sw_common.cuh"
#include "sw_cb_kernel.inc"

extern "C" hipError_t swa_launch_narrow_bound_g2(int K, const swa_narrow_params* p, int blocks, hipStream_t st)
{
#define SWA_CBK(KK) case KK: return launch_bound<KK, 2>(*p, blocks, st);
  switch (K) {
    SWA_CBK(5) SWA_CBK(6) SWA_CBK(7) SWA_CBK(8) SWA_CBK(9) SWA_CBK(10) SWA_CBK(11) SWA_CBK(12) SWA_CBK(13) SWA_CBK(14) SWA_CBK(15) SWA_CBK(16) SWA_CBK(17) SWA_CBK(18) SWA_CBK(19) SWA_CBK(20) SWA_CBK(21) SWA_CBK(22) SWA_CBK(23) SWA_CBK(24) SWA_CBK(25) SWA_CBK(26) SWA_CBK(27) SWA_CBK(28) SWA_CBK(29) SWA_CBK(30) SWA_CBK(31) SWA_CBK(32) SWA_CBK(33) SWA_CBK(34) SWA_CBK(35) SWA_CBK(36) SWA_CBK(37) SWA_CBK(38) SWA_CBK(39) SWA_CBK(40) SWA_CBK(41) SWA_CBK(42) SWA_CBK(43) SWA_CBK(44) SWA_CBK(45) SWA_CBK(46) SWA_CBK(47) SWA_CBK(48)
    default: return hipErrorInvalidValue;
  }
#undef SWA_CBK
}
