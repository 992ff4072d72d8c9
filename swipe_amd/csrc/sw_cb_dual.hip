// Bound build of the two-query kernel: chains of 16 lanes (queries of 257..512 rows) and of 8 lanes (129..256 rows),
// K = 17..32 rows per lane (kernel and notes: sw_cb_dual_kernel.inc; 8-lane chains of 33..62 rows: sw_cb_dual_long.hip, 4-lane chains: sw_cb_dual_g4.hip).
#include "sw_common.cuh"
#include "sw_cb_dual_kernel.inc"

extern "C" hipError_t swa_launch_dual_bound_long(int K, const swa_mp_params* p, int cus, hipStream_t st);
extern "C" hipError_t swa_launch_dual_bound_g4(int K, const swa_mp_params* p, int cus, hipStream_t st);
extern "C" int swa_dual_bound_available(int G, int K, int nres)
{
  if (nres != 32 || K < 17) return 0;
  return G == 16 ? K <= 32 : G == 8 || G == 4 ? K <= 62 : 0;
}
extern "C" hipError_t swa_launch_dual_bound(int G, int K, const swa_mp_params* p, int cus, hipStream_t st)
{
#define SWA_DBK(KK) case KK: return G == 16 ? launch_dual_bound<KK, 16>(*p, cus, st) : launch_dual_bound<KK, 8>(*p, cus, st);
  if (G == 4) return swa_launch_dual_bound_g4(K, p, cus, st);
  if (G != 16 && G != 8) return hipErrorInvalidValue;
  if (G == 8 && K > 32) return swa_launch_dual_bound_long(K, p, cus, st);
  switch (K) {
    SWA_DBK(17) SWA_DBK(18) SWA_DBK(19) SWA_DBK(20) SWA_DBK(21) SWA_DBK(22) SWA_DBK(23) SWA_DBK(24)
    SWA_DBK(25) SWA_DBK(26) SWA_DBK(27) SWA_DBK(28) SWA_DBK(29) SWA_DBK(30) SWA_DBK(31) SWA_DBK(32)
    default: return hipErrorInvalidValue;
  }
#undef SWA_DBK
}
