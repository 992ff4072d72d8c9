// The LDS profile of the row-shifted kernels (swa_narrow_split_kernel, swa_narrow_one_kernel): see sw_kernels.hip.
#ifndef SW_PROFILE_CUH
#define SW_PROFILE_CUH
#include "sw_common.cuh"
template <int K, int G>
__device__ __forceinline__ void build_profile_f16_split(unsigned char* lds, const swa_query* q, float add, int row0 = 0)
{
  constexpr int C = (K + 7) / 8;
  unsigned short* t = (unsigned short*)lds;
  const int total = 32 * C * 16 * 8;
  for (int e = threadIdx.x; e < total; e += blockDim.x) {
    const int k = e & 7, l = (e >> 3) & (G - 1), c = (e >> 7) % C, d = (e >> 7) / C;
    const int local = c * 8 + k;
    const int row = row0 + l * K + local;
    float v = -1.0f;
    if (local < K && row < q->qlen && d != SWA_PAD) v = (float)q->matrix[(d << 5) + q->qseq[row]];
    t[e] = (unsigned short)float_to_half_bits(v + add);
  }
}
#endif
