// Bound build of the two-query kernel (swa_dual_kernel, sw_mp_kernel.inc) for the non-nucleotide alphabets: the
// column-biased recurrence of sw_cb_kernel.inc - every value of step u of a 16-step period stored + u R, so that the
// horizontal gap state needs no "- R" - on one database sequence per chain against TWO queries in the two f16 halves:
// 5 instead of 6.5 instructions per cell pair (no row maxima either: the bound is read off E at each period's end).  As there, the result is an upper bound at most 15 R above the score
// and only serves searches with a score threshold (swa_search2_topk, swa_search_frames_topk: pairs of equally long
// query frames of a translated search); sequences whose bound reaches the threshold are recomputed exactly.
// Nucleotide searches have thresholds of 20..30 at R = 2 - inside the slack - and keep the exact kernel.
#include "sw_common.cuh"

#define SWA_DB_PERIOD 16

template <int K, int W, int G, int N>
__global__ void __launch_bounds__(256, W)
swa_dual_bound_kernel(swa_mp_params p)
{
  static_assert(N % G == 0 && N % 2 == 0 && K + N + 2 <= 68, "period must be whole blocks of G steps");
  constexpr int NRES = 32;
  constexpr int C = (K + 3) / 4;
  constexpr u32 CS = C * 256;
  constexpr int NB = G == 16 ? 1 : 8 / G;                 // batches per wave
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  signal_block_started(p.done);
  {
    u32* t = (u32*)lds;
    const int total = (NRES + 1) * C * 16 * 4;
    const float add = 2.0f * p.gapextend_f;               // the diagonal crosses one row and one column
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
      const int k = e & 3, ll = (e >> 2) & (G - 1), c = (e >> 6) % C, d = (e >> 6) / C;
      const int local = c * 4 + k, row = ll * K + local;
      const bool live = local < K && row < p.qlen && d < NRES;
      const float v1 = live && row < (p.qlen_a ? p.qlen_a : p.qlen) ? (float)p.matrix[(d << 5) + p.qseq[row]] : -1.0f;
      const float v2 = live && row < (p.qlen_b ? p.qlen_b : p.qlen) ? (float)p.matrix[(d << 5) + p.qseq2[row]] : -1.0f;
      t[e] = float_to_half_bits(v1 + add) | (float_to_half_bits(v2 + add) << 16);
    }
  }
  __syncthreads();

  const int lane = threadIdx.x & 63, lg = lane & (G - 1), q = lane / G;
  const int half = G == 16 ? 0 : (q & 1), row = G == 16 ? q : ((q >> 1) & 3), bsel = G == 16 ? 0 : (q >> 3);
  const u32 l16 = (u32)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds + (u32)(lane & 15) * 16;
  const bool inner_first = G < 16 && lg == 0 && (lane & 15) != 0;
  const bool chain_last = G < 16 && lg == G - 1;
  const h2 negQR = as_h2(p.negQR);
  const h2 zero = {0, 0};
  const h2 one = {(_Float16)1.0f, (_Float16)1.0f};
  const h2 send_mul = chain_last ? zero : one;
  const h2 send_const = as_h2(p.negKR) + as_h2(p.rowc[1]);          // -K R + R
  const h2 negNR = zero - as_h2(p.rowc[N]);
  const u32 PADOFF = NRES * CS;

  for (;;) {
    int w = 0;
    if (lane == 0) w = atomicAdd(p.counter, 1);
    w = __builtin_amdgcn_readfirstlane(w);
    const int b0 = NB * w;
    if (b0 >= p.nbatches) break;
    int steps = 0;
#pragma unroll
    for (int i = 0; i < NB; ++i)
      if (b0 + i < p.nbatches) { const int st = p.batches[b0 + i].steps; steps = st > steps ? st : steps; }
    const int b = b0 + bsel;
    swa_batch bd;
    bd.offset = 0;
    bd.steps = 0;
    if (b < p.nbatches) bd = p.batches[b];
    const uint16_t* s = p.stream + (int64_t)bd.offset * 64 + row * 16 + lg;
    const int mychunks = (bd.steps + 15) >> 4;
    const int total = steps + G;
    const int shift = 8 * half;

    h2 H[K], E[K];
#pragma unroll
    for (int r = 0; r < K; ++r) { H[r] = as_h2(p.rowc[r]); E[r] = as_h2(p.rowc[r + 1]); }
    h2 diag = zero - as_h2(p.rowc[1]), hsend = zero, fsend = zero;
    h2 S = zero;
    u32 cur = PADOFF;
    u32 raw = mychunks > 0 ? (((u32)s[0] >> shift) & 0xFF) : (u32)SWA_PAD;

#define SWA_DB_STEP(U)                                                                         \
    {                                                                                          \
      const u32 pl2 = (u32)__builtin_amdgcn_update_dpp(0, (int)pl, DPP_ROW_SHL1, 0xF, 0xF, true); \
      const u32 shifted = row_shr1(cur, pl);                                                   \
      cur = inner_first ? pl : shifted;                                                        \
      pl = pl2;                                                                                \
      const h2 hup = as_h2((u32)__builtin_amdgcn_update_dpp((int)p.rowc[U], (int)as_u32(hsend), DPP_ROW_SHR1, 0xF, 0xF, false)); \
      h2 F = as_h2((u32)__builtin_amdgcn_update_dpp(0, (int)as_u32(fsend), DPP_ROW_SHR1, 0xF, 0xF, true)); \
      const h2 hd = diag;                                                                      \
      diag = hup;                                                                              \
      const u32 aoff = cur + l16;                                                              \
      u4v nx = *(lds_u4_ptr)(uintptr_t)(aoff);                                                 \
      h2 a = hd + as_h2(nx.x);                                                                 \
      _Pragma("unroll") for (int c = 0; c < C; ++c) {                                          \
        const u4v pa = nx;                                                                     \
        if (c + 1 < C) nx = *(lds_u4_ptr)(uintptr_t)(aoff + (c + 1) * 256);                    \
        _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                        \
          const int r = c * 4 + k;                                                             \
          if (r < K) {                                                                         \
            const h2 h = pk_max3(a, E[r], F);                                                  \
            const u32 w1 = k == 0 ? pa.y : k == 1 ? pa.z : k == 2 ? pa.w : nx.x;               \
            const h2 t = h + negQR;                                                            \
            const h2 hold = H[r];                                                              \
            if (r + 1 < K) a = hold + as_h2(w1);                                               \
            H[r] = h;                                                                          \
            F = pk_max(F, t);                                                                  \
            E[r] = pk_max3(E[r], t, as_h2(p.rowc[r + 2 + (U)]));                               \
          }                                                                                    \
        }                                                                                      \
      }                                                                                        \
      const h2 sadd = chain_last ? as_h2(p.rowc[(U) + 1]) : send_const;                        \
      hsend = __builtin_elementwise_fma(H[K - 1], send_mul, sadd);                             \
      fsend = __builtin_elementwise_fma(F, send_mul, sadd);                                    \
    }
#define SWA_DB_PAIR(U) { if ((U) - blk * G < n) { SWA_DB_STEP(U) SWA_DB_STEP((U) + 1) } }

    for (int base = 0; base < total; base += N) {
#pragma unroll
      for (int blk = 0; blk < N / G; ++blk) {
        const int m = base / G + blk;
        if (m * G < total) {
          u32 pl = (raw < (u32)NRES ? raw : (u32)NRES) * CS;
          const int col = (m + 1) * G;
          raw = ((col >> 4) < mychunks) ? (((u32)s[(int64_t)(col >> 4) * 64 + (col & 15)] >> shift) & 0xFF) : (u32)SWA_PAD;
          const int n = total - m * G < G ? total - m * G : G;
          if constexpr (G >= 2) SWA_DB_PAIR(blk * G + 0)
          if constexpr (G >= 4) SWA_DB_PAIR(blk * G + 2)
          if constexpr (G >= 8) { SWA_DB_PAIR(blk * G + 4) SWA_DB_PAIR(blk * G + 6) }
          if constexpr (G >= 16) { SWA_DB_PAIR(blk * G + 8) SWA_DB_PAIR(blk * G + 10) SWA_DB_PAIR(blk * G + 12) SWA_DB_PAIR(blk * G + 14) }
        }
      }
      // end of the period: bring the state back by N R; the row maxima carry on and are folded after the last step
#pragma unroll
      for (int r = 0; r < K; r += 2) {
        if (r + 1 < K) S = pk_max3(S, E[r] - as_h2(p.rowc[r + 1]), E[r + 1] - as_h2(p.rowc[r + 2]));
        else S = pk_max(S, E[r] - as_h2(p.rowc[r + 1]));
      }
#pragma unroll
      for (int r = 0; r < K; ++r) { H[r] = H[r] + negNR; E[r] = E[r] + negNR; }
      diag = diag + negNR;
      hsend = hsend + negNR;
      fsend = fsend + negNR;
    }
    S = S - negQR;                                      // the bound was kept on H - Q (see above)
#undef SWA_DB_PAIR
#undef SWA_DB_STEP

    S = pk_max(S, as_h2((u32)__builtin_amdgcn_update_dpp(0, (int)as_u32(S), DPP_ROW_SHR(1), 0xF, 0xF, true)));
    if (G >= 4) S = pk_max(S, as_h2((u32)__builtin_amdgcn_update_dpp(0, (int)as_u32(S), DPP_ROW_SHR(2), 0xF, 0xF, true)));
    if (G >= 8) S = pk_max(S, as_h2((u32)__builtin_amdgcn_update_dpp(0, (int)as_u32(S), DPP_ROW_SHR(4), 0xF, 0xF, true)));
    if (G == 16) S = pk_max(S, as_h2((u32)__builtin_amdgcn_update_dpp(0, (int)as_u32(S), DPP_ROW_SHR(8), 0xF, 0xF, true)));
    int id = -1, s1 = -1, s2 = -1;
    if (lg == G - 1 && b < p.nbatches) {
      id = p.slots[(int64_t)b * SWA_SLOTS + row * 2 + half];
      s1 = (int)(float)S.x; s2 = (int)(float)S.y;
      if (id >= 0) { p.scores[id] = s1; p.scores2[id] = s2; }
    }
    const bool o1 = id >= 0 && s1 >= p.limit, o2 = id >= 0 && s2 >= p.limit;
    const u64 m1 = __ballot(o1), m2 = __ballot(o2);
    const int n1 = __popcll(m1), n2 = __popcll(m2);
    const u64 below = (1ull << lane) - 1;
    if ((n1 | n2) && p.done) __threadfence();            // a follower may pick an entry up at once: the scores first
    if (n1) {
      int base = 0;
      if (lane == 0) base = atomicAdd(p.ovf_count, n1);
      base = __builtin_amdgcn_readfirstlane(base);
      if (o1) p.ovf_list[base + __popcll(m1 & below)] = id;
    }
    if (n2) {
      int base = 0;
      if (lane == 0) base = atomicAdd(p.ovf_count2, n2);
      base = __builtin_amdgcn_readfirstlane(base);
      if (o2) p.ovf_list2[base + __popcll(m2 & below)] = id;
    }
  }
  signal_block_done(p.finished, p.done);
}

template <int K, int G>
static hipError_t launch_dual_bound(const swa_mp_params& p, int cus, hipStream_t st)
{
  constexpr int W = K <= 24 ? 3 : 2;                      // LDS: 33 x ceil(K / 4) x 256 B per block
  const size_t lds = (size_t)33 * ((K + 3) / 4) * 256;
  auto kern = swa_dual_bound_kernel<K, W, G, SWA_DB_PERIOD>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  const int per_wave = G == 16 ? 1 : 8 / G;
  const int items = (p.nbatches + per_wave - 1) / per_wave;
  int blocks = (items + 3) / 4;
  if (blocks > cus * W) blocks = cus * W;
  hipLaunchKernelGGL(kern, dim3(blocks < 1 ? 1 : blocks), dim3(256), lds, st, p);
  return hipGetLastError();
}
// builds: chains of 16 lanes (queries of 257..512 rows) and of 8 lanes (129..256 rows), K = 17..32 rows per lane
extern "C" int swa_dual_bound_available(int G, int K, int nres) { return nres == 32 && (G == 16 || G == 8) && K >= 17 && K <= 32; }
extern "C" hipError_t swa_launch_dual_bound(int G, int K, const swa_mp_params* p, int cus, hipStream_t st)
{
#define SWA_DBK(KK) case KK: return G == 16 ? launch_dual_bound<KK, 16>(*p, cus, st) : launch_dual_bound<KK, 8>(*p, cus, st);
  if (G != 16 && G != 8) return hipErrorInvalidValue;
  switch (K) {
    SWA_DBK(17) SWA_DBK(18) SWA_DBK(19) SWA_DBK(20) SWA_DBK(21) SWA_DBK(22) SWA_DBK(23) SWA_DBK(24)
    SWA_DBK(25) SWA_DBK(26) SWA_DBK(27) SWA_DBK(28) SWA_DBK(29) SWA_DBK(30) SWA_DBK(31) SWA_DBK(32)
    default: return hipErrorInvalidValue;
  }
#undef SWA_DBK
}
