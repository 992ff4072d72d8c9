// swipe_amd device code, the small kernels around the search: residue-stream formatting, the pipelined open's entry
// stripping, six-frame translation, the hit filter, the alignment phase's end points and the wave-per-sequence re-queue.
// A translation unit of its own (round 4): HIP loads a code object when its first kernel is launched, and an open must not
// wait for the 17 MB of first-pass builds in sw_kernels.hip before its first format kernel can run.
#include "sw_common.cuh"

// ------------------------------------------------------------------ stream formatting
// Builds the batch-interleaved residue stream from the raw database.  One thread per
// (batch, chunk, lane): writes the u16 (residue of slot A | residue of slot B << 8) that the
// quarter-wave `grp` consumes at step 16*chunk + l.  Reads of a sequence are contiguous over
// l, writes are fully coalesced.
extern "C" __global__ void __launch_bounds__(256)
swa_format_stream(swa_seqs sq, const int32_t* __restrict__ slots, const swa_batch* __restrict__ batches,
                  int nbatches, uint16_t* __restrict__ stream)
{
  const int b = blockIdx.x;
  if (b >= nbatches) return;
  const swa_batch bd = batches[b];
  const int32_t* sl = slots + (int64_t)b * SWA_SLOTS;
  uint16_t* out = stream + (int64_t)bd.offset * 64;
  const int total = ((bd.steps + 15) >> 4) * 64;
  for (int e = threadIdx.x; e < total; e += blockDim.x) {
    const int chunk = e >> 6, lane = e & 63, grp = lane >> 4, l = lane & 15;
    const int64_t t = (int64_t)chunk * 16 + l;
    u32 v = 0;
    for (int h = 0; h < 2; ++h) {
      const int32_t s = sl[grp * 2 + h];
      u32 r = SWA_PAD;
      if (s >= 0) {
        int64_t o, n;
        seq_span(sq, s, o, n);
        if (t < n) r = seq_residue(sq, o + t);
      }
      v |= r << (8 * h);
    }
    out[e] = (uint16_t)v;
  }
}

// The one-sequence-per-row stream of a NUCLEOTIDE shard at 4 bits per base: [batch][16-column chunk][row 0..3][column
// 0..15] nibbles = 32 bytes per chunk instead of 128 (the slot-B half of the u16 form is padding there).  Code 0 pads:
// the reference's nucleotide matrices score it -1 against everything (matrices.cc:531-538), like SWA_PAD.
extern "C" __global__ void __launch_bounds__(256)
swa_format_stream4(swa_seqs sq, const int32_t* __restrict__ slots, const swa_batch* __restrict__ batches,
                   int nbatches, uint8_t* __restrict__ stream)
{
  const int b = blockIdx.x;
  if (b >= nbatches) return;
  const swa_batch bd = batches[b];
  const int32_t* sl = slots + (int64_t)b * SWA_SLOTS;
  uint8_t* out = stream + (int64_t)bd.offset * 32;
  const int total = ((bd.steps + 15) >> 4) * 32;
  for (int e = threadIdx.x; e < total; e += blockDim.x) {
    const int chunk = e >> 5, byte = e & 31, grp = byte >> 3, l = (byte & 7) * 2;
    const int32_t s = sl[grp * 2];
    u32 v = 0;
    if (s >= 0) {
      int64_t o, n;
      seq_span(sq, s, o, n);
      const int64_t t = (int64_t)chunk * 16 + l;
      if (t < n) v = seq_residue(sq, o + t) & 15u;
      if (t + 1 < n) v |= (seq_residue(sq, o + t + 1) & 15u) << 4;
    }
    out[e] = (uint8_t)v;
  }
}

// Six-frame translation of a nucleotide shard into the protein residues the DP kernels consume - the
// pre-pass for translated-database searches (-p 3 / -p 4).  db_translate (database.cc:1182-1218): virtual
// sequence v = 6*s + 3*strand + frame holds (len_s - frame) / 3 residues, table[256a + 16b + c] over the
// three IUPAC nibbles of a codon, strand 1 reading the reverse complement.  One thread per output residue,
// consecutive lanes on consecutive residues, so writes are coalesced and a wave reads 192 contiguous
// nucleotide bytes per load; a chromosome-sized sequence and a thousand short reads load the GPU alike.
// A block takes 4096 consecutive output residues: one thread locates the first virtual sequence by binary
// search in the global offsets, the block stages the next 2048 offsets in LDS and every output finds its
// sequence by an 11-step search there (a tile crossing more than 2048 sequence boundaries - average length
// below 2 - falls back to the global search).  HBM-bound: 1 B read + 2 B written per base.
#define SWA_TR_TILE 4096
#define SWA_TR_WIN 2048
extern "C" __global__ void __launch_bounds__(256)
swa_translate_frames(const uint8_t* __restrict__ nt, int packed, const int64_t* __restrict__ ntoff,
                     const int64_t* __restrict__ voff, int64_t nv, const uint8_t* __restrict__ table,
                     uint8_t* __restrict__ prot, int64_t total)
{
  // packed: the nucleotide shard's own 4-bit form, two bases per byte, low nibble first (what the pipelined open leaves on the
  // device); else one base per byte
  auto base = [&](int64_t i) -> u32 { return packed ? ((u32)nt[i >> 1] >> ((int)(i & 1) * 4)) & 15u : (u32)nt[i] & 15u; };
  __shared__ uint8_t tab[4096];
  __shared__ int64_t win[SWA_TR_WIN + 1];
  __shared__ int64_t s_v0;
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) tab[i] = table[i];
  for (int64_t r0 = (int64_t)blockIdx.x * SWA_TR_TILE; r0 < total; r0 += (int64_t)gridDim.x * SWA_TR_TILE) {
    const int64_t r1 = r0 + SWA_TR_TILE < total ? r0 + SWA_TR_TILE : total;
    if (threadIdx.x == 0) {
      int64_t lo = 0, hi = nv;                            // largest v with voff[v] <= r0
      while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (voff[mid] <= r0) lo = mid; else hi = mid;
      }
      s_v0 = lo;
    }
    __syncthreads();
    const int64_t v0 = s_v0;
    for (int i = threadIdx.x; i <= SWA_TR_WIN; i += blockDim.x)
      win[i] = v0 + i <= nv ? voff[v0 + i] : INT64_MAX;
    __syncthreads();
    const bool covered = win[SWA_TR_WIN] >= r1;           // every boundary of the tile is in the window
    for (int64_t r = r0 + threadIdx.x; r < r1; r += blockDim.x) {
      int64_t v, vo;
      if (covered) {
        int lo = 0, hi = SWA_TR_WIN;                      // largest i with win[i] <= r
        while (hi - lo > 1) {
          const int mid = (lo + hi) >> 1;
          if (win[mid] <= r) lo = mid; else hi = mid;
        }
        v = v0 + lo;
        vo = win[lo];
      } else {
        int64_t lo = v0, hi = nv;
        while (hi - lo > 1) {
          const int64_t mid = (lo + hi) >> 1;
          if (voff[mid] <= r) lo = mid; else hi = mid;
        }
        v = lo;
        vo = voff[lo];
      }
      const int64_t sq = v / 6, k = r - vo;
      const int t = (int)(v - 6 * sq), f = t % 3;
      const int64_t o = ntoff[sq], len = ntoff[sq + 1] - o;
      u32 a, b, c;
      if (t < 3) {
        const int64_t p = o + f + 3 * k;
        a = base(p); b = base(p + 1); c = base(p + 2);
      } else {                                            // complement of a nibble = its 4 bits reversed
        const int64_t p = o + len - 1 - f - 3 * k;
        a = __brev(base(p)) >> 28; b = __brev(base(p - 1)) >> 28; c = __brev(base(p - 2)) >> 28;
      }
      prot[r] = tab[256 * a + 16 * b + c];
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------ hit filter
// The hits_enter acceptance test (hits.cc:174-184) over all scores of the shard: counts
// totalhits / obvious and compacts candidates (index, score) for the host-side top-K.
extern "C" __global__ void __launch_bounds__(256)
swa_filter_hits(const int* __restrict__ scores, const long long* __restrict__ scores64, int n, int which,
                long long minscore, long long maxscore, int* __restrict__ cand_count,
                int cand_cap, swa_cand* __restrict__ cand, unsigned long long* __restrict__ tallies)
{
  const int lane = threadIdx.x & 63;
  unsigned long long total = 0, obvious = 0;
  for (int base = blockIdx.x * blockDim.x + (int)(threadIdx.x & ~63u); base < n; base += gridDim.x * blockDim.x) {
    const int i = base + lane;
    long long sc = -1;
    const bool valid = i < n;
    if (valid) { sc = scores[i]; if (sc == SWA_SCORE_IN_64) sc = scores64[i]; }
    const bool obv = valid && sc > maxscore;
    const bool tot = valid && sc >= minscore;
    const bool keep = tot && !obv;
    obvious += obv;
    total += tot;
    const u64 mk = __ballot(keep);
    const int nk = __popcll(mk);
    if (nk) {
      int base = 0;
      if (lane == 0) base = atomicAdd(cand_count, nk);
      base = __builtin_amdgcn_readfirstlane(base);
      const int pos = base + __popcll(mk & ((1ull << lane) - 1));
      if (keep && pos < cand_cap) cand[pos] = swa_cand{sc, i, which};
    }
  }
  for (int sh = 32; sh > 0; sh >>= 1) { total += __shfl_down(total, sh); obvious += __shfl_down(obvious, sh); }
  if (lane == 0) { if (total) atomicAdd(&tallies[0], total); if (obvious) atomicAdd(&tallies[1], obvious); }
}

// ------------------------------------------------------------------ alignment end points
// search16s (search16s.cc:297-548) for the few sequences of the alignment phase: exact score, the
// 0-based column where the final maximum is first reached and the smallest row holding it there.
// One thread per sequence, H/E columns in global scratch ([row][thread], coalesced); at most a few
// hundred sequences per query, so throughput is irrelevant here.
extern "C" __global__ void __launch_bounds__(64)
swa_endpoints_kernel(swa_seqs sq, const int32_t* __restrict__ ids, const uint8_t* __restrict__ minus, int n,
                     const uint8_t* __restrict__ qseq, int qlen,
                     const int32_t* __restrict__ matrix, long long Q, long long R,
                     long long* __restrict__ Hs, long long* __restrict__ Es,
                     long long* __restrict__ out_score, long long* __restrict__ out_pos, long long* __restrict__ out_q)
{
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const int stride = gridDim.x * blockDim.x;
  int64_t o, len;
  seq_span(sq, ids[t], o, len);
  // minus[t]: the reverse complement of a nucleotide sequence, as db_getsequence hands it out for
  // strand 1 (database.cc:1327-1339); complementing a one-hot/IUPAC nibble = reversing its 4 bits
  const bool rc = minus && minus[t];
  for (int i = 0; i < qlen; ++i) { Hs[(int64_t)i * stride + t] = 0; Es[(int64_t)i * stride + t] = 0; }
  long long S = 0, bp = 0, bq = -1;                       // d_best = d_begin, q_best = -1 (search16s.cc:483-486)
  for (int64_t j = 0; j < len; ++j) {
    const int sym = rc ? (int)(__brev(seq_residue(sq, o + len - 1 - j)) >> 28) : (int)seq_residue(sq, o + j);
    const int32_t* row = matrix + (sym << 5);
    long long hd = 0, f = 0, cm = 0, cq = -1;
    for (int i = 0; i < qlen; ++i) {
      const int64_t a = (int64_t)i * stride + t;
      const long long n0 = Hs[a];
      long long e = Es[a];
      long long h = hd + row[qseq[i]];
      if (f > h) h = f;
      if (e > h) h = e;
      if (h < 0) h = 0;
      if (h > cm) { cm = h; cq = i; }
      Hs[a] = h;
      const long long tt = h - Q;
      e -= R; if (tt > e) e = tt;
      f -= R; if (tt > f) f = tt;
      Es[a] = e;
      hd = n0;
    }
    if (cm > S) { S = cm; bp = j; bq = cq; }
  }
  out_score[t] = S; out_pos[t] = bp; out_q[t] = bq;
}

// The same end points, one WAVE per sequence (the kernel the alignment phase uses; the one-thread form above
// remains as the 64-bit fallback for scoring systems whose scores could leave 32 bits).  Systolic like the
// search kernels but in plain int32 with position tracking: lane g owns query rows [row0 + g*K, +K), works on
// column t - g at step t, and hands H and F of its last row to lane g + 1 (ds_bpermute).  Database residues
// are staged through a 128-entry LDS ring one 64-column block ahead of use; the substitution matrix sits in
// LDS.  Queries longer than 64*K rows take several passes; the bottom row of a pass is handed over through
// bh/bf (one int pair per column, in place: lane 63 writes column t - 63 long after lane 0 read it).
// Ties as search16s.cc:391-405: among the cells holding the maximum, the smallest column, then the smallest row.
// POS = false: the score only (re-queue use) - no position bookkeeping in the inner loop
// one sequence [o, o + len) against the query, by the 64 lanes of the calling wave (a block of its own: M and ring are
// its LDS); returns the wave-wide best / first column / smallest row in every lane
// x of lane g - 1 in lane g of a wave64, 0 in lane 0: row_bcast:15 gives the lanes of rows 1..3 lane 15 of the row before
// (row 0 keeps the 0), row_shr:1 then overwrites every lane that has a left neighbour in its own row of 16 and leaves the
// four row heads alone (bound_ctrl off: an invalid source lane keeps what is there).  Both controls are what rocPRIM's
// warp scans use on every non-Navi target.
// (-DSWA_LANE_UP_BPERMUTE builds the form that ran on hardware until round 4 - ds_bpermute through the LDS crossbar - for an A/B.)
__device__ __forceinline__ int lane_up1(int x)
{
#ifdef SWA_LANE_UP_BPERMUTE
  const int up = __shfl_up(x, 1);
  return (threadIdx.x & 63) == 0 ? 0 : up;
#else
  const int heads = __builtin_amdgcn_update_dpp(0, x, 0x142, 0xe, 0xf, false);
  return __builtin_amdgcn_update_dpp(heads, x, 0x111, 0xf, 0xf, false);
#endif
}

template <int K, bool POS>
__device__ __forceinline__ void endpoints_wave_one(const int* M, uint8_t* ring, const swa_seqs& sq, int64_t o,
                                                   int len, bool rc, const uint8_t* __restrict__ qseq, int qlen, int Q, int R,
                                                   int* mybh, int* mybf, int& best, int& bcol, int& brow)
{
  const int g = threadIdx.x;
  auto residue = [&](int c) -> u32 {
    if (c >= len) return 0;
    return rc ? (__brev(seq_residue(sq, o + len - 1 - c)) >> 28) : seq_residue(sq, o + c);
  };
  best = 0; bcol = 0; brow = -1;
  for (int row0 = 0; row0 < qlen; row0 += 64 * K) {
    const bool first_pass = row0 == 0, more = row0 + 64 * K < qlen;
    int qs[K], hp[K], ee[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int r = row0 + g * K + k;
      qs[k] = r < qlen ? (int)qseq[r] : -1;
      hp[k] = 0;
      ee[k] = 0;
    }
    int pbest = 0, pcol = 0, prow = -1;
    int hin = 0, fin = 0, diag = 0;             // from the row above this lane's rows: H, F of column c; H of column c-1
    u32 nextd = residue(g);
    __syncthreads();
    const int steps = len + 63;
    for (int t = 0; t < steps; ++t) {
      if ((t & 63) == 0) {
        __syncthreads();
        ring[(t + g) & 127] = (uint8_t)nextd;
        nextd = residue(t + 64 + g);
        __syncthreads();
      }
      const int c = t - g;
      const bool active = c >= 0 && c < len;
      if (g == 0) {                              // top boundary: zeros, or the previous pass's bottom row
        if (first_pass || !active) { hin = 0; fin = 0; }
        else {                                   // agent-scope loads: written by lane 63 in the previous pass
          hin = __hip_atomic_load(mybh + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          fin = __hip_atomic_load(mybf + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      int hout = 0, fout = 0;
      if (active) {
        const int* mrow = M + ((int)ring[c & 127] << 5);
        int hd = diag, f = fin;
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const int n0 = hp[k];
          int e = ee[k];
          int h = hd + (qs[k] >= 0 ? mrow[qs[k]] : -1);
          h = max(max(h, f), max(e, 0));
          if constexpr (POS) {
            if (qs[k] >= 0 && h > pbest) { pbest = h; pcol = c; prow = row0 + g * K + k; }
          } else {
            pbest = max(pbest, qs[k] >= 0 ? h : 0);
          }
          hp[k] = h;
          const int tt = h - Q;
          e = max(e - R, tt);
          f = max(f - R, tt);
          ee[k] = e;
          hd = n0;
        }
        hout = hp[K - 1];
        fout = f;
        diag = hin;                              // H(row above, c) is the diagonal of column c + 1
        if (g == 63 && more) {
          __hip_atomic_store(mybh + c, hout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(mybf + c, fout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      // lane g - 1's last row to lane g: two DPP moves each instead of a ds_bpermute through the LDS crossbar (whose round
      // trip was a third of a step for the one wave that works on a sequence - the re-queue's tail)
      hin = lane_up1(hout);
      fin = lane_up1(fout);
    }
    if (pbest > best || (pbest == best && pbest > 0 && (pcol < bcol || (pcol == bcol && prow < brow)))) {
      best = pbest; bcol = pcol; brow = prow;
    }
    __threadfence();
    __syncthreads();
  }
  for (int sh = 32; sh > 0; sh >>= 1) {
    const int ob = __shfl_down(best, sh), oc = __shfl_down(bcol, sh), orow = __shfl_down(brow, sh);
    if (ob > best || (ob == best && ob > 0 && (oc < bcol || (oc == bcol && orow < brow)))) { best = ob; bcol = oc; brow = orow; }
  }
}

template <int K, bool POS = true>
__global__ void __launch_bounds__(64)
swa_endpoints_wave_kernel(swa_seqs sq, const int32_t* __restrict__ ids, const uint8_t* __restrict__ minus, int n,
                          const uint8_t* __restrict__ qseq, int qlen, const int32_t* __restrict__ matrix, int Q, int R,
                          int* __restrict__ bh, int* __restrict__ bf, const int64_t* __restrict__ boff,
                          long long* __restrict__ out_score, long long* __restrict__ out_pos, long long* __restrict__ out_q,
                          int* __restrict__ scores)
{
  __shared__ int M[1024];
  __shared__ uint8_t ring[128];
  const int w = blockIdx.x, g = threadIdx.x;
  if (w >= n) return;
  for (int i = g; i < 1024; i += 64) M[i] = matrix[i];
  int64_t o, len64;
  seq_span(sq, ids[w], o, len64);
  const int len = (int)len64;
  const bool rc = minus && minus[w];
  int best, bcol, brow;
  endpoints_wave_one<K, POS>(M, ring, sq, o, len, rc, qseq, qlen, Q, R, bh ? bh + boff[w] : nullptr,
                             bf ? bf + boff[w] : nullptr, best, bcol, brow);
  if (g == 0) {
    if (scores) scores[ids[w]] = best;           // re-queue use: the score of the sequence, in place
    else { out_score[w] = best; out_pos[w] = bcol; out_q[w] = brow; }
  }
}

// The re-queue list worked off WITHOUT the host: the first-pass kernel left `*count` sequence indices in `list`
// (ballot-compacted, one atomic per wave); a persistent grid of single-wave blocks takes entries off a work-queue head
// until min(*count, cap) and writes the exact int32 score of each in place.  The host learns the count only when the
// whole search has been enqueued and synchronises once (swipe_amd.cpp settle_search); lists longer than cap are taken
// over by the host there.  Single pass of the wave kernel only: qlen <= 64 K.
template <int K>
__global__ void __launch_bounds__(64)
swa_requeue_wave_kernel(swa_seqs sq, const int32_t* __restrict__ list, const int32_t* __restrict__ count, int cap, int32_t* __restrict__ work,
                        const uint8_t* __restrict__ qseq, int qlen, const int32_t* __restrict__ matrix, int Q, int R,
                        int* __restrict__ scores)
{
  __shared__ int M[1024];
  __shared__ uint8_t ring[128];
  __shared__ int next;
  const int g = threadIdx.x;
  int n = *count;
  if (n > cap) n = cap;
  if (n <= 0) return;
  for (int i = g; i < 1024; i += 64) M[i] = matrix[i];
  for (;;) {
    // the queue head goes through LDS + barrier, not "if (lane 0) atomic; readfirstlane": with the barriers of the body
    // inside this loop the compiler threaded the lanes' w = 0 past the readfirstlane and lanes 1..63 never left the loop
    __syncthreads();
    if (g == 0) next = atomicAdd(work, 1);
    __syncthreads();
    const int w = next;
    if (w >= n) break;
    const int id = list[w];
    int64_t o, len64;
    seq_span(sq, id, o, len64);
    const int len = (int)len64;
    int best, bcol, brow;
    endpoints_wave_one<K, false>(M, ring, sq, o, len, false, qseq, qlen, Q, R, nullptr, nullptr, best, bcol, brow);
    if (g == 0) scores[id] = best;
  }
}

// The same list, one BLOCK of four waves per sequence (round 5: the re-queue's tail).  What a search waits for at the end is the
// longest re-queued sequence on the ONE wave that works on it: steps x (rows per lane x the dependent chain of a cell).  Four
// waves on the sequence make the systolic array 256 lanes long - K = ceil(qlen / 256) rows per lane instead of ceil(qlen / 64):
// 2 instead of 6 for the 375-row bench query - at the price of one block barrier per step and 192 more steps of skew.  Lane
// g - 1 hands its last row to lane g by DPP inside a wave and through a double-buffered LDS word pair across the three wave
// boundaries (written before the step's barrier, read after it).  Same arithmetic, same queue head, same list as the wave
// kernel; no kernel waits for another, nothing spins.  Score only (the alignment phase keeps the wave kernel).
template <int K>
__global__ void __launch_bounds__(256)
swa_requeue_block_kernel(swa_seqs sq, const int32_t* __restrict__ list, const int32_t* __restrict__ count, int cap, int32_t* __restrict__ work,
                         const uint8_t* __restrict__ qseq, int qlen, const int32_t* __restrict__ matrix, int Q, int R,
                         int* __restrict__ scores)
{
  __shared__ int M[1024];
  __shared__ uint8_t ring[512];
  __shared__ int bnd[2][4][2];
  __shared__ int wbest[4];
  __shared__ int next;
  const int g = threadIdx.x, lane = g & 63, wave = g >> 6;
  int n = *count;
  if (n > cap) n = cap;
  if (n <= 0) return;
  for (int i = g; i < 1024; i += 256) M[i] = matrix[i];
  int qs[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int r = g * K + k;
    qs[k] = r < qlen ? (int)qseq[r] : -1;
  }
  for (;;) {
    // (the barrier at the head is also what keeps the lanes together from one entry to the next: see swa_requeue_wave_kernel)
    __syncthreads();
    if (g == 0) next = atomicAdd(work, 1);
    __syncthreads();
    const int w = next;
    if (w >= n) break;
    const int id = list[w];
    int64_t o, len64;
    seq_span(sq, id, o, len64);
    const int len = (int)len64;
    auto residue = [&](int c) -> u32 { return c < len ? seq_residue(sq, o + c) : 0u; };
    int hp[K], ee[K];
#pragma unroll
    for (int k = 0; k < K; ++k) { hp[k] = 0; ee[k] = 0; }
    int pbest = 0, hin = 0, fin = 0, diag = 0;
    u32 nextd = residue(g);
    const int steps = len + 255;
    for (int t = 0; t < steps; ++t) {
      if ((t & 255) == 0) {
        __syncthreads();
        ring[(t + g) & 511] = (uint8_t)nextd;
        nextd = residue(t + 256 + g);
        __syncthreads();
      }
      const int c = t - g;
      const bool active = c >= 0 && c < len;
      int hout = 0, fout = 0;
      if (active) {
        const int* mrow = M + ((int)ring[c & 511] << 5);
        int hd = diag, f = fin;
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const int n0 = hp[k];
          int e = ee[k];
          int h = hd + (qs[k] >= 0 ? mrow[qs[k]] : -1);
          h = max(max(h, f), max(e, 0));
          pbest = max(pbest, qs[k] >= 0 ? h : 0);
          hp[k] = h;
          const int tt = h - Q;
          e = max(e - R, tt);
          f = max(f - R, tt);
          ee[k] = e;
          hd = n0;
        }
        hout = hp[K - 1];
        fout = f;
        diag = hin;
      }
      int h1 = lane_up1(hout), f1 = lane_up1(fout);
      if (lane == 63) { bnd[t & 1][wave][0] = hout; bnd[t & 1][wave][1] = fout; }
      __syncthreads();
      if (lane == 0 && wave > 0) { h1 = bnd[t & 1][wave - 1][0]; f1 = bnd[t & 1][wave - 1][1]; }
      hin = h1;
      fin = f1;
    }
    for (int sh = 32; sh > 0; sh >>= 1) pbest = max(pbest, __shfl_down(pbest, sh));
    if (lane == 0) wbest[wave] = pbest;
    __syncthreads();
    if (g == 0) scores[id] = max(max(wbest[0], wbest[1]), max(wbest[2], wbest[3]));
  }
}

// ------------------------------------------------------------------ launchers
extern "C" hipError_t swa_launch_format(const swa_seqs* sq, const int32_t* slots, const swa_batch* batches, int nbatches,
                                        void* stream, int nibbles, hipStream_t st)
{
  if (nbatches <= 0) return hipSuccess;
  if (nibbles) hipLaunchKernelGGL(swa_format_stream4, dim3(nbatches), dim3(256), 0, st, *sq, slots, batches, nbatches, (uint8_t*)stream);
  else hipLaunchKernelGGL(swa_format_stream, dim3(nbatches), dim3(256), 0, st, *sq, slots, batches, nbatches, (uint16_t*)stream);
  return hipGetLastError();
}
// Pipelined open of protein volumes: a chunk of the .psq arrives as the file holds it - entries [residues NUL] back to back,
// cut anywhere - and is copied into the shard's residue array without the terminators.  The chunk holds raw bytes [c0, c1)
// of the range's entry stream, in which sequence s starts at offsets[s] + s (every earlier entry has one terminator); the
// sequences s0 .. s0 + n - 1 have residues in it.  One wave per sequence; *flags collects the OR of every residue byte
// (codes must stay below 32: they index the LDS profile); the terminators are not looked at, as in the reference
// (database.cc:1237-1258).
extern "C" __global__ void __launch_bounds__(256)
swa_unterminate(const uint8_t* __restrict__ chunk, long long c0, long long c1, const int64_t* __restrict__ offsets, int s0, int n,
                uint8_t* __restrict__ residues, unsigned* __restrict__ flags)
{
  const int lane = threadIdx.x & 63;
  const int64_t waves = (int64_t)gridDim.x * 4;
  unsigned acc = 0;
  for (int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); i < n; i += waves) {
    const int64_t s = s0 + i;
    const int64_t o = offsets[s], rs = o + s, re = offsets[s + 1] + s;       // residues at raw [rs, re)
    const int64_t b = rs > c0 ? rs : c0, e = re < c1 ? re : c1;
    const uint8_t* src = chunk + (b - c0);
    uint8_t* dst = residues + o + (b - rs);
    for (int64_t k = lane; k < e - b; k += 64) { const uint8_t v = src[k]; acc |= v; dst[k] = v; }
  }
  if (acc & ~0x1Fu) atomicOr(flags, acc);
}
// Pipelined open of NUCLEOTIDE volumes (database.cc:1237-1323 on the device).  A chunk holds WHOLE entries of the .nsq as the
// file has them: sequence s occupies raw bytes [raw[s], raw[s + 1]) = packed bases, 4 per byte, first base in the two high
// bits (the last packed byte carries the remainder count in its low 2 bits - the length is already known: offsets) followed
// by its ambiguity data: a big-endian header word (bit 31: 8-byte entries) and runs (code, length, position) that overwrite
// bases.  Output: the shard's 4-bit residue array (one-hot A=1 C=2 G=4 T=8, or the ambiguity code), residue i in byte i >> 1,
// low nibble first, whatever sequence it belongs to.  One wave per sequence; a lane takes one output dword (8 bases).  The
// first and the last dword of a sequence may be shared with its neighbours: those are OR-ed into the array, which the loader
// zeroed; the dwords in between are stored.  Ambiguity runs follow behind a fence, nibble by nibble (AND out, OR in), with the
// reference's last-writer-wins order where runs overlap.
extern "C" __global__ void __launch_bounds__(256)
swa_unpack_nt(const uint8_t* __restrict__ chunk, long long c0, const int64_t* __restrict__ raw, const int64_t* __restrict__ offsets,
              int s0, int n, uint8_t* __restrict__ residues)
{
  const int lane = threadIdx.x & 63;
  const int64_t waves = (int64_t)gridDim.x * 4;
  unsigned* out = reinterpret_cast<unsigned*>(residues);
  for (int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); i < n; i += waves) {
    const int64_t s = s0 + i;
    const int64_t o = offsets[s], len = offsets[s + 1] - o;
    const uint8_t* body = chunk + (raw[s] - c0);
    if (len > 0) {
      const int64_t d0 = o >> 3, d1 = (o + len - 1) >> 3;                 // output dwords [d0, d1]
      for (int64_t d = d0 + lane; d <= d1; d += 64) {
        unsigned w = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int64_t k = d * 8 + j - o;                                 // base of this sequence in nibble j of dword d
          if (k >= 0 && k < len) w |= (1u << ((body[k >> 2] >> ((3 - (int)(k & 3)) << 1)) & 3)) << (4 * j);
        }
        if (d == d0 || d == d1) atomicOr(out + d, w);
        else out[d] = w;
      }
    }
    const int64_t packed = (len >> 2) + 1, abytes = raw[s + 1] - raw[s] - packed;
    if (abytes >= 4) {
      __threadfence();                                                     // the bases are in place before a run overwrites them
      const uint8_t* a = body + packed;
      const unsigned hdr = ((unsigned)a[0] << 24) | ((unsigned)a[1] << 16) | ((unsigned)a[2] << 8) | a[3];
      const int esize = (hdr >> 31) ? 8 : 4;
      const int64_t nent = (abytes - 4) / esize;
      auto entry = [&](int64_t e, unsigned& code, int64_t& run, int64_t& off) {
        const uint8_t* q = a + 4 + e * esize;
        unsigned long long v = 0;
        for (int b = 0; b < esize; ++b) v = (v << 8) | q[b];
        if (esize == 8) { code = (unsigned)(v >> 60); run = (int64_t)((v >> 48) & 0xfff) + 1; off = (int64_t)(v & 0x0000fffffffffffULL); }
        else { code = (unsigned)(v >> 28) & 15u; run = (int64_t)((v >> 24) & 0xf) + 1; off = (int64_t)(v & 0x00ffffff); }
      };
      auto apply = [&](unsigned code, int64_t run, int64_t off) {
        for (int64_t r = 0; r < run && off + r < len; ++r) {
          const int64_t g = o + off + r;
          const int sh = 4 * (int)(g & 7);
          atomicAnd(out + (g >> 3), ~(15u << sh));
          atomicOr(out + (g >> 3), code << sh);
        }
      };
      // The reference applies the entries one after the other, so where runs overlap the last one in the file wins
      // (database.cc:1296-1321).  Formatters write them ascending and disjoint, and then the order does not matter: a lane per
      // entry.  A table that is not (an entry that starts before its predecessor ends) is applied by ONE lane, in file order.
      bool disorder = false;
      for (int64_t e = lane + 1; e < nent; e += 64) {
        unsigned c0_, c1_;
        int64_t r0, o0, r1, o1;
        entry(e - 1, c0_, r0, o0);
        entry(e, c1_, r1, o1);
        disorder |= o1 < o0 + r0;
      }
      if (__ballot(disorder) == 0) {
        for (int64_t e = lane; e < nent; e += 64) {
          unsigned code;
          int64_t run, off;
          entry(e, code, run, off);
          apply(code, run, off);
        }
      } else if (lane == 0) {
        for (int64_t e = 0; e < nent; ++e) {
          unsigned code;
          int64_t run, off;
          entry(e, code, run, off);
          apply(code, run, off);
        }
      }
    }
  }
}
extern "C" hipError_t swa_launch_unpack_nt(const uint8_t* chunk, long long c0, const int64_t* raw, const int64_t* offsets, int s0, int n,
                                           uint8_t* residues, hipStream_t st)
{
  if (n <= 0) return hipSuccess;
  const int blocks = (n + 3) / 4 < 16384 ? (n + 3) / 4 : 16384;
  hipLaunchKernelGGL(swa_unpack_nt, dim3(blocks), dim3(256), 0, st, chunk, c0, raw, offsets, s0, n, residues);
  return hipGetLastError();
}
extern "C" hipError_t swa_launch_unterminate(const uint8_t* chunk, long long c0, long long c1, const int64_t* offsets, int s0, int n,
                                             uint8_t* residues, unsigned* flags, hipStream_t st)
{
  if (n <= 0) return hipSuccess;
  const int blocks = (n + 3) / 4 < 16384 ? (n + 3) / 4 : 16384;
  hipLaunchKernelGGL(swa_unterminate, dim3(blocks), dim3(256), 0, st, chunk, c0, c1, offsets, s0, n, residues, flags);
  return hipGetLastError();
}
// the sequences the alignment phase wants back on the host, packed one after the other (one block per sequence)
extern "C" __global__ void __launch_bounds__(256)
swa_gather_sequences(swa_seqs sq, const int* __restrict__ ids, const int64_t* __restrict__ out_off, int n,
                     uint8_t* __restrict__ out)
{
  const int i = blockIdx.x;
  if (i >= n) return;
  int64_t o, len;
  seq_span(sq, ids[i], o, len);
  const int64_t d = out_off[i];
  for (int64_t k = threadIdx.x; k < len; k += blockDim.x) out[d + k] = (uint8_t)seq_residue(sq, o + k);   // one byte per residue
}
extern "C" hipError_t swa_launch_gather(const swa_seqs* sq, const int* ids, const int64_t* out_off, int n, uint8_t* out,
                                        hipStream_t st)
{
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(swa_gather_sequences, dim3(n), dim3(256), 0, st, *sq, ids, out_off, n, out);
  return hipGetLastError();
}
// a view's copy of a set's batch table: same steps, chunk offsets counted from the view's base pointer
extern "C" __global__ void swa_rebase_batches(const swa_batch* __restrict__ src, swa_batch* __restrict__ dst, int n, u32 delta)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { swa_batch b = src[i]; b.offset += delta; dst[i] = b; }
}
extern "C" hipError_t swa_launch_rebase(const swa_batch* src, swa_batch* dst, int n, uint32_t delta, hipStream_t st)
{
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(swa_rebase_batches, dim3((n + 255) / 256), dim3(256), 0, st, src, dst, n, delta);
  return hipGetLastError();
}
// long sequences searched as overlapping windows (ids nseq + v): the score of parent i is the maximum over its windows
// [wfirst[i], wfirst[i + 1]) - exactly, see swipe_amd.cpp "windows".  Scores beyond 32 bits sit in scores64 behind the
// sentinel, for windows as for sequences.
extern "C" __global__ void swa_fold_windows(int* __restrict__ scores, long long* __restrict__ scores64,
                                            const int32_t* __restrict__ parents, const int32_t* __restrict__ wfirst,
                                            int nparents, int nseq)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nparents) return;
  long long best = -1;
  for (int v = wfirst[i]; v < wfirst[i + 1]; ++v) {
    long long sc = scores[nseq + v];
    if (sc == SWA_SCORE_IN_64) sc = scores64[nseq + v];
    best = sc > best ? sc : best;
  }
  const int p = parents[i];
  if (best >= SWA_SCORE_IN_64) { scores[p] = SWA_SCORE_IN_64; scores64[p] = best; }
  else scores[p] = (int)best;
}
extern "C" hipError_t swa_launch_fold(int* scores, long long* scores64, const int32_t* parents, const int32_t* wfirst,
                                      int nparents, int nseq, hipStream_t st)
{
  if (nparents <= 0) return hipSuccess;
  hipLaunchKernelGGL(swa_fold_windows, dim3((nparents + 255) / 256), dim3(256), 0, st, scores, scores64, parents, wfirst, nparents, nseq);
  return hipGetLastError();
}
// excluded sequences (OID mask / taxid filter) report -1 so that no score threshold >= 0 ever accepts them
extern "C" __global__ void swa_mark_excluded(int* __restrict__ scores, const int* __restrict__ ids, int n)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) scores[ids[i]] = -1;
}
extern "C" hipError_t swa_launch_mark_excluded(int* scores, const int* ids, int n, hipStream_t st)
{
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(swa_mark_excluded, dim3((n + 255) / 256), dim3(256), 0, st, scores, ids, n);
  return hipGetLastError();
}
extern "C" hipError_t swa_launch_translate(const uint8_t* nt, int packed, const int64_t* ntoff, const int64_t* voff, int64_t nv,
                                           const uint8_t* table, uint8_t* prot, int64_t total, hipStream_t st)
{
  if (total <= 0) return hipSuccess;
  const int64_t want = (total + SWA_TR_TILE - 1) / SWA_TR_TILE;
  const int blocks = (int)(want < 8192 ? want : 8192);
  hipLaunchKernelGGL(swa_translate_frames, dim3(blocks), dim3(256), 0, st, nt, packed, ntoff, voff, nv, table, prot, total);
  return hipGetLastError();
}
extern "C" hipError_t swa_launch_endpoints(const swa_seqs* sq, const int32_t* ids, const uint8_t* minus, int n,
                                           const uint8_t* qseq, int qlen, const int32_t* matrix, long long Q, long long R,
                                           long long* Hs, long long* Es, long long* out, hipStream_t st)
{
  if (n <= 0) return hipSuccess;
  const int blocks = (n + 63) / 64;
  hipLaunchKernelGGL(swa_endpoints_kernel, dim3(blocks), dim3(64), 0, st, *sq, ids, minus, n, qseq, qlen, matrix, Q, R,
                     Hs, Es, out, out + n, out + 2 * (size_t)n);
  return hipGetLastError();
}
// wave-per-sequence end points; bh/bf/boff may be null when qlen <= 64 * rows-per-lane(qlen) (single pass).
// scores != null: re-queue use - only scores[ids[i]] is written (out may be null)
extern "C" int swa_endpoints_rows_for(int qlen)
{
  static const int rows[] = {2, 4, 6, 8, 12, 16, 24, 32};
  for (int r : rows) if (qlen <= 64 * r) return r;
  return 32;
}
extern "C" hipError_t swa_launch_endpoints_wave(const swa_seqs* sq, const int32_t* ids, const uint8_t* minus, int n, const uint8_t* qseq, int qlen,
                                                const int32_t* matrix, int Q, int R, int* bh, int* bf,
                                                const int64_t* boff, long long* out, int* scores, hipStream_t st)
{
  if (n <= 0) return hipSuccess;
#define SWA_EPW(KK) { if (scores) hipLaunchKernelGGL((swa_endpoints_wave_kernel<KK, false>), dim3(n), dim3(64), 0, st, *sq, ids, minus, n, \
                                       qseq, qlen, matrix, Q, R, bh, bf, boff, out, out, out, scores); \
                      else hipLaunchKernelGGL((swa_endpoints_wave_kernel<KK, true>), dim3(n), dim3(64), 0, st, *sq, ids, minus, n, \
                                       qseq, qlen, matrix, Q, R, bh, bf, boff, out, out + n, out + 2 * (size_t)n, scores); }
  switch (swa_endpoints_rows_for(qlen)) {
    case 2: SWA_EPW(2); break;
    case 4: SWA_EPW(4); break;
    case 6: SWA_EPW(6); break;
    case 8: SWA_EPW(8); break;
    case 12: SWA_EPW(12); break;
    case 16: SWA_EPW(16); break;
    case 24: SWA_EPW(24); break;
    default: SWA_EPW(32); break;
  }
#undef SWA_EPW
  return hipGetLastError();
}
extern "C" hipError_t swa_launch_requeue_wave(const swa_seqs* sq, const int32_t* list, const int32_t* count, int cap, int32_t* work, const uint8_t* qseq, int qlen,
                                              const int32_t* matrix, int Q, int R, int* scores, int blocks, hipStream_t st)
{
#define SWA_RQW(KK) hipLaunchKernelGGL((swa_requeue_wave_kernel<KK>), dim3(blocks), dim3(64), 0, st, *sq, list, count, \
                                       cap, work, qseq, qlen, matrix, Q, R, scores)
  switch (swa_endpoints_rows_for(qlen)) {
    case 2: SWA_RQW(2); break;
    case 4: SWA_RQW(4); break;
    case 6: SWA_RQW(6); break;
    case 8: SWA_RQW(8); break;
    case 12: SWA_RQW(12); break;
    case 16: SWA_RQW(16); break;
    case 24: SWA_RQW(24); break;
    default: SWA_RQW(32); break;
  }
#undef SWA_RQW
  return hipGetLastError();
}
// rows per lane of the block form: qlen <= 256 K, K = 1 .. 4 (0: the query is too long for it - the wave kernel's passes take it)
extern "C" int swa_requeue_block_rows_for(int qlen) { return qlen <= 1024 ? (qlen + 255) / 256 : 0; }
extern "C" hipError_t swa_launch_requeue_block(const swa_seqs* sq, const int32_t* list, const int32_t* count, int cap, int32_t* work, const uint8_t* qseq, int qlen,
                                               const int32_t* matrix, int Q, int R, int* scores, int blocks, hipStream_t st)
{
#define SWA_RQB(KK) hipLaunchKernelGGL((swa_requeue_block_kernel<KK>), dim3(blocks), dim3(256), 0, st, *sq, list, count, \
                                       cap, work, qseq, qlen, matrix, Q, R, scores)
  switch (swa_requeue_block_rows_for(qlen)) {
    case 1: SWA_RQB(1); break;
    case 2: SWA_RQB(2); break;
    case 3: SWA_RQB(3); break;
    case 4: SWA_RQB(4); break;
    default: return hipErrorInvalidValue;
  }
#undef SWA_RQB
  return hipGetLastError();
}
extern "C" hipError_t swa_launch_filter(const int* scores, const long long* scores64, int n, int which, long long minscore,
                                        long long maxscore, int* cand_count, int cand_cap, swa_cand* cand,
                                        unsigned long long* tallies, hipStream_t st)
{
  if (n <= 0) return hipSuccess;
  int blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(swa_filter_hits, dim3(blocks), dim3(256), 0, st, scores, scores64, n, which, minscore, maxscore,
                     cand_count, cand_cap, cand, tallies);
  return hipGetLastError();
}
