// EXPERIMENT, not part of the product: the re-queue drain inside the first-pass kernels as it stood when it was
// abandoned - kept so that tools/ubench/drain_test.hip builds and shows the codegen-dependent results on gfx950.
//
// ROOT CAUSE (round 5, found with tools/gfx950sim: the wrong scores are deterministic, single-wave mode included - no race).
// The claim loop of drain_list ends in a DIVERGENT branch - `if (lane == 0) scores[id] = best;` (or the lane-0 atomicAdd of
// DRAIN_MARK) - whose join point is the loop latch, and the next iteration starts with `if (lane == 0) { claim }` followed by
// the convergent readfirstlane.  hipcc threads the else side of the tail branch across the back-edge: on that path lane != 0
// is known, so `mine` is the constant -1, readfirstlane(-1) folds to -1 and the lanes go to `return`.  In the ISA the tail's
// lane-0 block executes `s_xor_b64 s[4:5], exec, -1` and the latch ORs s[4:5] into the set of lanes that left the loop: from
// the SECOND claimed sequence on EXEC = 0x1 and the whole wave DP runs in lane 0 alone (scores of at most two matches).  This
// is the hole in LLVM's convergence rules that its ConvergentOperations document describes (a convergent operation in a cycle
// whose iterations are not anchored; the convergence-control tokens that close it are not what hipcc 7.2 emits) - not a
// random miscompile, which is why it came and went with unrelated code (a printf or a call at the tail changes what can be
// threaded).  The rule for every persistent "claim, broadcast, work" loop in this repository: NO divergent branch may join
// at the latch - a uniform value is stored by every lane (-DSWA_DRAIN_FIXED below), or a convergent operation sits between
// the tail branch and the back-edge (swa_requeue_wave_kernel's __syncthreads at the loop head has been doing that since
// round 3, for the same reason, then unexplained).
// One database sequence against the query by the 64 lanes of ONE wave, 32-bit arithmetic: the body of the alignment
// phase's end-point kernel and of the re-queue (sw_util_kernels.hip), and - round 4 - of the re-queue DRAIN that every
// single-launch first-pass kernel runs when its queue of batches is exhausted (see drain_requeue below).
#ifndef SW_WAVE_DP_CUH
#define SW_WAVE_DP_CUH
#include "sw_common.cuh"

// (the product no longer has these two: they lived in sw_device.h / sw_common.cuh while the drain was wired into the kernels)
struct swa_drain {
  swa_seqs seqs;               /* the shard's sequences (and windows) by id */
  int32_t* work;               /* queue head of the list - the kernel behind continues from it */
  int32_t* work2;
  int32_t cap;                 /* entries of a list the device works off (the host preset that many to -1) */
  int32_t Q, R;                /* gap open + extend, gap extend */
  int32_t on;
};
// an entry of a list that waves of the SAME kernel may pick up while it runs: an agent-scope store (MI355X has one L2 per XCD;
// a plain store can sit dirty in the writer's L2 until the kernel ends)
__device__ __forceinline__ void list_publish(int32_t* at, int32_t id)
{
  __hip_atomic_store(at, id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// WAVE = false: the wave is a block of its own (64 threads) and synchronises with __syncthreads; true: it is one wave of a
// larger block whose other waves are elsewhere - LDS written by some lanes is read by others of the SAME wave only, which
// the hardware keeps in order (one wave's LDS instructions execute in issue order); the fence keeps the compiler from
// moving them across
template <bool WAVE> __device__ __forceinline__ void wave_dp_sync()
{
  if constexpr (WAVE) {
#ifdef SWA_WAVE_SYNC_WAIT
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_s_waitcnt(0xc07f);
#else
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#endif
    __builtin_amdgcn_wave_barrier();
  } else {
    __syncthreads();
  }
}

// one sequence [o, o + len) against the query, by the 64 lanes of the calling wave (a block of its own: M and ring are
// its LDS); returns the wave-wide best / first column / smallest row in every lane
template <int K, bool POS, bool WAVE = false>
__device__ __forceinline__ void endpoints_wave_one(const int* M, uint8_t* ring, const swa_seqs& sq, int64_t o,
                                                   int len, bool rc, const uint8_t* __restrict__ qseq, int qlen, int Q, int R,
                                                   int* mybh, int* mybf, int& best, int& bcol, int& brow)
{
  const int g = WAVE ? (int)(threadIdx.x & 63) : (int)threadIdx.x;
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
    wave_dp_sync<WAVE>();
    const int steps = len + 63;
    for (int t = 0; t < steps; ++t) {
      if ((t & 63) == 0) {
        wave_dp_sync<WAVE>();
        ring[(t + g) & 127] = (uint8_t)nextd;
        nextd = residue(t + 64 + g);
        wave_dp_sync<WAVE>();
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
      const int hnext = __shfl_up(hout, 1), fnext = __shfl_up(fout, 1);
      if (g > 0) { hin = hnext; fin = fnext; }
    }
    if (pbest > best || (pbest == best && pbest > 0 && (pcol < bcol || (pcol == bcol && prow < brow)))) {
      best = pbest; bcol = pcol; brow = prow;
    }
    __threadfence();
    wave_dp_sync<WAVE>();
  }
  for (int sh = 32; sh > 0; sh >>= 1) {
    const int ob = __shfl_down(best, sh), oc = __shfl_down(bcol, sh), orow = __shfl_down(brow, sh);
    if (ob > best || (ob == best && ob > 0 && (oc < bcol || (oc == bcol && orow < brow)))) { best = ob; bcol = oc; brow = orow; }
  }
  // lane 0 holds the wave's result: hand it to every lane.  Not a nicety: callers use it in lane 0 only, and when nothing but a
  // lane-0 branch consumed the reduction above, builds of the re-queue drain came out with the lane reads executed under that
  // branch's mask - lane 0 then saw only its own rows (tools/ubench/drain_test.hip: scores too small, depending on what else
  // was in the loop).  A read of lane 0 by all lanes pins the reduction where every lane takes part.
  best = __builtin_amdgcn_readfirstlane(best);
  bcol = __builtin_amdgcn_readfirstlane(bcol);
  brow = __builtin_amdgcn_readfirstlane(brow);
}


// ---- the re-queue drain ---------------------------------------------------------------------------------------------
// Sequences a first-pass kernel cannot settle (packed range left, or - bound build - bound at or above the threshold) are
// appended to a list and recomputed exactly, a wave per sequence.  Round 3 ran that beside the first pass as a SECOND
// kernel on a second stream (a "follower" polling the list), which rests on two kernels being resident together - HIP
// promises no such thing, and on MI355X a 512-thread producer froze beside spinning followers.  Now the producer does it
// itself: a wave whose queue of batches is exhausted takes entries off the list until none is left, then ends; whatever
// is appended after that (the last, shortest batches) is taken by ONE small kernel behind it in the same stream
// (swa_requeue_wave_kernel, same queue head).  No kernel waits for another kernel.  The long sequences - the ones that
// cost a wave hundreds of microseconds - are listed early (batches run longest first) and are recomputed by the first waves
// to run dry while the rest of the grid still works on batches.
// The list's first `cap` entries are preset to -1 by the host: a producer bumps the count BEFORE it writes its entries, so a
// claimed position may be empty for an instant; the claimer waits for it (the writer is a running wave of this very kernel,
// between two adjacent instructions).  Claims go through compare-and-swap on the queue head so that the head never passes
// the count and the kernel behind continues exactly where the drain stopped.
// LDS: [M: 32 x 32 ints][per wave: 128-byte residue ring (+ 16 spare)], after the kernel's own profile.
__host__ __device__ constexpr size_t drain_lds_bytes(int threads) { return 4096 + (size_t)(threads / 64) * 144; }

__device__ __forceinline__ void drain_init(const swa_drain& d, unsigned char* base, const int32_t* __restrict__ matrix)
{
  if (!d.on) return;
  int* M = (int*)base;
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) M[i] = matrix[i];
}

// one listed sequence, by the calling wave.  NOT inlined: one copy of the wave DP per translation unit and row count instead
// of one per first-pass build (compile time), its registers are allotted apart from the caller's hot loop - and inlined
// into the claim loop below the compiler produced code in which no wave ever left the DP (gfx950, ROCm 7.2; the same body
// called straight from a kernel runs: tools/ubench/drain_test.hip).
template <int KW>
__device__ __attribute__((noinline)) void drain_one(const uint8_t* residues, const int64_t* offsets, int packed, int nseq,
                                                    const int64_t* wstart, const int32_t* wlen, const int* M, uint8_t* ring, int id,
                                                    const uint8_t* qseq, int qlen, int Q, int R, int* scores)
{
  const swa_seqs sq{residues, offsets, packed, nseq, wstart, wlen};
  int64_t o, len64;
  seq_span(sq, id, o, len64);
  int best, bcol, brow;
  endpoints_wave_one<KW, false, true>(M, ring, sq, o, (int)len64, false, qseq, qlen, Q, R, nullptr, nullptr, best, bcol, brow);
  if ((threadIdx.x & 63) == 0) scores[id] = best;
}

template <int KW>
__device__ __forceinline__ void drain_list(const swa_drain& d, unsigned char* base, const int32_t* list, const int32_t* count,
                                           int32_t* work, const uint8_t* __restrict__ qseq, int qlen, int* __restrict__ scores)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int* M = (const int*)base;
  uint8_t* ring = base + 4096 + wave * 144;
#ifdef SWA_DRAIN_DEBUG
#define DRAIN_MARK(k) do { if (lane == 0) atomicAdd(d.work + 12 + (k), 1); } while (0)
#else
#define DRAIN_MARK(k) do { } while (0)
#endif
  for (;;) {
    int mine = -1;
    if (lane == 0) {
      int id = -1, w = -1;
      int head = __hip_atomic_load(work, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (;;) {
        int n = __hip_atomic_load(count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (n > d.cap) n = d.cap;
        if (head >= n) break;
        const int seen = atomicCAS(work, head, head + 1);
        if (seen == head) { w = head; break; }
        head = seen;
      }
      if (w >= 0) {
        DRAIN_MARK(0);
        for (;;) {
          id = __hip_atomic_load(list + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (id >= 0) break;
          __builtin_amdgcn_s_sleep(2);
        }
      }
      if (w >= 0) DRAIN_MARK(1);
      mine = id;
    }
    // lane 0's claim to every lane, as a scalar: everything below is uniform control flow
    const int id = __builtin_amdgcn_readfirstlane(mine);
    if (id < 0) return;
    DRAIN_MARK(2);
#ifdef SWA_DRAIN_INLINE
    {
      int64_t o, len64;
      seq_span(d.seqs, id, o, len64);
      int best, bcol, brow;
      endpoints_wave_one<KW, false, true>(M, ring, d.seqs, o, (int)len64, false, qseq, qlen, d.Q, d.R, nullptr, nullptr, best, bcol, brow);
#ifdef SWA_DRAIN_FIXED
      scores[id] = best;                         // uniform value, stored by every lane: nothing diverges before the back-edge
#else
      if (lane == 0) scores[id] = best;          // the tail branch that gets threaded across the back-edge (see the file header)
#endif
#ifdef SWA_DRAIN_PRINT
      if (lane == 0 && id < 3) printf("id %d o %lld len %d qlen %d Q %d R %d M33 %d M34 %d ringoff %d q0 %d r0 %d best %d\n", id, (long long)o, (int)len64, qlen, d.Q, d.R, M[33], M[34], (int)(ring - base), (int)qseq[0], (int)seq_residue(d.seqs, o), best);
#endif
    }
#else
    drain_one<KW>(d.seqs.residues, d.seqs.offsets, d.seqs.packed, d.seqs.nseq, d.seqs.wstart, d.seqs.wlen, M, ring, id, qseq, qlen, d.Q, d.R, scores);
#endif
    DRAIN_MARK(3);
  }
}
#endif
