// Structures shared by the device kernels (sw_kernels.hip) and the host library (swipe_amd.cpp).
#ifndef SW_DEVICE_H
#define SW_DEVICE_H
#include <stdint.h>

#define SWA_SLOTS 8            /* sequences per batch: 4 DPP rows x 2 packed halves */
#define SWA_PAD 31             /* residue code used past the end of a sequence; scores -1 vs everything.
                                  No reference alphabet produces code 31 (query.cc:51-109), and the
                                  reference's matrices leave it at the default -1 (matrices.cc:531). */
#define SWA_SCORE_IN_64 0x7fffffff   /* scores[i] sentinel: the value lives in scores64[i] */

struct swa_batch {
  uint32_t offset;             /* start of the batch in the residue stream, in 128-byte chunks */
  int32_t steps;               /* DP columns the batch needs = longest sequence rounded up to 2;
                                  the stream holds ceil(steps / 16) chunks of 16 columns */
};

/* How kernels address the sequences of a shard.  ids < nseq are database sequences, residues [offsets[id],
   offsets[id + 1]); ids >= nseq are WINDOWS of long sequences (swipe_amd.cpp "windows"): residues
   [wstart[id - nseq], + wlen[id - nseq]) of the same array.  packed: two 4-bit residues per byte, low nibble
   first (nucleotide shards; residue index i lives in byte i >> 1). */
struct swa_seqs {
  const uint8_t* residues;
  const int64_t* offsets;
  int32_t packed;
  int32_t nseq;
  const int64_t* wstart;
  const int32_t* wlen;
};

struct swa_cand {              /* one survivor of the hit filter (hits_enter's acceptance test on the device) */
  long long score;
  int32_t idx;                 /* shard-local sequence index */
  int32_t which;               /* 0 / 1: first / second score array of a two-query search */
};

struct swa_query {
  const uint8_t* qseq;         /* query residues, reference symbol codes (< 32) */
  const int32_t* matrix;       /* 32x32, (db symbol << 5) | query symbol */
  int32_t qlen;
};

struct swa_narrow_params {
  const swa_query* query;
  const uint16_t* stream;      /* [batch][chunk][row 0..3][lane 0..15] residue pairs (A | B << 8) */
  const swa_batch* batches;
  const int32_t* slots;        /* [batch][SWA_SLOTS] shard-local sequence index or -1 */
  int32_t nbatches;
  int32_t* counter;            /* work queue head */
  int32_t* scores;             /* [nseq] */
  int32_t limit;               /* 2048 - hi: scores >= limit are re-queued */
  int32_t* ovf_count;
  int32_t* ovf_list;
  uint32_t negQ, negR;         /* packed f16 pairs: -(gapopen+gapextend), -gapextend */
  /* row-shifted form (swa_narrow_split_kernel) */
  int32_t shifted;             /* 0 plain form, 1 row-shifted form */
  int32_t waves;               /* tuning: waves per SIMD the kernel is compiled for (0 = default) */
  int32_t pipe;                /* split kernel build: 0 staged, 1 pipelined within a step (K = 30..36), 2 across steps (K = 45..48);
                                  -1 = whichever measured fastest for this K */
  float gapextend_f;           /* R, added to every profile entry */
  uint32_t negQR, negKR;       /* packed f16 pairs: -(gapopen) = -(Q - R), -K R */
  uint32_t rowc[80];           /* packed f16 pairs r*R for r = 0..K+1 (bound build: ..K+period+1) */
  /* one pass of a long query (MP build of swa_narrow_split_kernel) */
  void* boundary;              /* (H, F) f16 pairs of the pass's last row: 8 bytes per stream element, stream layout */
  long long boundary_base;     /* stream chunk (swa_batch.offset units) that boundary[0] belongs to */
  int32_t row0;                /* first query row of the pass */
  int32_t pass, last;          /* pass index; 1 when no pass follows */
  /* bound builds, single pass (sw_cb_kernel.inc, round 6) */
  int32_t concat;              /* batches a chain works through back to back, without draining or resetting in between
                                  (an item of the work queue = concat x 16 / G batches); <= 1: one, the round-3 kernel */
  int32_t concat_items;        /* items of that size at the head of the queue; the batches behind them go one set of 16 / G at a time */
  int32_t twin;                /* builds at two waves per SIMD: blocks of 8 waves with the profile twice, the second copy - N R (step 0
                                  of a period renormalises H through it); 0: blocks of 4 waves, one copy, the round-3 form */
};

/* generic multi-pass kernel (sw_mp_kernel.inc) */
struct swa_mp_params {
  const uint8_t* qseq;         /* query 1 */
  const uint8_t* qseq2;        /* query 2 (dual mode), same length */
  const int32_t* matrix;
  int32_t qlen, npass, rows_per_lane;
  int32_t tune_w;              /* tuning override of the waves-per-SIMD build (0 = default) */
  int32_t nibbles;             /* swa_dual_kernel, 16-lane chains: the stream holds 4-bit residues, 32 bytes per chunk */
  int32_t qlen_a, qlen_b;      /* two DIFFERENT queries (swa_search_pair_topk): rows of query 1 / 2, the shorter padded to
                                  qlen with rows that score -1 against everything; 0 = qlen */
  const uint16_t* stream;
  const swa_batch* batches;
  const int32_t* slots;
  int32_t nbatches;
  int32_t* counter;            /* super-batch queue head (one grab = one batch per wave of a block) */
  int32_t* scores;
  int32_t* scores2;            /* dual mode: scores of query 2 */
  long long* scores64;
  long long limit;
  int32_t* ovf_count;
  int32_t* ovf_list;
  int32_t* ovf_count2;
  int32_t* ovf_list2;
  void* boundary;              /* per wave: boundary_cols x 4 rows x (H, F) */
  int32_t boundary_cols;
  /* pass builds of swa_dual_kernel: boundary holds 8 bytes per stream element, as in swa_narrow_params */
  long long boundary_base;
  int32_t row0, pass, last;
  long long gapopenextend, gapextend;
  float gapextend_f;
  uint32_t negQR, negR, negKR;
  uint32_t rowc[80];
  /* bound build of the two-query kernel (sw_cb_dual_kernel.inc, round 6): sets of batches a chain works through back to back;
     concat_items is filled in by the launcher (items of that size at the head of the queue), concat_tail = sets handed out singly
     at its end (-1: four per resident wave) */
  int32_t concat, concat_items, concat_tail;
};
#endif
