/* Synthetic database generator (bench/test infrastructure shipped with the library so that the
   10M-100M sequence benchmark databases of BASELINE.json can be produced on the GPU box).
   Pure function of (seed, seqno, position); bit-identical to swipe_amd/synth.py. */
#ifndef SWIPE_AMD_SYNTH_H
#define SWIPE_AMD_SYNTH_H
#include <stdint.h>
#ifndef SWA_API
#define SWA_API __attribute__((visibility("default")))
#endif
#ifdef __cplusplus
extern "C" {
#endif
/* Length of synthetic sequence `seqno` (query/qlen: the planted-homolog template, may be NULL/0). */
SWA_API int64_t swa_synth_length(uint64_t seed, int64_t seqno, const int32_t* len_table /*4096*/,
                         const uint8_t* query, int64_t qlen);
/* Fills offsets[0..nseq] (offsets[0] = 0) for seqnos [first, first+nseq); returns total residues. */
SWA_API int64_t swa_synth_offsets(uint64_t seed, int64_t first, int64_t nseq, const int32_t* len_table,
                          const uint8_t* query, int64_t qlen, int64_t* offsets, int threads);
/* Writes the residues of seqnos [first, first+nseq) at residues + offsets[i]. */
SWA_API int swa_synth_fill(uint64_t seed, int64_t first, int64_t nseq, const int32_t* len_table,
                   const uint8_t* res_table /*4096*/, const uint8_t* query, int64_t qlen,
                   const int64_t* offsets, uint8_t* residues, int threads);
#ifdef __cplusplus
}
#endif
#endif
