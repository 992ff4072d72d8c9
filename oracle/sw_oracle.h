/* TEST INFRASTRUCTURE - CPU restatement of the reference (torognes/swipe v2.1.1) hot path.

   Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
   library, and only as the checker.  The product (swipe_amd/) never links or calls it.

   Parity status: PINNED.  Every function below is checked against outputs of the compiled
   reference itself (oracle/_ref/ref_harness and oracle/_ref/swipe, built by oracle/Makefile
   from /root/reference) by tests/test_oracle_vs_reference.py in the build container, and
   against the committed fixtures those runs produced (tests/golden/) everywhere else.
   The reference ships no tests or golden vectors of its own.

   All citations are file:line into /root/reference.
*/
#ifndef SW_ORACLE_H
#define SW_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- score matrices (matrices.cc:520-591) ------------------------------------------ */
/* M is long[32*32], index (db_symbol << 5) + query_symbol, unset cells = -1 */
void swo_matrix_clear(long* M);                                    /* matrices.cc:531 */
int  swo_matrix_builtin(const char* name, long* M);                /* matrices.cc:540-559; 1 if known */
void swo_matrix_nucleotide(long match, long mismatch, long* M);    /* matrices.cc:533-538 */
int  swo_matrix_parse(const char* text, long* M);                  /* matrices.cc:352-430 (file/string form) */
/* lo/hi over all 1024 cells and the lane-width limits: 128-hi, 65536-hi (matrices.cc:561-578) */
void swo_score_limits(const long* M, long* lo, long* hi, long* limit7, long* limit16);

/* ---- DP kernels ---------------------------------------------------------------------- */
/* search63.cc:28-89.  gap penalties arrive as BYTE exactly as in the reference prototype. */
long swo_fullsw(const unsigned char* dseq, long dlen, const unsigned char* qseq, long qlen,
                const long* M, unsigned char gapopenextend, unsigned char gapextend);
/* One lane of search7 (search7.cc:565-958): 0x80-biased bytes, paddsb/psubsb/pmaxub,
   4-column blocks zero-padded past the end.  Returns what scores[i] receives. */
long swo_search7_lane(const unsigned char* dseq, long dlen, const unsigned char* qseq, long qlen,
                      const long* M, unsigned char gapopenextend, unsigned char gapextend);
/* One lane of search16 (search16.cc:99-546): 0x8000-biased words, paddsw/psubsw/pmaxsw.
   *bestpos = d_best - d_begin as at search16.cc:411-414/464. */
long swo_search16_lane(const unsigned char* dseq, long dlen, const unsigned char* qseq, long qlen,
                       const long* M, unsigned short gapopenextend, unsigned short gapextend,
                       long* bestpos);

/* One lane of search16s (search16s.cc:106-548): as search16 with ONE column per block; *bestpos =
   0-based column where the final maximum was first reached, *bestq = smallest row holding it in
   that column (search16s.cc:391-405); (0, -1) if the score never rose above zero. */
long swo_search16s_lane(const unsigned char* dseq, long dlen, const unsigned char* qseq, long qlen,
                        const long* M, unsigned short gapopenextend, unsigned short gapextend,
                        long* bestpos, long* bestq);

/* ---- the per-chunk escalation loop (swipe.cc:1416-1592) ---------------------------- */
typedef struct {
  long compute7, compute16, compute63;   /* swipe.cc:1426, 1495, 1553 */
} swo_counters;
/* residues: concatenated db sequences, sequence s = residues[offsets[s] .. offsets[s+1]).
   scores[s] receives the score the reference passes to hits_enter for sequence s. */
void swo_search_chunk(const unsigned char* residues, const int64_t* offsets, long nseq,
                      const unsigned char* qseq, long qlen, const long* M,
                      long gapopenextend, long gapextend, long* scores, swo_counters* counters);
/* same result, scalar 63-bit recurrence only, `threads` pthreads over sequences (the "port"
   CPU baseline of bench.py; also the fast checker for large parity cases) */
void swo_search_all63(const unsigned char* residues, const int64_t* offsets, long nseq,
                      const unsigned char* qseq, long qlen, const long* M,
                      long gapopenextend, long gapextend, long* scores, int threads);

/* ---- Karlin-Altschul statistics (stats.cc:44-247, blastkar_partial.c:656-748) ------- */
typedef struct { double lambda, K, H, alpha, beta; } swo_ka;
int  swo_stats_protein(const char* matrix, long gapopen, long gapextend, swo_ka* p);          /* stats.cc:169-247 */
int  swo_stats_nucleotide(long match, long mismatch, long gapopen, long gapextend, swo_ka* p); /* stats.cc:44-167 */
int  swo_stats_default_gaps(const char* matrix, long* gapopen, long* gapextend);              /* stats.cc:249-325 */
int  swo_length_adjustment(double K, double logK, double alpha_d_lambda, double beta,
                           int query_length, long db_length, int db_num_seqs, int* adj);      /* blastkar_partial.c:656 */

/* ---- hit list (hits.cc:163-222, 283-511, 1777-1779) ------------------------------------ */
typedef struct { long seqno, score, qstrand, qframe, dstrand, dframe; } swo_hit;
typedef struct {
  long keephits, count, scorethreshold, upperscorethreshold, init_threshold, totalhits, obvious;
  int stats_available;
  double lambda, K, Kmn, logK, lambda_d_log2, logK_d_log2;
  long lenadj, m, n;
  swo_hit* list;
} swo_hits;
/* hits_init for symtype 0 (nucleotide) and 1 (protein); dbseqs/dbsyms as db_getseqcount/symcount */
swo_hits* swo_hits_new(long descriptions, long alignments, long minscore, long maxscore,
                       double minexpect, double expect, int symtype, int querystrands,
                       const char* matrix, long match, long mismatch, long gapopen, long gapextend,
                       long qlen, long dbseqs, long dbsyms, long effdbsize);
void swo_hits_enter(swo_hits* h, long seqno, long score, long qstrand, long qframe, long dstrand, long dframe);
double swo_hits_expect(const swo_hits* h, long score);   /* Kmn * exp(-lambda*score) */
double swo_hits_bits(const swo_hits* h, long score);     /* lambda/ln2*score - lnK/ln2 */
void swo_hits_free(swo_hits* h);

/* ---- translated searches (query.cc:377-506, database.cc:1182-1218) -------------------------- */
/* table[4096] indexed 256*a + 16*b + c by three IUPAC nibbles; 0 for an unassigned genetic code number */
int  swo_translate_table(int gencode, unsigned char* table);
/* frame (0..2) of strand (0/1) of dna[dlen] (nibble codes) into prot; returns (dlen - frame) / 3 */
long swo_translate(const unsigned char* dna, long dlen, int strand, int frame, const unsigned char* table,
                   unsigned char* prot);

/* ---- alignment phase (align.cc:38-519 as called from hits_align, hits.cc:546-618) ---------- */
typedef struct { long q_start, d_start, q_end, d_end, score; } swo_alignment;
/* gapopen/gapextend as passed by hits_align (NOT open+extend).  hint_score != 0: trust
   (hint_q_end, hint_d_end) as the end cell (the search16s result, hits.cc:587-596); otherwise the
   forward pass finds score and end (first strict maximum in query-row-major order).  Writes the
   edit script ("M12D2I1...", D = query symbol against a gap, I = database symbol against a gap)
   NUL-terminated into cigar; returns its full length, or -1 where the reference would stop with
   "Internal error in align function." */
long swo_align(const unsigned char* qseq, long qlen, const unsigned char* dseq, long dlen, const long* M,
               long gapopen, long gapextend, long hint_score, long hint_q_end, long hint_d_end,
               swo_alignment* res, char* cigar, long cigar_room);

#ifdef __cplusplus
}
#endif
#endif
