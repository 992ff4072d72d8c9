/* swipe_amd - MI355X-native inter-sequence Smith-Waterman database search.

   C ABI of the drop-in boundary described in SURVEY.md section 8(b).  SWIPE has no plugin
   API; its narrowest seam around the hot path is search_chunk() (swipe.cc:1365-1596):
   input = a contiguous range of database sequence numbers, output = one exact local
   alignment score per sequence, handed to hits_enter().  The functions below replace
   exactly that seam and the few reference functions that feed it:

     reference (file:line)                              this ABI
     --------------------------------------------------------------------------------------
     db_open + db_mapsequences  (database.cc:775,1082)  swa_db_open / swa_db_from_memory
     db_getseqcount/symcount/longest (swipe.h:305-309)  swa_db_info
     score_matrix_init -> score_matrix_63, limits
                        (matrices.cc:520-591)           swa_set_scoring
     search7 + search16 + fullsw under search_chunk
                        (swipe.h:200-258, swipe.cc:1416-1592)  swa_search
     hits_enter loop + top-K list (hits.cc:163-222)     swa_search_topk / swa_hits_*
     search16s end points (swipe.h:237-249)             swa_search_endpoints[_strand]
     hits_init thresholds, E-values (hits.cc:283-511,
                        1777-1779; stats.cc)            swa_stats_init / swa_evalue / swa_bits
     the qstrand/qframe x dstrand/dframe loops of search_chunk
                        (swipe.cc:277-337, 1379-1404)   swa_search_frames_topk / swa_fhits_merge
     translate, db_translate (query.cc:377-506,
                        database.cc:1182-1218)          swa_translate[_table] / swa_db_open_translated
     align_chunk + hits_align + align (swipe.cc:339-414,
                        hits.cc:546-618, align.cc)      swa_align_hits / swa_traceback / swa_db_sequence
     db_check_inclusion, db_showheader, alias masks
                        (database.cc:670-772, 1424-1481,
                        asnparse.cc)                    swa_headers_* / swa_db_set_inclusion

   Conventions: plain pointers and sizes, caller owns every buffer, the callee keeps no host
   pointer past return.  Every function returns SWA_OK (0) or a negative SWA_E* code; the
   message for the last failure on the calling thread is swa_last_error().  (The reference
   never returns an error: fatal() prints and exit(1)s, swipe.cc:158-170 - the CLI driver on
   top of this ABI reproduces that.)  One host thread per device handle.
   There is NO CPU fallback: without a usable HIP device every compute entry point fails.
*/
#ifndef SWIPE_AMD_H
#define SWIPE_AMD_H
#include <stdint.h>
#ifndef SWA_API
#define SWA_API __attribute__((visibility("default")))
#endif
#ifdef __cplusplus
extern "C" {
#endif

#define SWA_OK          0
#define SWA_EINVAL     -1   /* bad argument */
#define SWA_ENODEV     -2   /* no HIP device / HIP runtime error */
#define SWA_ENOMEM     -3   /* device or host memory; no C++ exception leaves the library (a failed host allocation is this
                               status where the reference's xmalloc ends the process, swipe.cc:158-182) */
#define SWA_EIO        -4   /* database files unreadable or malformed */
#define SWA_ESTATE     -5   /* call order (e.g. search before set_scoring) */
#define SWA_ERANGE     -6   /* caller's buffer too small; the needed size is reported */

#define SWA_SYMTYPE_NUCLEOTIDE 0   /* reference symtype 0: 4-bit base masks, A=1 C=2 G=4 T=8 */
#define SWA_SYMTYPE_PROTEIN    1   /* reference symtype 1: NCBIstdaa codes 0..27 */
/* translated searches (reference -p 2/3/4, swipe.cc:289-324): the names only select statistics in
   swa_stats_init; the searches themselves are protein searches over translated frames */
#define SWA_SYMTYPE_TRANSLATED_QUERY 2   /* nucleotide query in 3/6 frames against a protein database */
#define SWA_SYMTYPE_TRANSLATED_DB    3   /* protein query against a nucleotide database in 6 frames */
#define SWA_SYMTYPE_TRANSLATED_BOTH  4   /* 3/6 query frames against 6 database frames */

typedef struct swa_db swa_db;     /* one database shard resident in HBM, read-only after open */

typedef struct {
  int64_t seqcount;      /* sequences in this shard */
  int64_t symcount;      /* residues in this shard */
  int64_t longest;       /* longest sequence in this shard */
  int64_t first_seqno;   /* global number of the shard's first sequence */
  int64_t total_seqcount, total_symcount;   /* whole database (all shards), for statistics */
  int64_t hbm_bytes;     /* device memory held */
  int64_t frames;        /* 1, or 6 for a nucleotide shard held as its six translations (then seqcount,
                            symcount, longest and total_* count nucleotide sequences / bases) */
} swa_db_info_t;

/* counters of the escalation loop; the reference's compute7/compute16/compute63
   (swipe.cc:111-119) counted sequences per lane width in the same way */
typedef struct {
  int64_t narrow;        /* sequences scored by the packed 16-bit-lane kernel */
  int64_t wide;          /* re-queued to the 32-bit kernel (score >= narrow limit) */
  int64_t full;          /* re-queued to the 64-bit kernel (score >= 2^31 - hi) */
  int64_t cells;         /* DP cells computed in the first pass (symcount * qlen) */
  double  kernel_ms;     /* device time of the first-pass kernel (HIP events) */
  double  total_ms;      /* device time of the whole search (HIP events) */
  int32_t narrow_rows;   /* query rows per lane (K) of the first-pass kernel, 0 if it did not run */
  int32_t narrow_shifted;/* 0: plain form (8.5 ops per cell pair); row-shifted form (7.5) with 1: 16, 2: 8, 3: 4 lanes
                            per sequence pair (queries of at most 928 / 384 / 192 rows); 4: single-pass two-query kernel;
                            5: row-shifted form, one launch per pass of 16 x narrow_rows query rows (queries > 928 rows);
                            6: two-query kernel, one launch per pass (queries > 1008 nucleotide / 512 other rows);
                            7: row-shifted form with 2 lanes per sequence pair (queries of 81..96 rows; at most 96 with lanes=2);
                            8: bound build of the row-shifted form (swa_search_topk only, see there);
                            9: bound build, one launch per pass (queries > 928 rows);
                            10: bound build of the two-query kernel (non-nucleotide pairs of 65..512 rows);
                            11: row-shifted form, ONE lane per sequence pair (queries of at most 48 rows);
                            12: two-query kernel, ONE lane per sequence (at most 48 nucleotide / 32 other rows) */
  int32_t loading_parts; /* > 0: the shard was still loading (swa_db_open_async) and the first pass ran part by part, one launch
                            per part as the parts arrived; kernel_ms then includes waiting for them.  0: one resident shard */
  int32_t requeue_form;  /* how the sequences that left the packed range were recomputed (search16's role, search16.cc:320-546):
                            0 nothing was launched for them (no first pass); 1 on the device behind the first pass, one wave per
                            sequence; 2 the same with a block of four waves per sequence (option requeue_block); 3 by the host-driven
                            32/64-bit kernels after the synchronisation.  A list the device kernel could not finish (more than
                            65 536 entries, scores beyond 32 bits) is completed by form 3 and still reports 1 or 2 */
} swa_counters_t;

typedef struct { int64_t seqno; int64_t score; } swa_hit_t;

SWA_API const char* swa_last_error(void);
SWA_API int swa_device_count(void);

/* Debugging aid (the reference has none; CHANGES:54 mentions Valgrind once): with SWA_REDZONES=1 in the environment when
   the library is first used, every device allocation carries a 4 KiB guard region in front and one right behind its last
   requested byte, filled with a pattern.  swa_redzones_check reads all guards of all live allocations of the process back
   (after a device synchronisation): *buffers = allocations looked at, *touched = guard bytes that changed - a kernel wrote
   outside its buffer; report (may be NULL) describes the first few.  SWA_ESTATE when red zones are off. */
SWA_API int swa_redzones_check(int64_t* buffers, int64_t* touched, char* report, int64_t report_cap);

/* ---- database --------------------------------------------------------------------------- */
/* Opens BLAST v4 volume(s) `basename` (.pin/.psq or .nin/.nsq, or a .pal/.nal alias) and loads
   the sequences [first_seqno, last_seqno] (last_seqno < 0: to the end) onto `device`,
   re-formatted for the kernels.  Mirrors db_open + db_mapsequences. */
SWA_API int swa_db_open(const char* basename, int symtype, int device,
                int64_t first_seqno, int64_t last_seqno, swa_db** out);
/* The same without waiting for the residues: returns once the index files are read and the device buffers exist, while a
   loader thread streams the sequence files into HBM (reader threads -> page-locked ring -> copy engine -> format kernels,
   every stage overlapped).  The reference does the same thing with mmap: db_mapsequences (database.cc:1082-1131) maps a
   chunk at a time, so its first search runs while the files page in (swipe.cc:1716-1742).  Every entry point accepts the
   handle at once: swa_search and swa_search_topk with a single-pass query START on the parts that have arrived and follow
   the loader (same results as on the resident shard; counters.loading_parts > 0 says it happened), everything else
   waits for the load.  A load error (unreadable file, residue code out of range) is returned by the first call that needs
   the data and by swa_db_wait; the handle can then only be closed.  swa_db_open is this call + swa_db_wait.
   Pipelined for every regular database (round 5): protein volumes ([residues NUL]* entries), nucleotide volumes (.nsq entries
   [packed bases | ambiguity data], unpacked on the device, database.cc:1237-1323), with or without the OID mask of a MEMB_BIT
   alias (the excluded sequences are marked; the adopted tables hold the members).  Only an index whose entries overlap or
   run backwards is left to the old reader, which reports it.  Options (environment, read when the open begins): SWA_PIPELINED=0 old reader always; SWA_LOAD_PART /
   SWA_LOAD_CHUNK bytes per part / per page-locked staging chunk; SWA_LOAD_THREADS reader threads; SWA_LOAD_TRACE=1. */
SWA_API int swa_db_open_async(const char* basename, int symtype, int device,
                      int64_t first_seqno, int64_t last_seqno, swa_db** out);
SWA_API int swa_db_wait(swa_db* db);      /* blocks until the shard is resident; the load's status */
/* how far the load is: bytes of sequence file handed to the copy engine / in all, parts searchable / in all (all zero on a
   handle that is not loading, or no longer).  A shard over its HBM budget (swa_db_open_streamed, whose parts are filled behind
   the open): page-locked bytes of the parts that are in place / of all parts, parts in place / in all - these stay when the
   load is through, the parts do.  swa_db_wait on such a handle blocks until every part is in place. */
SWA_API int swa_db_load_progress(swa_db* db, int64_t* bytes_loaded, int64_t* bytes_total, int32_t* parts_ready, int32_t* parts_total);
/* Same from host arrays: sequence s = residues[offsets[s] .. offsets[s+1]) in reference symbol
   codes.  total_* describe the whole database when this is one shard of it (pass 0 to use
   the shard's own counts). */
SWA_API int swa_db_from_memory(const uint8_t* residues, const int64_t* offsets, int64_t nseq,
                       int symtype, int device, int64_t first_seqno,
                       int64_t total_seqcount, int64_t total_symcount, swa_db** out);
/* Translated database (reference -p 3 / -p 4: db_getsequence translating on every fetch,
   database.cc:1360-1388).  Here the nucleotide shard is translated ONCE, on the GPU, into its six frames
   (genetic code db_gencode, 1..23) and from then on behaves as a protein shard of 6 * nseq virtual
   sequences, virtual index 6 * (seqno - first_seqno) + 3 * dstrand + dframe, in the reference's order.
   swa_search then delivers 6 scores per sequence; hit lists come from swa_search_frames_topk. */
SWA_API int swa_db_open_translated(const char* basename, int db_gencode, int device, int64_t first_seqno,
                           int64_t last_seqno, swa_db** out);
SWA_API int swa_db_from_memory_translated(const uint8_t* nt_residues, const int64_t* offsets, int64_t nseq,
                                  int db_gencode, int device, int64_t first_seqno, int64_t total_seqcount,
                                  int64_t total_symcount, swa_db** out);
/* Databases larger than the device memory they may use (the reference maps any range of a database a chunk at a time,
   db_mapsequences, database.cc:1082-1131).  With hbm_budget_bytes below what the resident form needs, the shard stays
   in page-locked HOST memory, cut into parts of at most half the budget; every search walks the parts through two
   device slots - one is searched while the next part travels over PCIe and is formatted on the other (double
   buffering) - and merges the per-part candidates, so hit lists, counts and scores are those of the resident shard.
   The handle answers every search entry point (nucleotide parts carry the 4-bit one-sequence-per-row tables their both-strand
   searches run over) and - the owning part bound to a slot for the call - swa_search_endpoints[_strand], swa_align_hits and
   swa_db_sequence; swa_db_set_inclusion re-plans every part's tables for the admitted sequences (db_check_inclusion, database.cc:670-772).  hbm_budget_bytes <= 0 or large
   enough: an ordinary resident shard.
   The budget covers what the default searches use: the two slots with their residues, tables and first-pass stream.  What a
   part does not carry is built per part and search OUTSIDE it: the pair stream of a nucleotide part for a single-strand
   search, the 16-bit stream for a matrix that scores the padding symbol, window views of very long sequences, 64-bit
   score arrays.  swa_db_from_memory_streamed holds the database twice in host memory for a moment (the caller's arrays + the
   page-locked parts).  swa_db_open_streamed does not (round 6; the reference maps only what it is about to search,
   db_mapsequences, database.cc:1082-1131): it returns once the index is read and the parts are planned, and a loader thread
   fills the parts' page-locked blocks behind it, in the order the first search walks them, straight out of the sequence
   files (nucleotide entries unpacked and their ambiguity runs applied on its reader threads) - no copy of the database in
   ordinary host memory.  A search that arrives early binds each part when it is in place; swa_db_wait /
   swa_db_load_progress report the load; a load error is returned by the call that needs the part. */
SWA_API int swa_db_from_memory_streamed(const uint8_t* residues, const int64_t* offsets, int64_t nseq, int symtype, int device,
                                int64_t first_seqno, int64_t total_seqcount, int64_t total_symcount,
                                int64_t hbm_budget_bytes, swa_db** out);
SWA_API int swa_db_open_streamed(const char* basename, int symtype, int device, int64_t first_seqno, int64_t last_seqno,
                         int64_t hbm_budget_bytes, swa_db** out);
/* Host arithmetic only, no device: the layout swa_db_open_async would give a shard whose sequences have these lengths
   (offsets: nseq + 1 prefix sums; part_bytes as option load_part) - out[0] parts, [1] batches of the merged table, [2] its
   stream chunks, [3] the stream chunks ONE global length sort would need, [4] sequences missing / listed twice / batches
   mis-sized, [5] places where the merged table's batch lengths increase, [6] overlapping stream regions.  A diagnostic
   for the CPU tests: the loader itself only runs where there is a GPU. */
SWA_API int swa_debug_load_layout(const int64_t* offsets, int64_t nseq, int64_t part_bytes, int64_t* out);
SWA_API int swa_db_info(const swa_db* db, swa_db_info_t* info);
/* Host-only: read sequences [first_seqno, last_seqno] of a BLAST v4 database into malloc'ed
   arrays in reference symbol codes (what db_getsequence returns, database.cc:1237-1401:
   protein = NCBIstdaa bytes; nucleotide = 4-bit base masks with ambiguities applied).
   Release both arrays with swa_free.  Needs no GPU. */
SWA_API int swa_blastdb_read(const char* basename, int symtype, int64_t first_seqno, int64_t last_seqno,
                     uint8_t** residues, int64_t** offsets, int64_t* nseq,
                     int64_t* total_seqcount, int64_t* total_symcount, int64_t* longest);
SWA_API void swa_free(void* p);
/* Host-only: write sequences (reference symbol codes) as ONE BLAST v4 volume basename.{pin,psq,phr} /
   {nin,nsq,nhr} that the reference's db_open accepts (database.cc:566-601) - streaming, O(1) extra memory.  The image
   has no makeblastdb; the benchmark uses this to hand the same synthetic database to the reference CLI (CPU baseline)
   and to time a cold swa_db_open.  Definition lines render as "lcl|s<first_id + i> seq<first_id + i>".  Nucleotide
   input must be A/C/G/T masks (1, 2, 4, 8): ambiguity runs are not written (SWA_EINVAL).  At most 4 GiB per volume. */
SWA_API int swa_blastdb_write(const char* basename, int symtype, const uint8_t* residues, const int64_t* offsets,
                      int64_t nseq, int64_t first_id, const char* title);
/* Host-only: first definition line of sequence `seqno` rendered as the reference's db_showheader does
   for hit lists ("lcl|id title", asnparse.cc:753-887) and the sequence's length. */
SWA_API int swa_blastdb_defline(const char* basename, int symtype, int64_t seqno, char* buf, int64_t buflen,
                        int64_t* seqlen);
/* All definition lines of the entry (identical sequences merged by the formatter carry several), one per
   line, as the reference prints them above an alignment (hits.cc:1870, maxdeflines = LONG_MAX).
   *needed = bytes incl. NUL; SWA_ERANGE when buflen is smaller. */
SWA_API int swa_blastdb_deflines(const char* basename, int symtype, int64_t seqno, char* buf, int64_t buflen,
                         int64_t* needed);
/* Definition lines and inclusion filters of a database, host only (db_open's alias / mask / taxid handling,
   database.cc:670-772, 1403-1481; header rendering, asnparse.cc).  taxidfile may be NULL (-x, one taxid per
   line).  A MEMB_BIT alias with its OIDLIST mask is honoured automatically, as in the reference. */
typedef struct swa_headers swa_headers;
#define SWA_HEADERS_SHOW_GIS   1   /* -I: include gi|N ids */
#define SWA_HEADERS_SHOW_TAXID 2   /* -H: append |taxid|N|link|N|memb|N */
SWA_API int swa_headers_open(const char* basename, int symtype, const char* taxidfile, swa_headers** out);
SWA_API void swa_headers_close(swa_headers* h);
/* volume totals, the masked alias's NSEQ / LENGTH (equal to the totals when unmasked: what statistics and
   the "Database size" line use, hits.cc:333-342), longest sequence, title */
SWA_API int swa_headers_info(const swa_headers* h, int64_t* seqcount, int64_t* symcount, int64_t* masked_seqcount,
                     int64_t* masked_symcount, int64_t* longest, char* title, int64_t title_cap);
SWA_API int swa_headers_time(const swa_headers* h, char* buf, int64_t cap);   /* creation stamp of the first volume */
/* the definition lines of `seqno` that pass the membership / taxid filters, one per line; *needed = bytes
   incl. NUL, SWA_ERANGE when buflen is smaller */
SWA_API int swa_headers_get(const swa_headers* h, int64_t seqno, int flags, char* buf, int64_t buflen, int64_t* needed);
/* db_check_inclusion for sequences [first_seqno, first_seqno + n): include[i] = 1 if it is searched */
SWA_API int swa_headers_inclusion(const swa_headers* h, int64_t first_seqno, int64_t n, uint8_t* include);
/* Restrict a resident shard to a subset (include[i] != 0 for sequence first_seqno + i; NULL = all).  Excluded
   sequences are skipped by every search and report score -1.  swa_db_open applies a masked alias's OID mask
   by itself; a taxid list goes through swa_headers_inclusion + this call. */
SWA_API int swa_db_set_inclusion(swa_db* db, const uint8_t* include, int64_t n);
SWA_API void swa_db_close(swa_db* db);
/* Tuning and test knobs of one handle.  None of them changes a result; the defaults are the measured choices of
   DESIGN.md 4.9, so a SWIPE integration never needs this call - it exists for the A/B tools and so that the parity
   tests reach every kernel build.  key / value are text ("bound", "0"); value NULL restores the default.  Keys:
     bound            top-K searches: -1 auto, 0 exact first pass always, 1 bound build whenever one exists
     lanes            lanes per sequence pair of the first-pass kernel (1, 2, 4, 8, 16) if the query fits; 0 auto
     pipe             profile-load build of the first-pass kernel: -1 measured best, 0 / 1 / 2
     blocks_per_cu    persistent blocks per CU of the first-pass kernel; 0 = 8
     narrow_variant   1 plain (8.5-instruction) form, 2 row-shifted form, 0 auto
     force_mp, mp_k, mp_w, dual_mp, dual_kmax      block-synchronous multi-pass kernels and their shapes
     boundary_mb      cap of the pass hand-over buffer of long queries, MiB; -1 = from free memory
     wave_requeue     0: re-queued sequences always by the batch kernels; -1 auto
     requeue_host     1: the host reads the re-queue list between the passes (two more stream synchronisations)
     requeue_block    1: the device-driven re-queue takes a block of four waves per sequence for queries of at most 1 024 rows
                      (swa_requeue_block_kernel); 0 (default): one wave per sequence.  counters.requeue_form says which ran
     concat           bound builds on chains of lanes: sets of batches a chain works through back to back, without draining or
                      resetting its state in between; -1 = 16; 1 or 0 = every set on its own (the round-3 kernel)
     concat_tail      ... except the last concat_tail sets of the work queue, handed out one at a time; -1 = four per resident wave
     twin             bound builds at two waves per SIMD: 8-wave blocks holding the query profile twice, the second copy
                      renormalising the stored H values on the way; 0 = 4-wave blocks, one copy (the round-3 form)
     stream_reserve   bytes of a budgeted shard's device slot set aside for what does not scale with the part; -1 = 8 MiB
     requeue_follow   accepted and ignored (rounds 2-3 ran the re-queue kernel beside the first pass on a second stream; it now
                      always runs behind it in the same stream - no kernel waits for another kernel)
     pipelined, load_part, load_chunk, load_threads, load_delay_ms, load_trace      the pipelined open (swa_db_open_async),
                      read when the open begins: see there
     window           long database sequences are searched as overlapping windows and folded back (scores stay exact:
                      the overlap is the longest span a positive-scoring alignment can have): -1 auto, 0 never,
                      n > 0 every sequence longer than n
     window_step      distance between window starts; 0 = from the query length and scoring system
     long_lanes       0: the bound build's 2-, 4- and 8-lane chains stop at 48 rows per lane like the exact build's
                      (default 1: up to 62 rows, i.e. queries of 97..124 / 193..248 / 385..496 rows on half the lanes)
     endpoints_thread 1 ("thread"): one-thread 64-bit end-point kernel; 0 ("wave")
     watchdog_s       n > 0: a search whose stream has not drained after n seconds fails with SWA_ENODEV and the device's
                      control block (queue heads, re-queue counts) in swa_last_error() instead of blocking for ever; the
                      handle is unusable afterwards.  0 (default): block in hipStreamSynchronize
   A new handle takes its initial values from the environment variables SWA_<KEY> ONCE, at creation; the search path
   never reads the environment.  Unknown keys and unparsable values return SWA_EINVAL. */
SWA_API int swa_set_option(swa_db* db, const char* key, const char* value);

/* Diagnostics of the kernel selection (host arithmetic, no device): the build of the one-query first pass a search of a
   qlen-row query would run under a scoring system with highest matrix entry hi - G lanes per sequence pair, K rows per
   lane, the bound build or the exact one - i.e. the argmax over the measured table of every build (swa_kernel_rate:
   kernel GCUPS of build (bound, G, K) on MI355X, 0 = no such build; csrc/kernel_rates.inc).  G = 0: the
   query takes the multi-pass path.  lanes as the option of that name (0 = free); mean_len <= 0 = 325. */
SWA_API int swa_kernel_choice(int64_t qlen, int want_bound, int64_t hi, int64_t gapopenextend, int64_t gapextend, int64_t longest,
                      double mean_len, int lanes, int32_t* G, int32_t* K, int32_t* bound, int32_t* predicted_gcups);
SWA_API int swa_kernel_rate(int bound, int G, int K);
/* the same for the two-query kernels (swa_search2*, swa_search_pair_topk, frame pairs): nres = 16 nucleotide alphabet, 32 others */
SWA_API int swa_kernel_choice2(int nres, int64_t qlen, int want_bound, int64_t hi, int64_t gapopenextend, int64_t gapextend,
                       int64_t longest, double mean_len, int lanes, int32_t* G, int32_t* K, int32_t* bound, int32_t* predicted_gcups);
SWA_API int swa_kernel_rate2(int nres, int bound, int G, int K);

/* ---- scoring ------------------------------------------------------------------------------ */
/* matrix: 32*32 scores, index (db_symbol << 5) | query_symbol, as score_matrix_63
   (matrices.cc:583-590); gapopenextend = gapopen + gapextend (swipe.cc:1126). */
SWA_API int swa_set_scoring(swa_db* db, const int64_t* matrix, int64_t gapopenextend, int64_t gapextend);

/* ---- search --------------------------------------------------------------------------------- */
/* Exact Smith-Waterman score of `query` (reference symbol codes) against every sequence of the
   shard: scores[s - first_seqno]; for a translated shard 6 scores per sequence,
   scores[6 * (s - first_seqno) + 3 * dstrand + dframe].  `scores` may be NULL (bench: results stay on
   device). */
SWA_API int swa_search(swa_db* db, const uint8_t* query, int64_t qlen, int64_t* scores,
               swa_counters_t* counters);
/* The hits_enter loop on device: keeps the `keep` best (score desc, seqno desc) among
   minscore <= score <= maxscore; *totalhits counts scores >= minscore, *obvious counts scores
   > maxscore (hits.cc:174-178).  hits[] receives *nhits entries, already ordered.
   Only scores >= minscore are observable here, so when minscore is well above the spread of unrelated
   sequences the first pass may be the BOUND build of the kernel (6 instead of 7.5 instructions per
   cell pair; it yields an upper bound at most 15 x gapextend above the score) and every sequence whose
   bound reaches minscore is recomputed exactly before the list is made: hits, *totalhits and *obvious
   are identical to the exact pass's (counters.narrow_shifted = 8; swa_set_option(db, "bound", "0")
   disables it).  First pass, re-queue kernels and the filter are enqueued back to back; the call synchronises with
   the device once. */
SWA_API int swa_search_topk(swa_db* db, const uint8_t* query, int64_t qlen, int64_t keep,
                    int64_t minscore, int64_t maxscore, swa_hit_t* hits, int64_t* nhits,
                    int64_t* totalhits, int64_t* obvious, swa_counters_t* counters);
/* Two queries of EQUAL length in one pass over the shard (both halves of the packed lanes see
   the same database residue).  The nucleotide search of the reference is exactly this: the plus
   strand and the reverse-complemented query against the same database (swipe.cc:1403-1411,
   hits entered with dstrand = 1 for the second, swipe.cc:1470-1471). */
SWA_API int swa_search2(swa_db* db, const uint8_t* query1, const uint8_t* query2, int64_t qlen,
                int64_t* scores1, int64_t* scores2, swa_counters_t* counters);
/* hits of both queries in one list; which[i] = 0 / 1 tells the query (strand) of hits[i]; on equal
   score and seqno the entry of query 1 comes first, as in the reference's insertion order */
SWA_API int swa_search2_topk(swa_db* db, const uint8_t* query1, const uint8_t* query2, int64_t qlen,
                     int64_t keep, int64_t minscore, int64_t maxscore, swa_hit_t* hits,
                     int32_t* which, int64_t* nhits, int64_t* totalhits, int64_t* obvious,
                     swa_counters_t* counters);
/* Two DIFFERENT queries in one pass over the shard - the unit of work of the reference is a query FILE
   (swipe.cc:2561-2575), and two queries in the two halves of every packed lane cost 5 (bound build) / 6.5 instructions
   per cell pair where one query and two sequences cost 6 / 7.5.  qlen1 and qlen2 may differ: the shorter query is
   padded with rows that score -1 against everything, so pairing pays when the lengths are within about a quarter of
   each other.  Each query has its own list length, score window, hit list and counts; every result equals what two
   swa_search_topk calls return.  counters describe the shared first pass (cells = symcount x (qlen1 + qlen2)). */
SWA_API int swa_search_pair_topk(swa_db* db, const uint8_t* query1, int64_t qlen1, const uint8_t* query2, int64_t qlen2,
                         int64_t keep1, int64_t minscore1, int64_t maxscore1, int64_t keep2, int64_t minscore2,
                         int64_t maxscore2, swa_hit_t* hits1, int64_t* nhits1, int64_t* totalhits1, int64_t* obvious1,
                         swa_hit_t* hits2, int64_t* nhits2, int64_t* totalhits2, int64_t* obvious2,
                         swa_counters_t* counters);
/* The general form behind every -p mode: nq query frames (1 for -p 1/3, 3 or 6 for -p 2/4; also the two
   strands of -p 0) against every frame the shard holds, merged into ONE hit list exactly as the
   reference's search_chunk + hits_enter leave it (swipe.cc:1403-1592, hits.cc:163-222): score descending,
   sequence number descending, then query frame (position in `queries`) and database frame ascending - the
   reference's insertion order for equal (score, seqno).  qtags[i] = 3 * qstrand + qframe is copied into the
   hits of query i.  Frames of length 0 are legal. */
typedef struct { int64_t seqno, score; int32_t qstrand, qframe, dstrand, dframe; } swa_fhit_t;
SWA_API int swa_search_frames_topk(swa_db* db, int nq, const uint8_t* const* queries, const int64_t* qlens,
                           const int32_t* qtags, int64_t keep, int64_t minscore, int64_t maxscore,
                           swa_fhit_t* hits, int64_t* nhits, int64_t* totalhits, int64_t* obvious,
                           swa_counters_t* counters);

/* Alignment-phase end points (search16s, swipe.h:237-249; called from align_chunk, swipe.cc:381): for
   each listed sequence the exact score, the 0-based database column where the final maximum is first
   reached and the smallest query row holding it in that column (search16s.cc:391-405).  Values equal
   the reference's whenever its 16-bit lanes do not saturate (score < SCORELIMIT_16, the only case in
   which it uses them, swipe.cc:404). */
SWA_API int swa_search_endpoints(swa_db* db, const uint8_t* query, int64_t qlen, const int64_t* seqnos, int64_t n,
                         int64_t* scores, int64_t* bestpos, int64_t* bestq);
/* Same for (sequence, database strand, database frame) triples, the reference's packed start_list entries
   (seqno << 3) | (dstrand << 2) | dframe (swipe.cc:359-362).  Nucleotide shard: dstrands[i] = 1 takes the
   reverse complement, which is how the reference aligns minus-strand hits - plus query against
   db_getsequence(seqno, strand 1) (database.cc:1327-1339).  Translated shard: (dstrand, dframe) picks one
   of the six translations.  dstrands / dframes may be NULL (= 0). */
SWA_API int swa_search_endpoints_strand(swa_db* db, const uint8_t* query, int64_t qlen, const int64_t* seqnos,
                         const int32_t* dstrands, const int32_t* dframes, int64_t n, int64_t* scores,
                         int64_t* bestpos, int64_t* bestq);
/* db_getsequence (swipe.h:344, database.cc:1237) out of the resident shard: residues of one sequence in
   reference symbol codes - reverse-complemented for dstrand 1 of a nucleotide shard, the chosen translation
   of a translated shard (*ntlen = its nucleotide length, else 0; may be NULL).  *len is set even when cap
   is too small (the call then returns SWA_ERANGE). */
SWA_API int swa_db_sequence(swa_db* db, int64_t seqno, int dstrand, int dframe, uint8_t* buf, int64_t cap,
                    int64_t* len, int64_t* ntlen);

/* The alignment phase for a list of hits: align_chunk + hits_align + align (swipe.cc:339-414,
   hits.cc:546-618, align.cc:469-519).  End points come from the GPU (search16s semantics), the start
   point and the edit script from the host part (traceback.cpp, linear space).  Coordinates are 0-based
   and inclusive, in the frame the reference aligns in: the query as given against the database
   sequence, reverse-complemented when dstrands[i] = 1.  The edit script is the reference's alignment
   string: runs "M<n>" (column pair), "D<n>" (query symbols against a gap), "I<n>" (database symbols
   against a gap), e.g. "M120D2M31I1M7".  Counts follow count_align (hits.cc:1021-1109). */
typedef struct {
  int64_t seqno;
  int32_t dstrand, dframe;
  int32_t hinted;                 /* 1: the search16s end point was used (hits.cc:587), 0: forward sweep */
  int32_t reserved;
  int64_t score;
  int64_t q_start, q_end, d_start, d_end;
  int64_t dlen;                   /* length of the sequence aligned against (a translation for translated shards) */
  int64_t dlennt;                 /* its nucleotide length for translated shards (hits.cc:567), else 0 */
  int64_t identities, positives, indels, aligned, gaps;
  int64_t cigar_offset, cigar_len; /* edit script = text[cigar_offset .. +cigar_len), NUL-terminated */
} swa_alignment_t;
/* text receives all edit scripts back to back; *text_used = bytes needed.  SWA_ERANGE if text_cap is
   smaller (out[] is complete even then; call again with a larger buffer for the scripts). */
SWA_API int swa_align_hits(swa_db* db, const uint8_t* query, int64_t qlen, const int64_t* seqnos,
                   const int32_t* dstrands, const int32_t* dframes, int64_t n, swa_alignment_t* out, char* text,
                   int64_t text_cap, int64_t* text_used);

/* The host part alone, for one sequence the caller holds (e.g. rank 0 finishing hits of other shards):
   align() as hits_align calls it.  M as for swa_set_scoring; gapopen/gapextend are the -G/-E values
   (NOT open+extend).  hint_score != 0: (hint_q_end, hint_d_end) is trusted as the end cell and
   hint_score as the score (what swa_search_endpoints returns, subject to the hits.cc:587 rule which the
   caller applies); hint_score == 0: the forward sweep finds both.  Pure host arithmetic, no device. */
SWA_API int swa_traceback(const uint8_t* query, int64_t qlen, const uint8_t* dseq, int64_t dlen, const int64_t* M,
                  int64_t gapopen, int64_t gapextend, int64_t hint_score, int64_t hint_q_end, int64_t hint_d_end,
                  swa_alignment_t* out, char* text, int64_t text_cap, int64_t* text_used);

/* Merge per-shard top-K lists (each ordered) into the global top-K with the reference's
   comparator - what the MPI master does with tag_search_report (swipe.cc:1951-1974). */
SWA_API int swa_hits_merge(const swa_hit_t* lists, const int64_t* counts, int nlists, int64_t stride,
                   int64_t keep, swa_hit_t* out, int64_t* nout);

/* Same for the frame-tagged lists of swa_search_frames_topk (each shard's list ordered, shards disjoint). */
SWA_API int swa_fhits_merge(const swa_fhit_t* lists, const int64_t* counts, int nlists, int64_t stride,
                    int64_t keep, swa_fhit_t* out, int64_t* nout);

/* ---- several devices behind one handle ------------------------------------------------------------------------
   The reference parallelises INSIDE the binary: run_threads / worker (swipe.cc:1599-1699) hand chunks of sequence
   numbers to -a N pthreads that all enter one hit list under hitsmutex, and mpiswipe's master farms chunks out and
   re-enters the workers' reported hits (swipe.cc:1812-2160, tag_search_report 1951-1974, 2320).  A swa_group is that
   layer for N devices: the database is cut into N contiguous, residue-balanced shards (swa_shard_bounds), shard i
   lives on devices[i] and is served by ONE host thread of its own for the whole life of the group (the "one host
   thread per device handle" rule above); a search runs on all shards at once, every shard reduces to its own top-K
   on its device, and the N short lists are merged on the host with the reference comparator (swa_hits_merge /
   swa_fhits_merge) - provably the single-list result because the reference's list is exactly the global top-K under
   (score desc, seqno desc) (hits.cc:188-219).  4 KB per device and search cross the host: an in-process group needs
   no RCCL; the RCCL gather belongs to the one-process-per-GPU launch (swipe_amd/parallel.py, bench.py --gpus N).
   devices may name one device several times (two shards on one GPU: how the 1-GPU test box exercises this layer).
   Shards that would be empty (more shards than sequences) are not created.  Every call below is the group form of
   the swa_* call of the same name and returns what that call returns on one shard holding the whole database;
   counters are summed over the shards, the two times are the slowest shard's. */
typedef struct swa_group swa_group;
/* cuts[0..nshards]: shard r = sequences [cuts[r], cuts[r+1]) of a database whose sequence s starts at residue
   offsets[s] (nseq + 1 prefix sums, offsets[0] may be non-zero): the first sequence at or beyond each r/nshards of
   the residues.  swa_blastdb_shard_bounds reads the lengths from the index files of a BLAST v4 database. */
SWA_API int swa_shard_bounds(const int64_t* offsets, int64_t nseq, int nshards, int64_t* cuts);
SWA_API int swa_blastdb_shard_bounds(const char* basename, int symtype, int nshards, int64_t* cuts);
/* db_gencode = 0: the database as it is (symtype 0 / 1); 1..23: a nucleotide database held as its six translations
   (swa_db_open_translated).  devices[i] = HIP device of shard i. */
SWA_API int swa_group_open(const char* basename, int symtype, int db_gencode, int nshards, const int* devices, swa_group** out);
/* every shard with an HBM budget of its own (swa_db_open_streamed: the reference maps any range of a database a chunk at a
   time with any thread count, database.cc:1082-1131); hbm_budget_bytes is PER DEVICE.  Searches, end points, alignments and
   sequence fetches and inclusion masks (swa_group_set_inclusion, masked aliases) work as on resident shards. */
SWA_API int swa_group_open_streamed(const char* basename, int symtype, int nshards, const int* devices, int64_t hbm_budget_bytes,
                            swa_group** out);
SWA_API int swa_group_from_memory(const uint8_t* residues, const int64_t* offsets, int64_t nseq, int symtype, int db_gencode,
                          int nshards, const int* devices, int64_t first_seqno, int64_t total_seqcount,
                          int64_t total_symcount, swa_group** out);
SWA_API void swa_group_close(swa_group* g);
/* swa_group_open returns as soon as every shard's index is read; the residues stream in behind it (swa_db_open_async per
   shard) and a search follows the loaders.  swa_group_wait blocks until every shard is resident and returns the first load
   error (truncated file, residue code out of range, ...); swa_group_load_progress sums swa_db_load_progress over the
   shards.  Mirrors: db_open + db_mapsequences over all volumes, database.cc:1082-1131. */
SWA_API int swa_group_wait(swa_group* g);
SWA_API int swa_group_load_progress(swa_group* g, int64_t* bytes_loaded, int64_t* bytes_total, int32_t* parts_ready, int32_t* parts_total);
/* *nshards = shards actually created; info describes the union of the shards (hbm_bytes summed) */
SWA_API int swa_group_info(const swa_group* g, swa_db_info_t* info, int* nshards);
/* shard i's own handle (borrowed; for swa_db_info and the A/B tools - searches go through the group) */
SWA_API int swa_group_shard(const swa_group* g, int i, swa_db** db);
SWA_API int swa_group_set_scoring(swa_group* g, const int64_t* matrix, int64_t gapopenextend, int64_t gapextend);
SWA_API int swa_group_set_option(swa_group* g, const char* key, const char* value);
/* include[i] for sequence first_seqno + i of the whole range the group holds (NULL = all) */
SWA_API int swa_group_set_inclusion(swa_group* g, const uint8_t* include, int64_t n);
SWA_API int swa_group_search(swa_group* g, const uint8_t* query, int64_t qlen, int64_t* scores, swa_counters_t* counters);
SWA_API int swa_group_search_topk(swa_group* g, const uint8_t* query, int64_t qlen, int64_t keep, int64_t minscore,
                          int64_t maxscore, swa_hit_t* hits, int64_t* nhits, int64_t* totalhits, int64_t* obvious,
                          swa_counters_t* counters);
SWA_API int swa_group_search_pair_topk(swa_group* g, const uint8_t* query1, int64_t qlen1, const uint8_t* query2, int64_t qlen2,
                               int64_t keep1, int64_t minscore1, int64_t maxscore1, int64_t keep2, int64_t minscore2,
                               int64_t maxscore2, swa_hit_t* hits1, int64_t* nhits1, int64_t* totalhits1, int64_t* obvious1,
                               swa_hit_t* hits2, int64_t* nhits2, int64_t* totalhits2, int64_t* obvious2,
                               swa_counters_t* counters);
SWA_API int swa_group_search_frames_topk(swa_group* g, int nq, const uint8_t* const* queries, const int64_t* qlens,
                                 const int32_t* qtags, int64_t keep, int64_t minscore, int64_t maxscore,
                                 swa_fhit_t* hits, int64_t* nhits, int64_t* totalhits, int64_t* obvious,
                                 swa_counters_t* counters);
/* the alignment phase: every hit is aligned by the shard that holds its sequence (align_chunk on the worker that
   owns the chunk, swipe.cc:339-414), all shards at once; out[] / text in the order of the hit list */
SWA_API int swa_group_align_hits(swa_group* g, const uint8_t* query, int64_t qlen, const int64_t* seqnos,
                         const int32_t* dstrands, const int32_t* dframes, int64_t n, swa_alignment_t* out, char* text,
                         int64_t text_cap, int64_t* text_used);
SWA_API int swa_group_db_sequence(swa_group* g, int64_t seqno, int dstrand, int dframe, uint8_t* buf, int64_t cap,
                          int64_t* len, int64_t* ntlen);

/* ---- statistics (host arithmetic, bit-exact with hits.cc/stats.cc) ------------------------- */
typedef struct {
  int available;                       /* 0: no K-A parameters for this scoring system */
  double lambda, K, H, alpha, beta;
  double Kmn, logK, lambda_d_log2, logK_d_log2;
  int64_t lenadj, m, n;
  int64_t scorethreshold, upperscorethreshold;   /* after the E-value cut, hits.cc:486-508 */
} swa_stats_t;
/* symtype 0..4 as the reference's -p; for the translated types pass NUCLEOTIDE counts where the query /
   database is nucleotide (qlen = query.nt[0].len, db_symcount = bases), as hits_init reads them. */
SWA_API int swa_stats_init(int symtype, const char* matrixname, int64_t match, int64_t mismatch,
                   int64_t gapopen, int64_t gapextend, int64_t qlen,
                   int64_t db_seqcount, int64_t db_symcount, int64_t effdbsize,
                   int64_t minscore, int64_t maxscore, double minexpect, double expect,
                   swa_stats_t* out);
SWA_API double swa_evalue(const swa_stats_t* st, int64_t score);   /* hits.cc:1777 */
SWA_API double swa_bits(const swa_stats_t* st, int64_t score);     /* hits.cc:1779 */
/* Genetic codes 1..23 (query.cc:118-170).  swa_translate_table fills table[4096], indexed 256a + 16b + c
   by the three IUPAC nibbles of a codon, with NCBIstdaa codes (translate_createtable, query.cc:377-455);
   swa_translate is translate() / db_translate() (query.cc:463-506): frame 0..2 of strand 0/1, *plen =
   (dlen - frame) / 3 residues into prot.  Host helpers for queries; databases are translated on the GPU. */
SWA_API const char* swa_gencode_name(int gencode);   /* NULL for an unassigned number */
SWA_API int swa_translate_table(int gencode, uint8_t* table);
SWA_API int swa_translate(const uint8_t* dna, int64_t dlen, int strand, int frame, const uint8_t* table,
                  uint8_t* prot, int64_t* plen);

/* Built-in matrices by name (matrices.cc:540-559); returns SWA_EINVAL for unknown names. */
SWA_API int swa_matrix_builtin(const char* name, int64_t* matrix);
SWA_API int swa_matrix_nucleotide(int64_t match, int64_t mismatch, int64_t* matrix);
SWA_API int swa_matrix_parse(const char* text, int64_t* matrix);
SWA_API int swa_default_gaps(const char* matrixname, int64_t* gapopen, int64_t* gapextend);

#ifdef __cplusplus
}
#endif
#endif
