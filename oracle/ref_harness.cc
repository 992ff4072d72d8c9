/* TEST INFRASTRUCTURE - not part of the product.

   Raw-score harness around the *compiled reference*: links the reference's own object
   files (built from /root/reference by oracle/Makefile into oracle/_ref/) and calls its
   DP kernels directly, printing what each lane width returns for EVERY database
   sequence -- including the saturated values the CLI never shows.  Used to pin the C
   restatement in sw_oracle.c and to generate tests/golden/*.raw.tsv.

   It re-creates only the argument plumbing of the reference's search_init
   (swipe.cc:1183-1251: dprofile, hearray, qtable[i] = dprofile + 64*qsym; 16*qsym for
   search16s as in align_init, swipe.cc:202-240) and calls
   search7 / search7_ssse3 / search16 / fullsw exactly as search_chunk does
   (swipe.cc:1432-1585).  Nothing here is shipped or timed.

   usage: ref_harness <db> <query.fasta> <symtype 0|1> <matrix|-> <gapopen> <gapextend>
                      [match mismatch [align]]
   output (one line per db sequence and query strand):
     seqno strand len s7_ssse3 s7_sse2 s16 bestpos16 s63 s16s bestpos16s bestq16s
   with the trailing word "align", additionally one line per (sequence, database strand) whose
   score is positive, from the reference's alignment phase run as align_chunk/hits_align do
   (swipe.cc:339-414, hits.cc:546-618): search16s end points of the PLUS query against the
   (reverse-complemented for strand 1) database sequence, then align() without and with them:
     A seqno dstrand s16s bestpos bestq | score qstart dstart qend dend cigar | same with hints or "-"

   symtype 2/3/4 (translated query / database / both; two more optional arguments: query and
   database genetic code): every (query frame, database frame) combination the reference searches
   (swipe.cc:289-324, 1375-1389), frames obtained from the reference's own translate() /
   db_getsequence():
     T seqno qtag dtag len s7_ssse3 s7_sse2 s16 bestpos16 s63 s16s bestpos16s bestq16s
   with qtag = 3*qstrand+qframe, dtag = 3*dstrand+dframe, and with "align" the A lines carry
   "seqno qtag dtag" in place of "seqno dstrand".
*/
#include "swipe.h"   /* found via -I/root/reference */

extern long query_gencode, db_gencode;      /* swipe.cc:80-81 (not in swipe.h) */
void translate_init(long qtableno, long dtableno);   /* query.cc:457 */

/* symtype 2/3/4: all frame combinations, kernels called as search_chunk / align_chunk call them */
static int translated_main(int with_align)
{
  struct db_thread_s* dbt = db_thread_create();
  struct db_thread_s* dbta[8];
  for (int i = 0; i < 8; i++) dbta[i] = db_thread_create();
  BYTE* dprofile = (BYTE*) xmalloc(4 * 16 * 32);
  const int qtags = (symtype == 3) ? 1 : 6, dtags = (symtype == 2) ? 1 : 6;
  printf("# SCORELIMIT_7=%ld SCORELIMIT_16=%ld gapopenextend=%ld gapextend=%ld\n",
         SCORELIMIT_7, SCORELIMIT_16, gapopenextend, gapextend);
  long seqbase = 0;
  for (long vol = 0; vol < db_getvolumecount(); vol++) {
    long n = db_getseqcount_volume(vol);
    if (n == 0) continue;
    long cnt = n * dtags;
    long* seqnos = (long*) xmalloc(cnt * sizeof(long));
    long *s7a = (long*) xmalloc(cnt * sizeof(long)), *s7b = (long*) xmalloc(cnt * sizeof(long));
    long *s16 = (long*) xmalloc(cnt * sizeof(long)), *bp16 = (long*) xmalloc(cnt * sizeof(long));
    long *s16s = (long*) xmalloc(cnt * sizeof(long)), *bp16s = (long*) xmalloc(cnt * sizeof(long)), *bq16s = (long*) xmalloc(cnt * sizeof(long));
    for (long i = 0; i < n; i++)
      for (int d = 0; d < dtags; d++) seqnos[i * dtags + d] = ((seqbase + i) << 3) | ((d / 3) << 2) | (d % 3);
    for (int qt = 0; qt < qtags; qt++) {
      char* q = query.aa[qt].seq;
      long qlen = query.aa[qt].len;
      BYTE** qtable = (BYTE**) xmalloc((qlen > 0 ? qlen : 1) * sizeof(BYTE*));
      BYTE** qtable_s = (BYTE**) xmalloc((qlen > 0 ? qlen : 1) * sizeof(BYTE*));
      for (long i = 0; i < qlen; i++) { qtable[i] = dprofile + 64 * q[i]; qtable_s[i] = dprofile + 16 * q[i]; }
      BYTE* hearray = (BYTE*) xmalloc((qlen > 0 ? qlen : 1) * 32);
      db_mapsequences(dbt, seqbase, seqbase + n - 1);
      search7_ssse3(qtable, gapopenextend, gapextend, (BYTE*) score_matrix_7t, dprofile, hearray, dbt, cnt, seqnos, s7a, qlen);
      search7(qtable, gapopenextend, gapextend, (BYTE*) score_matrix_7, dprofile, hearray, dbt, cnt, seqnos, s7b, qlen);
      search16((WORD**) qtable, gapopenextend, gapextend, (WORD*) score_matrix_16, (WORD*) dprofile, (WORD*) hearray, dbt, cnt,
               seqnos, s16, bp16, qlen);
      search16s((WORD**) qtable_s, gapopenextend, gapextend, (WORD*) score_matrix_16, (WORD*) dprofile, (WORD*) hearray, dbta,
                cnt, seqnos, s16s, bp16s, bq16s, qlen);
      db_mapsequences(dbt, seqbase, seqbase + n - 1);
      for (long k = 0; k < cnt; k++) {
        long seqno = seqnos[k] >> 3, ds = (seqnos[k] >> 2) & 1, df = seqnos[k] & 3;
        char* address; long length, ntlen;
        db_getsequence(dbt, seqno, ds, df, &address, &length, &ntlen, 0);
        long* he63 = (long*) xmalloc((qlen > 0 ? qlen : 1) * 2 * sizeof(long));
        long s63 = fullsw(address, address + length - 1, q, q + qlen, he63, score_matrix_63, gapopenextend, gapextend);
        free(he63);
        printf("T\t%ld\t%d\t%ld\t%ld\t%ld\t%ld\t%ld\t%ld\t%ld\t%ld\t%ld\t%ld\n", seqno, qt, 3 * ds + df, length - 1,
               s7a[k], s7b[k], s16[k], bp16[k], s63, s16s[k], bp16s[k], bq16s[k]);
        if (with_align && s16s[k] > 0 && qlen > 0 && length > 1) {
          long qs = 0, dst = 0, qe = 0, de = 0, sc = 0;
          char* aln = 0;
          align(q, address, qlen, length - 1, score_matrix_63, gapopen, gapextend, &qs, &dst, &qe, &de, &aln, &sc);
          printf("A\t%ld\t%d\t%ld\t%ld\t%ld\t%ld\t|\t%ld\t%ld\t%ld\t%ld\t%ld\t%s\t|", seqno, qt, 3 * ds + df, s16s[k], bp16s[k],
                 bq16s[k], sc, qs, dst, qe, de, aln);
          free(aln);
          if (s16s[k] < SCORELIMIT_16 && bq16s[k] > 0 && bp16s[k] != 0) {
            qs = dst = 0; qe = bq16s[k]; de = bp16s[k]; sc = s16s[k];
            align(q, address, qlen, length - 1, score_matrix_63, gapopen, gapextend, &qs, &dst, &qe, &de, &aln, &sc);
            printf("\t%ld\t%ld\t%ld\t%ld\t%ld\t%s\n", sc, qs, dst, qe, de, aln);
            free(aln);
          } else {
            printf("\t-\n");
          }
        }
      }
      free(qtable); free(qtable_s); free(hearray);
    }
    free(seqnos); free(s7a); free(s7b); free(s16); free(bp16); free(s16s); free(bp16s); free(bq16s);
    seqbase += n;
  }
  return 0;
}

int main(int argc, char** argv)
{
  if (argc < 7) {
    fprintf(stderr, "usage: %s db query symtype matrix gapopen gapextend [match mismatch]\n", argv[0]);
    return 2;
  }
  databasename = argv[1];
  queryname = argv[2];
  symtype = atol(argv[3]);
  matrixname = (argv[4][0] == '-' && !argv[4][1]) ? "BLOSUM62" : argv[4];
  gapopen = atol(argv[5]);
  gapextend = atol(argv[6]);
  matchscore = argc > 7 ? atol(argv[7]) : 1;
  mismatchscore = argc > 8 ? atol(argv[8]) : -3;
  gapopenextend = gapopen + gapextend;
  querystrands = 3;
  cpu_feature_ssse3 = 1;
  query_gencode = argc > 10 ? atol(argv[10]) : 1;
  db_gencode = argc > 11 ? atol(argv[11]) : 1;
  translate_init(query_gencode, db_gencode);

  db_open(symtype, databasename, 0);
  score_matrix_init();
  query_init(queryname, symtype, querystrands);
  if (!query_read()) { fprintf(stderr, "no query\n"); return 1; }
  if (symtype >= 2) return translated_main(argc > 9 && !strcmp(argv[9], "align"));

  struct db_thread_s* dbt = db_thread_create();
  struct db_thread_s* dbta[8];
  for (int i = 0; i < 8; i++) dbta[i] = db_thread_create();
  BYTE* dprofile = (BYTE*) xmalloc(4 * 16 * 32);

  int nstrands = symtype == 0 ? 2 : 1;
  printf("# SCORELIMIT_7=%ld SCORELIMIT_16=%ld gapopenextend=%ld gapextend=%ld\n",
         SCORELIMIT_7, SCORELIMIT_16, gapopenextend, gapextend);

  long seqbase = 0;
  for (long vol = 0; vol < db_getvolumecount(); vol++) {
    long n = db_getseqcount_volume(vol);
    if (n == 0) continue;
    db_mapsequences(dbt, seqbase, seqbase + n - 1);
    long* seqnos = (long*) xmalloc(n * sizeof(long));
    long* s7a = (long*) xmalloc(n * sizeof(long));
    long* s7b = (long*) xmalloc(n * sizeof(long));
    long* s16 = (long*) xmalloc(n * sizeof(long));
    long* bp16 = (long*) xmalloc(n * sizeof(long));
    long* s16s = (long*) xmalloc(n * sizeof(long));
    long* bp16s = (long*) xmalloc(n * sizeof(long));
    long* bq16s = (long*) xmalloc(n * sizeof(long));
    for (long i = 0; i < n; i++) seqnos[i] = (seqbase + i) << 3;

    for (int s = 0; s < nstrands; s++) {
      char* q = symtype == 0 ? query.nt[s].seq : query.aa[0].seq;
      long qlen = symtype == 0 ? query.nt[s].len : query.aa[0].len;
      BYTE** qtable = (BYTE**) xmalloc((qlen > 0 ? qlen : 1) * sizeof(BYTE*));
      for (long i = 0; i < qlen; i++) qtable[i] = dprofile + 64 * q[i];
      BYTE* hearray = (BYTE*) xmalloc((qlen > 0 ? qlen : 1) * 32);

      search7_ssse3(qtable, gapopenextend, gapextend, (BYTE*) score_matrix_7t, dprofile, hearray,
                    dbt, n, seqnos, s7a, qlen);
      search7(qtable, gapopenextend, gapextend, (BYTE*) score_matrix_7, dprofile, hearray,
              dbt, n, seqnos, s7b, qlen);
      search16((WORD**) qtable, gapopenextend, gapextend, (WORD*) score_matrix_16, (WORD*) dprofile,
               (WORD*) hearray, dbt, n, seqnos, s16, bp16, qlen);

      /* alignment phase layout: one column per block, qtable stride 16 (align_init, swipe.cc:220-240) */
      BYTE** qtable_s = (BYTE**) xmalloc((qlen > 0 ? qlen : 1) * sizeof(BYTE*));
      for (long i = 0; i < qlen; i++) qtable_s[i] = dprofile + 16 * q[i];
      search16s((WORD**) qtable_s, gapopenextend, gapextend, (WORD*) score_matrix_16, (WORD*) dprofile,
                (WORD*) hearray, dbta, n, seqnos, s16s, bp16s, bq16s, qlen);
      free(qtable_s);
      db_mapsequences(dbt, seqbase, seqbase + n - 1);

      for (long i = 0; i < n; i++) {
        char* address; long length, ntlen;
        db_getsequence(dbt, seqbase + i, 0, 0, &address, &length, &ntlen, 0);
        long* he63 = (long*) xmalloc((qlen > 0 ? qlen : 1) * 2 * sizeof(long));
        long s63 = fullsw(address, address + length - 1, q, q + qlen, he63, score_matrix_63,
                          gapopenextend, gapextend);
        free(he63);
        printf("%ld\t%d\t%ld\t%ld\t%ld\t%ld\t%ld\t%ld\t%ld\t%ld\t%ld\n", seqbase + i, s, length - 1,
               s7a[i], s7b[i], s16[i], bp16[i], s63, s16s[i], bp16s[i], bq16s[i]);
      }
      free(qtable);
      free(hearray);
    }
    if (argc > 9 && !strcmp(argv[9], "align")) {
      char* q = symtype == 0 ? query.nt[0].seq : query.aa[0].seq;
      long qlen = symtype == 0 ? query.nt[0].len : query.aa[0].len;
      BYTE** qtable_s = (BYTE**) xmalloc((qlen > 0 ? qlen : 1) * sizeof(BYTE*));
      for (long i = 0; i < qlen; i++) qtable_s[i] = dprofile + 16 * q[i];
      BYTE* hearray = (BYTE*) xmalloc((qlen > 0 ? qlen : 1) * 32);
      for (int ds = 0; ds < nstrands; ds++) {
        for (long i = 0; i < n; i++) seqnos[i] = ((seqbase + i) << 3) | (ds << 2);
        search16s((WORD**) qtable_s, gapopenextend, gapextend, (WORD*) score_matrix_16, (WORD*) dprofile,
                  (WORD*) hearray, dbta, n, seqnos, s16s, bp16s, bq16s, qlen);
        db_mapsequences(dbt, seqbase, seqbase + n - 1);
        for (long i = 0; i < n; i++) {
          char* address; long length, ntlen;
          db_getsequence(dbt, seqbase + i, ds, 0, &address, &length, &ntlen, 0);
          long dlen = length - 1;
          if (s16s[i] <= 0 || qlen == 0 || dlen == 0) continue;
          long qs = 0, dst = 0, qe = 0, de = 0, sc = 0;
          char* aln = 0;
          align(q, address, qlen, dlen, score_matrix_63, gapopen, gapextend, &qs, &dst, &qe, &de, &aln, &sc);
          printf("A\t%ld\t%d\t%ld\t%ld\t%ld\t|\t%ld\t%ld\t%ld\t%ld\t%ld\t%s\t|", seqbase + i, ds, s16s[i], bp16s[i], bq16s[i],
                 sc, qs, dst, qe, de, aln);
          free(aln);
          if (s16s[i] < SCORELIMIT_16 && bq16s[i] > 0 && bp16s[i] != 0) {      /* hits.cc:587 */
            qs = dst = 0; qe = bq16s[i]; de = bp16s[i]; sc = s16s[i];
            align(q, address, qlen, dlen, score_matrix_63, gapopen, gapextend, &qs, &dst, &qe, &de, &aln, &sc);
            printf("\t%ld\t%ld\t%ld\t%ld\t%ld\t%s\n", sc, qs, dst, qe, de, aln);
            free(aln);
          } else {
            printf("\t-\n");
          }
        }
      }
      free(qtable_s);
      free(hearray);
    }
    free(seqnos); free(s7a); free(s7b); free(s16); free(bp16); free(s16s); free(bp16s); free(bq16s);
    seqbase += n;
  }
  return 0;
}
