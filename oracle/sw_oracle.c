/* TEST INFRASTRUCTURE - see sw_oracle.h.  Plain C11 restatement of the reference hot path;
   every function cites the reference lines (file:line into /root/reference) it follows. */
#define _GNU_SOURCE
#include "sw_oracle.h"
#include "refdata.h"

#include <ctype.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>

/* NCBIstdaa letter -> code, reference query.cc:51-69 (map_ncbi_aa); -1 for anything else */
static int aa_code(int c)
{
  static const char order[] = "-ABCDEFGHIKLMNPQRSTVWXYZU*OJ";
  if (c == 0) return -1;
  const char* p = strchr(order, toupper(c));
  return p ? (int)(p - order) : -1;
}

/* ------------------------------------------------------------------ matrices ---------- */

void swo_matrix_clear(long* M)
{
  for (int i = 0; i < 1024; i++) M[i] = -1;          /* memset(..., -1, ...) matrices.cc:531 */
}

int swo_matrix_builtin(const char* name, long* M)
{
  swo_matrix_clear(M);
  for (int k = 0; k < REFDATA_NMATRICES; k++)
    if (strcasecmp(name, refdata_matrix_names[k]) == 0) {
      for (int a = 0; a < 28; a++)
        for (int b = 0; b < 28; b++)
          M[(a << 5) + b] = refdata_matrices[k][a][b];
      return 1;
    }
  return 0;
}

void swo_matrix_nucleotide(long match, long mismatch, long* M)
{
  swo_matrix_clear(M);
  for (int a = 1; a < 16; a++)                       /* matrices.cc:533-538 */
    for (int b = 1; b < 16; b++)
      M[(a << 5) + b] = (a == b) ? match : mismatch;
}

/* The text form accepted by score_matrix_read_file/_string (matrices.cc:352-517):
   '#' and blank lines skipped; a line starting with blank/tab lists the column symbols;
   every other line is "<row symbol> v v v ...", stored at [(row<<5)+col]. */
int swo_matrix_parse(const char* text, long* M)
{
  int order[256];
  int symbols = 0;
  swo_matrix_clear(M);
  const char* s = text;
  while (*s) {
    const char* eol = strchr(s, '\n');
    size_t len = eol ? (size_t)(eol - s) : strlen(s);
    if (len > 0 && s[0] != '#') {
      if (s[0] == ' ' || s[0] == '\t') {
        int k = 0;                                    /* q = order restarts, matrices.cc:391 */
        for (size_t i = 1; i < len; i++)
          if (!strchr(" \t\n", s[i])) { order[k++ & 255] = aa_code((unsigned char)s[i]); symbols++; }
      } else {
        int a = aa_code((unsigned char)s[0]);
        const char* p = s + 1;
        const char* end = s + len;
        for (int i = 0; i < symbols; i++) {
          char* q;
          while (p < end && isspace((unsigned char)*p)) p++;
          if (p >= end) break;
          long sc = strtol(p, &q, 10);
          if (q == p) return 0;
          int b = order[i & 255];
          if (a >= 0 && b >= 0 && a < 32 && b < 32) M[(a << 5) + b] = sc;
          p = q;
        }
      }
    }
    s += len + (eol ? 1 : 0);
  }
  return 1;
}

void swo_score_limits(const long* M, long* lo, long* hi, long* limit7, long* limit16)
{
  long l = 100, h = -100;                             /* matrices.cc:561-572 */
  for (int i = 0; i < 1024; i++) {
    if (M[i] < l) l = M[i];
    if (M[i] > h) h = M[i];
  }
  if (lo) *lo = l;
  if (hi) *hi = h;
  if (limit7) *limit7 = 128 - h;                      /* matrices.cc:575 */
  if (limit16) *limit16 = 65536 - h;                  /* matrices.cc:577 */
}

/* ------------------------------------------------------------------ 63-bit kernel ----- */

long swo_fullsw(const unsigned char* dseq, long dlen, const unsigned char* qseq, long qlen,
                const long* M, unsigned char gapopenextend, unsigned char gapextend)
{
  /* search63.cc:28-89: column-major sweep, he[2i] = H of the previous column at row i,
     he[2i+1] = E.  H floored at 0 (64-65); E and F are not. */
  long best = 0;
  long* he = (long*)calloc((size_t)(qlen > 0 ? 2 * qlen : 1), sizeof(long));
  for (long j = 0; j < dlen; j++) {
    const long* row = M + ((long)dseq[j] << 5);
    long f = 0, diag = 0;
    for (long i = 0; i < qlen; i++) {
      long up_left_next = he[2 * i];
      long e = he[2 * i + 1];
      long h = diag + row[qseq[i]];
      if (e > h) h = e;
      if (f > h) h = f;
      if (h < 0) h = 0;
      if (h > best) best = h;
      he[2 * i] = h;
      e -= gapextend;
      f -= gapextend;
      h -= gapopenextend;
      if (h > e) e = h;
      if (h > f) f = h;
      he[2 * i + 1] = e;
      diag = up_left_next;
    }
  }
  free(he);
  return best;
}

/* ------------------------------------------------------------------ 7-bit lane -------- */

static inline uint8_t adds8(uint8_t a, uint8_t b)      /* paddsb */
{
  int r = (int8_t)a + (int8_t)b;
  if (r > 127) r = 127;
  if (r < -128) r = -128;
  return (uint8_t)(int8_t)r;
}
static inline uint8_t subs8(uint8_t a, uint8_t b)      /* psubsb */
{
  int r = (int8_t)a - (int8_t)b;
  if (r > 127) r = 127;
  if (r < -128) r = -128;
  return (uint8_t)(int8_t)r;
}
static inline uint8_t maxu8(uint8_t a, uint8_t b) { return a > b ? a : b; }   /* pmaxub */

long swo_search7_lane(const unsigned char* dseq, long dlen, const unsigned char* qseq, long qlen,
                      const long* M, unsigned char gapopenextend, unsigned char gapextend)
{
  /* One SSE byte lane of search7: hearray bytes start at 0x80 (search7.cc:787; a fresh
     sequence is masked back to 0x80 at 698-705), S starts at Z (812), every 4-column block
     starts with H0-3 = F0-3 = Z (INITIALIZE, 565-583), residues past the end are symbol 0
     (840-843, 922-925).  A channel always runs at least one block (919-928). */
  const uint8_t Z = 0x80;
  const uint8_t Q = gapopenextend, R = gapextend;
  uint8_t* H = (uint8_t*)malloc((size_t)(qlen > 0 ? qlen : 1));
  uint8_t* E = (uint8_t*)malloc((size_t)(qlen > 0 ? qlen : 1));
  memset(H, Z, (size_t)(qlen > 0 ? qlen : 1));
  memset(E, Z, (size_t)(qlen > 0 ? qlen : 1));
  uint8_t S = Z;
  long blocks = (dlen + 3) / 4;
  if (blocks < 1) blocks = 1;
  for (long b = 0; b < blocks; b++) {
    unsigned char d[4];
    for (int c = 0; c < 4; c++) d[c] = (4 * b + c < dlen) ? dseq[4 * b + c] : 0;
    uint8_t hd[4] = {Z, Z, Z, Z}, f[4] = {Z, Z, Z, Z};
    for (long i = 0; i < qlen; i++) {
      uint8_t n0 = H[i], e = E[i], hn[4];
      for (int c = 0; c < 4; c++) {                    /* ONESTEP, search7.cc:585-595 */
        uint8_t p = (uint8_t)(signed char)M[((long)d[c] << 5) + qseq[i]];   /* (char) cast, matrices.cc:585 */
        uint8_t h = adds8(hd[c], p);
        h = maxu8(h, f[c]);
        h = maxu8(h, e);
        S = maxu8(S, h);
        f[c] = subs8(f[c], R);
        e = subs8(e, R);
        hn[c] = h;
        h = subs8(h, Q);
        e = maxu8(e, h);
        f[c] = maxu8(f[c], h);
      }
      H[i] = hn[3];
      E[i] = e;
      hd[0] = n0; hd[1] = hn[0]; hd[2] = hn[1]; hd[3] = hn[2];
    }
  }
  free(H);
  free(E);
  return (long)S - 0x80;                               /* search7.cc:894 */
}

/* ------------------------------------------------------------------ 16-bit lane ------- */

static inline uint16_t adds16(uint16_t a, uint16_t b)   /* paddsw */
{
  int r = (int16_t)a + (int16_t)b;
  if (r > 32767) r = 32767;
  if (r < -32768) r = -32768;
  return (uint16_t)(int16_t)r;
}
static inline uint16_t subs16(uint16_t a, uint16_t b)   /* psubsw */
{
  int r = (int16_t)a - (int16_t)b;
  if (r > 32767) r = 32767;
  if (r < -32768) r = -32768;
  return (uint16_t)(int16_t)r;
}
static inline uint16_t maxs16(uint16_t a, uint16_t b) { return (int16_t)a > (int16_t)b ? a : b; }  /* pmaxsw */

long swo_search16_lane(const unsigned char* dseq, long dlen, const unsigned char* qseq, long qlen,
                       const long* M, unsigned short gapopenextend, unsigned short gapextend,
                       long* bestpos)
{
  /* search16.cc:99-109 (ONESTEP), 320-546 (driver).  Same structure as the 7-bit lane with
     0x8000-biased words and a signed max; after each block the column position is recorded
     if S rose (search16.cc:411-414, 528-534). */
  const uint16_t Z = 0x8000;
  const uint16_t Q = gapopenextend, R = gapextend;
  size_t n = (size_t)(qlen > 0 ? qlen : 1);
  uint16_t* H = (uint16_t*)malloc(n * sizeof(uint16_t));
  uint16_t* E = (uint16_t*)malloc(n * sizeof(uint16_t));
  for (size_t i = 0; i < n; i++) H[i] = E[i] = Z;
  uint16_t S = Z, SL = Z;
  long best = 0;
  long blocks = (dlen + 3) / 4;
  if (blocks < 1) blocks = 1;
  for (long b = 0; b < blocks; b++) {
    unsigned char d[4];
    for (int c = 0; c < 4; c++) d[c] = (4 * b + c < dlen) ? dseq[4 * b + c] : 0;
    uint16_t hd[4] = {Z, Z, Z, Z}, f[4] = {Z, Z, Z, Z};
    for (long i = 0; i < qlen; i++) {
      uint16_t n0 = H[i], e = E[i], hn[4];
      for (int c = 0; c < 4; c++) {
        uint16_t p = (uint16_t)(short)M[((long)d[c] << 5) + qseq[i]];       /* (short) cast, matrices.cc:589 */
        uint16_t h = adds16(hd[c], p);
        h = maxs16(h, f[c]);
        h = maxs16(h, e);
        S = maxs16(S, h);
        f[c] = subs16(f[c], R);
        e = subs16(e, R);
        hn[c] = h;
        h = subs16(h, Q);
        e = maxs16(e, h);
        f[c] = maxs16(f[c], h);
      }
      H[i] = hn[3];
      E[i] = e;
      hd[0] = n0; hd[1] = hn[0]; hd[2] = hn[1]; hd[3] = hn[2];
    }
    if ((int16_t)S > (int16_t)SL) {                    /* _mm_cmpgt_epi16(S, SL) */
      long pos = 4 * (b + 1);
      best = pos < dlen ? pos : dlen;                  /* d_pos stops at d_end */
    }
    SL = S;
  }
  free(H);
  free(E);
  if (bestpos) *bestpos = best;
  return (long)(S ^ 0x8000);                           /* search16.cc:462 */
}

long swo_search16s_lane(const unsigned char* dseq, long dlen, const unsigned char* qseq, long qlen,
                        const long* M, unsigned short gapopenextend, unsigned short gapextend,
                        long* bestpos, long* bestq)
{
  /* search16s.cc: CDEPTH 1 (line 31), so every column starts with H0 = F0 = Z (INITIALIZE) and the
     best-cell bookkeeping of 391-405 runs after each column. */
  const uint16_t Z = 0x8000;
  const uint16_t Q = gapopenextend, R = gapextend;
  size_t n = (size_t)(qlen > 0 ? qlen : 1);
  uint16_t* H = (uint16_t*)malloc(n * sizeof(uint16_t));
  uint16_t* E = (uint16_t*)malloc(n * sizeof(uint16_t));
  for (size_t i = 0; i < n; i++) H[i] = E[i] = Z;
  uint16_t S = Z, SL = Z;
  long bp = 0, bq = -1;
  long cols = dlen > 0 ? dlen : 1;                     /* an empty sequence still runs one padded column */
  for (long j = 0; j < cols; j++) {
    const unsigned char d = j < dlen ? dseq[j] : 0;
    uint16_t hd = Z, f = Z;
    for (long i = 0; i < qlen; i++) {
      uint16_t n0 = H[i], e = E[i];
      uint16_t p = (uint16_t)(short)M[((long)d << 5) + qseq[i]];
      uint16_t h = adds16(hd, p);
      h = maxs16(h, f);
      h = maxs16(h, e);
      S = maxs16(S, h);
      f = subs16(f, R);
      e = subs16(e, R);
      H[i] = h;
      h = subs16(h, Q);
      e = maxs16(e, h);
      f = maxs16(f, h);
      E[i] = e;
      hd = n0;
    }
    if ((int16_t)S > (int16_t)SL) {
      bp = j;                                          /* d_best = d_pos - 1 (S cannot rise on the padded column of an empty sequence) */
      for (long i = qlen - 1; i >= 0; i--)
        if (H[i] == S) bq = i;
    }
    SL = S;
  }
  free(H);
  free(E);
  if (bestpos) *bestpos = bp;
  if (bestq) *bestq = bq;
  return (long)(S ^ 0x8000);
}

/* ------------------------------------------------------------------ escalation -------- */

void swo_search_chunk(const unsigned char* residues, const int64_t* offsets, long nseq,
                      const unsigned char* qseq, long qlen, const long* M,
                      long gapopenextend, long gapextend, long* scores, swo_counters* counters)
{
  /* swipe.cc:1416-1592: every sequence goes through the 7-bit kernel; those whose score is
     not < SCORELIMIT_7 are redone at 16 bit; those not < SCORELIMIT_16 by fullsw.  The gap
     penalties narrow to BYTE / WORD / BYTE at the three call sites (swipe.h:200-258). */
  long limit7, limit16;
  swo_score_limits(M, 0, 0, &limit7, &limit16);
  swo_counters c = {0, 0, 0};
  for (long s = 0; s < nseq; s++) {
    const unsigned char* d = residues + offsets[s];
    long dlen = (long)(offsets[s + 1] - offsets[s]);
    c.compute7++;
    long sc = swo_search7_lane(d, dlen, qseq, qlen, M, (unsigned char)gapopenextend, (unsigned char)gapextend);
    if (!(sc < limit7)) {
      c.compute16++;
      sc = swo_search16_lane(d, dlen, qseq, qlen, M, (unsigned short)gapopenextend, (unsigned short)gapextend, 0);
      if (!(sc < limit16)) {
        c.compute63++;
        sc = swo_fullsw(d, dlen, qseq, qlen, M, (unsigned char)gapopenextend, (unsigned char)gapextend);
      }
    }
    scores[s] = sc;
  }
  if (counters) *counters = c;
}

typedef struct {
  const unsigned char* residues; const int64_t* offsets; long nseq;
  const unsigned char* qseq; long qlen; const long* M; long goe, ge; long* scores;
  long next; pthread_mutex_t mu;
} all63_job;

static void* all63_worker(void* arg)
{
  all63_job* j = (all63_job*)arg;
  for (;;) {
    pthread_mutex_lock(&j->mu);
    long lo = j->next;
    j->next += 256;
    pthread_mutex_unlock(&j->mu);
    if (lo >= j->nseq) break;
    long hi = lo + 256 < j->nseq ? lo + 256 : j->nseq;
    for (long s = lo; s < hi; s++)
      j->scores[s] = swo_fullsw(j->residues + j->offsets[s], (long)(j->offsets[s + 1] - j->offsets[s]),
                                j->qseq, j->qlen, j->M, (unsigned char)j->goe, (unsigned char)j->ge);
  }
  return 0;
}

void swo_search_all63(const unsigned char* residues, const int64_t* offsets, long nseq,
                      const unsigned char* qseq, long qlen, const long* M,
                      long gapopenextend, long gapextend, long* scores, int threads)
{
  all63_job j = {residues, offsets, nseq, qseq, qlen, M, gapopenextend, gapextend, scores, 0,
                 PTHREAD_MUTEX_INITIALIZER};
  if (threads < 1) threads = 1;
  pthread_t* t = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
  for (int i = 0; i < threads; i++) pthread_create(&t[i], 0, all63_worker, &j);
  for (int i = 0; i < threads; i++) pthread_join(t[i], 0);
  free(t);
}

/* ------------------------------------------------------------------ statistics -------- */

int swo_stats_protein(const char* matrix, long gapopen, long gapextend, swo_ka* p)
{
  /* stats.cc:169-247: first row whose (open, extend) match within 0.1 */
  for (int t = 0; t < REFDATA_NKA; t++)
    if (strcasecmp(matrix, refdata_ka_tables[t].name) == 0) {
      for (int i = 0; i < refdata_ka_tables[t].n; i++) {
        const refdata_ka_row* r = &refdata_ka_tables[t].rows[i];
        if (fabs(r->go - (double)gapopen) < 0.1 && fabs(r->ge - (double)gapextend) < 0.1) {
          p->lambda = r->lambda; p->K = r->K; p->H = r->H; p->alpha = r->alpha; p->beta = r->beta;
          return 1;
        }
      }
      return 0;
    }
  return 0;
}

int swo_stats_default_gaps(const char* matrix, long* gapopen, long* gapextend)
{
  /* stats.cc:249-325: the row flagged BLAST_MATRIX_BEST */
  for (int t = 0; t < REFDATA_NKA; t++)
    if (strcasecmp(matrix, refdata_ka_tables[t].name) == 0)
      for (int i = 0; i < refdata_ka_tables[t].n; i++)
        if (refdata_ka_tables[t].rows[i].best) {
          *gapopen = (long)refdata_ka_tables[t].rows[i].go;
          *gapextend = (long)refdata_ka_tables[t].rows[i].ge;
          return 1;
        }
  return 0;
}

int swo_stats_nucleotide(long match, long mismatch, long gapopen, long gapextend, swo_ka* p)
{
  /* stats.cc:44-167: table by (reward, penalty); penalties at or above both maxima select the
     (0, 0) "linear" row (stats.cc:147-151) */
  for (int t = 0; t < REFDATA_NNT; t++)
    if (refdata_nt_tables[t].match == match && refdata_nt_tables[t].mismatch == mismatch) {
      const refdata_nt_table* nt = &refdata_nt_tables[t];
      if (gapopen >= nt->gomax && gapextend >= nt->gemax) { gapopen = 0; gapextend = 0; }
      for (int i = 0; i < nt->n; i++)
        if (fabs(nt->rows[i][0] - (double)gapopen) < 0.1 && fabs(nt->rows[i][1] - (double)gapextend) < 0.1) {
          p->lambda = nt->rows[i][2]; p->K = nt->rows[i][3]; p->H = nt->rows[i][4];
          p->alpha = nt->rows[i][5]; p->beta = nt->rows[i][6];
          return 1;
        }
      return 0;
    }
  return 0;
}

int swo_length_adjustment(double K, double logK, double alpha_d_lambda, double beta,
                          int query_length, long db_length, int db_num_seqs, int* adj)
{
  /* blastkar_partial.c:656-748 (NCBI public domain): fixed point of
       ell = alpha/lambda * (ln K + ln((m-ell)(n-N ell))) + beta
     by at most 20 bracketed iterations; floor of the largest ell known to be below it. */
  const double m = query_length, n = (double)db_length, N = db_num_seqs;
  double lo = 0.0, hi, cur = 0.0, next = 0.0;
  int ok = 0;
  {
    double a = N, mb = m * N + n;
    double c = n * m - (m > n ? m : n) / K;
    if (c < 0) { *adj = 0; return 1; }
    hi = 2 * c / (mb + sqrt(mb * mb - 4 * a * c));
  }
  for (int it = 1; it <= 20; it++) {
    cur = next;
    double space = (m - cur) * (n - N * cur);
    double prop = alpha_d_lambda * (logK + log(space)) + beta;
    if (prop >= cur) {
      lo = cur;
      if (prop - lo <= 1.0) { ok = 1; break; }
      if (lo >= hi) break;
    } else {
      hi = cur;
    }
    if (lo <= prop && prop <= hi) next = prop;
    else next = (it == 1) ? hi : (lo + hi) / 2;
  }
  if (ok) {
    *adj = (int)lo;
    double up = ceil(lo);
    if (up <= hi) {
      double space = (m - up) * (n - N * up);
      if (alpha_d_lambda * (logK + log(space)) + beta >= up) *adj = (int)up;
    }
  } else {
    *adj = (int)lo;
  }
  return ok ? 0 : 1;
}

/* ------------------------------------------------------------------ hit list ---------- */

swo_hits* swo_hits_new(long descriptions, long alignments, long minscore, long maxscore,
                       double minexpect, double expect, int symtype, int querystrands,
                       const char* matrix, long match, long mismatch, long gapopen, long gapextend,
                       long qlen, long dbseqs, long dbsyms, long effdbsize)
{
  /* hits.cc:283-511 for symtype 0..4; for the translated types qlen / dbsyms are NUCLEOTIDE counts where
     the query / database is nucleotide, as query.nt[0].len and db_getsymcount() are */
  swo_hits* h = (swo_hits*)calloc(1, sizeof(swo_hits));
  h->keephits = descriptions > alignments ? descriptions : alignments;
  long maxhits = dbseqs;
  if (symtype == 0 && querystrands == 3) maxhits *= 2;              /* hits.cc:290-294 */
  if (symtype == 2) maxhits *= querystrands == 3 ? 6 : 3;           /* hits.cc:295-301 */
  if (symtype == 3) maxhits *= 6;                                   /* hits.cc:302-305 */
  if (symtype == 4) maxhits *= querystrands == 3 ? 36 : 18;         /* hits.cc:306-312 */
  if (h->keephits > maxhits) h->keephits = maxhits;
  h->list = (swo_hit*)calloc((size_t)(h->keephits > 0 ? h->keephits : 1), sizeof(swo_hit));
  swo_ka p;
  int avail = symtype == 0 ? swo_stats_nucleotide(match, mismatch, gapopen, gapextend, &p)
            : symtype == 4 ? swo_stats_protein(matrix, 32767, 32767, &p)      /* ungapped row, hits.cc:401-410 */
                           : swo_stats_protein(matrix, gapopen, gapextend, &p);
  if (symtype == 2 || symtype == 4) qlen = qlen / 3;               /* hits.cc:436-437 */
  if ((symtype == 3 || symtype == 4) && effdbsize <= 0) dbsyms = dbsyms / 3;   /* hits.cc:446-449 */
  h->stats_available = avail;
  h->scorethreshold = minscore;
  h->upperscorethreshold = maxscore;
  if (avail) {
    h->lambda = p.lambda;
    h->K = p.K;
    h->logK = log(p.K);
    h->lambda_d_log2 = p.lambda / log(2.0);
    h->logK_d_log2 = h->logK / log(2.0);
    int seqcount = (int)dbseqs;                                      /* "int seqcount", hits.cc:330 */
    long dlen = effdbsize > 0 ? effdbsize : dbsyms;
    int lenadj = 0;
    swo_length_adjustment(p.K, h->logK, p.alpha / p.lambda, p.beta, (int)qlen, dlen, seqcount, &lenadj);
    h->lenadj = lenadj;
    h->m = qlen - lenadj;
    h->n = effdbsize > 0 ? effdbsize : dlen - (long)seqcount * lenadj;
    h->Kmn = p.K * (double)h->m * (double)h->n;
    long minscore_expect = (long)(ceil(-log(expect / h->Kmn) / p.lambda));   /* hits.cc:491 */
    if (minscore_expect > minscore) h->scorethreshold = minscore_expect;
    if (minexpect > 0.0) {
      long maxscore_expect = (long)(floor(-log(minexpect / h->Kmn) / p.lambda));
      if (maxscore_expect < maxscore) h->upperscorethreshold = maxscore_expect;
    }
  }
  h->init_threshold = h->scorethreshold;
  return h;
}

void swo_hits_enter(swo_hits* h, long seqno, long score, long qstrand, long qframe, long dstrand, long dframe)
{
  /* hits.cc:163-222: bounded list kept sorted by (score desc, seqno desc) */
  if (score > h->upperscorethreshold) h->obvious++;
  if (score >= h->init_threshold) h->totalhits++;
  if (score < h->scorethreshold || score > h->upperscorethreshold) return;
  long place = h->count;
  while (place > 0 && (score > h->list[place - 1].score ||
                       (score == h->list[place - 1].score && seqno > h->list[place - 1].seqno)))
    place--;
  long move = (h->count < h->keephits ? h->count : h->keephits - 1) - place;
  for (long j = move; j > 0; j--) h->list[place + j] = h->list[place + j - 1];
  if (place < h->keephits) {
    swo_hit e = {seqno, score, qstrand, qframe, dstrand, dframe};
    h->list[place] = e;
    if (h->count < h->keephits) h->count++;
  }
  if (h->count == h->keephits && h->keephits > 0) h->scorethreshold = h->list[h->keephits - 1].score;
}

double swo_hits_expect(const swo_hits* h, long score) { return h->Kmn * exp(-h->lambda * score); }      /* hits.cc:1777 */
double swo_hits_bits(const swo_hits* h, long score) { return h->lambda_d_log2 * score - h->logK_d_log2; } /* hits.cc:1779 */

void swo_hits_free(swo_hits* h)
{
  if (h) { free(h->list); free(h); }
}

/* ===== alignment phase: end points -> start points -> edit script =====================
   Restates align.cc (region 38-164, diff 236-467, align 469-519): the local alignment is
   delimited by a forward pass (or by the search16s hint) and a backward pass, then spelled
   out by the linear-space divide-and-conquer of Myers & Miller with the reference's tie rules.
   q = gap open, r = gap extend (a gap of k symbols costs q + k*r), M index (d << 5) + qsym. */
typedef struct { char* text; long used, room; char kind; long run; } script_t;

static void script_flush(script_t* s)                       /* push(), align.cc:184-207 */
{
  if (s->run <= 0) return;
  if (s->room - s->used < 32) {
    s->room = 2 * s->room + 64;
    s->text = (char*) realloc(s->text, (size_t) s->room);
  }
  s->used += sprintf(s->text + s->used, "%c%ld", s->kind, s->run);
}
static void script_add(script_t* s, char kind, long n)      /* newop(), align.cc:209-219 */
{
  if (s->kind == kind) { s->run += n; return; }
  script_flush(s);
  s->kind = kind;
  s->run = n;
}

typedef struct {
  const unsigned char *a, *b;    /* a = query, b = database sequence */
  const long* M;
  long q, r;
  script_t* out;
} mm_ctx;

static inline long mm_sub(const mm_ctx* c, long ai, long bj) { return c->M[((long) c->b[bj] << 5) + c->a[ai]]; }
static inline long max2(long x, long y) { return x > y ? x : y; }

/* One sweep of `rows` query rows over `n` database columns with a global (no zero floor) affine
   recurrence whose first-column gap was opened at cost `edge` (align.cc:341-373 forward,
   381-414 reverse).  dir = +1 walks a0+0.., b0+0..; dir = -1 walks from the far ends back.
   best[j] = best score ending anywhere, gapd[j] = best score ending in a gap in the query rows. */
static void mm_sweep(const mm_ctx* c, long a0, long b0, long rows_total, long n, long rows, int dir, long edge,
                     long* best, long* gapd)
{
  long t = -c->q;
  best[0] = 0;
  for (long j = 1; j <= n; j++) { t -= c->r; best[j] = t; gapd[j] = t - c->q; }
  t = -edge;
  for (long i = 1; i <= rows; i++) {
    long diag = best[0];
    t -= c->r;
    long h = t, f = t - c->q;
    best[0] = t;
    const long ai = dir > 0 ? a0 + i - 1 : a0 + rows_total - i;
    for (long j = 1; j <= n; j++) {
      const long bj = dir > 0 ? b0 + j - 1 : b0 + n - j;
      f = max2(f, h - c->q) - c->r;
      gapd[j] = max2(gapd[j], best[j] - c->q) - c->r;
      h = diag + mm_sub(c, ai, bj);
      if (f > h) h = f;
      if (gapd[j] > h) h = gapd[j];
      diag = best[j];
      best[j] = h;
    }
  }
  gapd[0] = best[0];
}

/* diff(), align.cc:236-467.  m query symbols from a0 against n database symbols from b0; lead /
   trail = what opening a query-side gap costs at the left / right edge (0 when one is already open) */
static void mm_solve(const mm_ctx* c, long a0, long b0, long m, long n, long lead, long trail)
{
  if (n == 0) { if (m > 0) script_add(c->out, 'D', m); return; }
  if (m == 0) { script_add(c->out, 'I', n); return; }
  if (m == 1) {                                              /* align.cc:260-328 */
    long top, at;
    if (lead <= trail) { top = -lead - (1 + n) * c->r - c->q; at = -1; }
    else               { top = -c->q - (1 + n) * c->r - trail; at = n; }
    for (long j = 0; j < n; j++) {
      long v = mm_sub(c, a0, b0 + j) - c->r * (n - 1);
      if (j > 0) v -= c->q;
      if (j < n - 1) v -= c->q;
      if (v > top) { top = v; at = j; }
    }
    if (at == -1) { script_add(c->out, 'D', 1); script_add(c->out, 'I', n); }
    else if (at == n) { script_add(c->out, 'I', n); script_add(c->out, 'D', 1); }
    else {
      if (at > 0) script_add(c->out, 'I', at);
      script_add(c->out, 'M', 1);
      if (at < n - 1) script_add(c->out, 'I', n - 1 - at);
    }
    return;
  }
  const long half = m / 2;
  long* buf = (long*) malloc((size_t) (4 * (n + 1)) * sizeof(long));
  long *fh = buf, *fe = buf + (n + 1), *rh = buf + 2 * (n + 1), *re = buf + 3 * (n + 1);
  mm_sweep(c, a0, b0, m, n, half, +1, lead, fh, fe);
  mm_sweep(c, a0, b0, m, n, m - half, -1, trail, rh, re);
  long top = 0, cut = -1;
  int through_gap = -1;
  for (long j = 0; j <= n; j++) {                            /* align.cc:419-432: first strict maximum */
    const long v = fh[j] + rh[n - j];
    if (through_gap < 0 || v > top) { top = v; cut = j; through_gap = 0; }
  }
  for (long j = 0; j <= n; j++) {                            /* align.cc:437-446: last >= wins */
    const long v = fe[j] + re[n - j] + c->q;
    if (v >= top) { top = v; cut = j; through_gap = 1; }
  }
  free(buf);
  if (!through_gap) {
    mm_solve(c, a0, b0, half, cut, lead, c->q);
    mm_solve(c, a0 + half, b0 + cut, m - half, n - cut, c->q, trail);
  } else {
    mm_solve(c, a0, b0, half - 1, cut, lead, 0);
    script_add(c->out, 'D', 2);
    mm_solve(c, a0 + half + 1, b0 + cut, m - half - 1, n - cut, 0, trail);
  }
}

long swo_align(const unsigned char* qseq, long qlen, const unsigned char* dseq, long dlen, const long* M,
               long gapopen, long gapextend, long hint_score, long hint_q_end, long hint_d_end,
               swo_alignment* res, char* cigar, long cigar_room)
{
  const long q = gapopen, r = gapextend;
  long score = 0, qe = hint_q_end, de = hint_d_end;
  long* hh = (long*) malloc((size_t) (dlen > 0 ? dlen : 1) * sizeof(long));
  long* ee = (long*) malloc((size_t) (dlen > 0 ? dlen : 1) * sizeof(long));
  if (hint_score) {
    score = hint_score;                                      /* align.cc:62-65 */
  } else {                                                   /* align.cc:70-106 */
    for (long j = 0; j < dlen; j++) { hh[j] = 0; ee[j] = -q; }
    for (long i = 0; i < qlen; i++) {
      long h = 0, diag = 0, f = -q;
      for (long j = 0; j < dlen; j++) {
        f = max2(f, h - q) - r;
        ee[j] = max2(ee[j], hh[j] - q) - r;
        h = diag + M[((long) dseq[j] << 5) + qseq[i]];
        if (h < 0) h = 0;
        if (f > h) h = f;
        if (ee[j] > h) h = ee[j];
        diag = hh[j];
        hh[j] = h;
        if (h > score) { score = h; qe = i; de = j; }
      }
    }
  }
  /* backward pass from the end cell until the score is recovered, align.cc:111-154 */
  long qs = -1, ds = -1, found = 0, cost = 0;
  if (qe < qlen && de < dlen) {
    for (long j = de; j >= 0; j--) hh[j] = ee[j] = -1;
    for (long i = qe; i >= 0 && !found; i--) {
      long h = -1, f = -1, diag = (i == qe) ? 0 : -1;
      for (long j = de; j >= 0; j--) {
        f = max2(f, h - q) - r;
        ee[j] = max2(ee[j], hh[j] - q) - r;
        h = diag + M[((long) dseq[j] << 5) + qseq[i]];
        if (f > h) h = f;
        if (ee[j] > h) h = ee[j];
        diag = hh[j];
        hh[j] = h;
        if (h > cost) {
          cost = h; qs = i; ds = j;
          if (cost >= score) { found = 1; break; }
        }
      }
    }
  }
  free(hh);
  free(ee);
  if (!found) return -1;                                     /* "Internal error in align function." */
  script_t s = {0, 0, 0, 0, 0};
  mm_ctx c = {qseq, dseq, M, q, r, &s};
  mm_solve(&c, qs, ds, qe - qs + 1, de - ds + 1, q, q);      /* align.cc:502-513 */
  script_flush(&s);
  res->q_start = qs; res->d_start = ds; res->q_end = qe; res->d_end = de; res->score = score;
  long n = s.used;
  if (cigar && cigar_room > 0) {
    long k = n < cigar_room - 1 ? n : cigar_room - 1;
    if (s.text) memcpy(cigar, s.text, (size_t) k);
    cigar[k] = 0;
  }
  free(s.text);
  return n;
}


/* ===== translated searches: genetic code tables and six-frame translation ================ */
/* translate_createtable (query.cc:377-455): table[256*a + 16*b + c] for three IUPAC nibbles (A=1 C=2 G=4
   T=8): the amino acid every compatible codon agrees on, B for a D/N mix, Z for an E/Q mix, else X */
int swo_translate_table(int gencode, unsigned char* table)
{
  if (gencode < 1 || gencode > 23 || !refdata_gencode[gencode - 1][0]) return 0;
  const char* code = refdata_gencode[gencode - 1];
  static const int tcag_of_bit[4] = {2, 1, 3, 0};           /* bit 0 = A, 1 = C, 2 = G, 3 = T -> index in T,C,A,G order */
  for (int a = 0; a < 16; a++)
    for (int b = 0; b < 16; b++)
      for (int c = 0; c < 16; c++) {
        char aa = '-';
        for (int i = 0; i < 4; i++)
          for (int j = 0; j < 4; j++)
            for (int k = 0; k < 4; k++) {
              if (!((a >> i) & 1) || !((b >> j) & 1) || !((c >> k) & 1)) continue;
              const char x = code[16 * tcag_of_bit[i] + 4 * tcag_of_bit[j] + tcag_of_bit[k]];
              if (aa == '-' || aa == x) aa = x;
              else if (aa == 'B' && (x == 'D' || x == 'N')) ;
              else if ((aa == 'D' && (x == 'B' || x == 'N')) || (aa == 'N' && (x == 'B' || x == 'D'))) aa = 'B';
              else if (aa == 'Z' && (x == 'Q' || x == 'E')) ;
              else if ((aa == 'E' && (x == 'Z' || x == 'Q')) || (aa == 'Q' && (x == 'Z' || x == 'E'))) aa = 'Z';
              else aa = 'X';
            }
        if (aa == '-') aa = 'X';
        table[256 * a + 16 * b + c] = (unsigned char) aa_code(aa);
      }
  return 1;
}

/* translate (query.cc:463-506) = db_translate (database.cc:1182-1218): frame f of strand s has
   (dlen - f) / 3 codons; strand 1 reads the reverse complement.  Returns the protein length. */
long swo_translate(const unsigned char* dna, long dlen, int strand, int frame, const unsigned char* table,
                   unsigned char* prot)
{
  static const unsigned char compl4[16] = {0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15};
  const long plen = dlen - frame >= 0 ? (dlen - frame) / 3 : 0;
  if (!strand) {
    long pos = frame;
    for (long k = 0; k < plen; k++, pos += 3)
      prot[k] = table[256 * dna[pos] + 16 * dna[pos + 1] + dna[pos + 2]];
  } else {
    long pos = dlen - 1 - frame;
    for (long k = 0; k < plen; k++, pos -= 3)
      prot[k] = table[256 * compl4[dna[pos] & 15] + 16 * compl4[dna[pos - 1] & 15] + compl4[dna[pos - 2] & 15]];
  }
  return plen;
}
