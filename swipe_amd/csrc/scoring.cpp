// Host-side scoring inputs of the hot path: substitution matrices, gap defaults,
// Karlin-Altschul statistics and the E-value / bit-score arithmetic.
//
// Mirrors, for the parts the GPU path needs, reference matrices.cc:520-591 (matrix tables and
// the unset-cell = -1 rule), stats.cc:44-325 (parameter lookup, default gaps),
// blastkar_partial.c:656-748 (NCBI length adjustment) and hits.cc:283-511, 1777-1779
// (effective search space, score thresholds, expect, bits).  Double arithmetic follows the
// reference's expression order so results are bit-identical under the same libm.
#include "../../include/swipe_amd.h"
#include "host_util.h"

#include <cctype>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <strings.h>

namespace {
#include "refdata.inc"

int stdaa_code(int ch)
{
  static const char alphabet[] = "-ABCDEFGHIKLMNPQRSTVWXYZU*OJ";   // query.cc:178
  if (!ch) return -1;
  const char* p = std::strchr(alphabet, std::toupper(ch));
  return p ? int(p - alphabet) : -1;
}

void fill_default(int64_t* m) { for (int i = 0; i < 1024; ++i) m[i] = -1; }   // matrices.cc:531

struct ka_params { double lambda, K, H, alpha, beta; };

bool lookup_protein(const char* name, int64_t go, int64_t ge, ka_params& out)
{
  for (int t = 0; t < REFDATA_NKA; ++t) {
    if (strcasecmp(name, refdata_ka_tables[t].name) != 0) continue;
    for (int i = 0; i < refdata_ka_tables[t].n; ++i) {
      const refdata_ka_row& r = refdata_ka_tables[t].rows[i];
      if (std::fabs(r.go - double(go)) < 0.1 && std::fabs(r.ge - double(ge)) < 0.1) {   // stats.cc:230-231
        out = {r.lambda, r.K, r.H, r.alpha, r.beta};
        return true;
      }
    }
    return false;
  }
  return false;
}

bool lookup_nucleotide(int64_t match, int64_t mismatch, int64_t go, int64_t ge, ka_params& out)
{
  for (int t = 0; t < REFDATA_NNT; ++t) {
    const refdata_nt_table& nt = refdata_nt_tables[t];
    if (nt.match != match || nt.mismatch != mismatch) continue;
    if (go >= nt.gomax && ge >= nt.gemax) { go = 0; ge = 0; }                            // stats.cc:147-151
    for (int i = 0; i < nt.n; ++i)
      if (std::fabs(nt.rows[i][0] - double(go)) < 0.1 && std::fabs(nt.rows[i][1] - double(ge)) < 0.1) {
        out = {nt.rows[i][2], nt.rows[i][3], nt.rows[i][4], nt.rows[i][5], nt.rows[i][6]};
        return true;
      }
    return false;
  }
  return false;
}

// NCBI BlastComputeLengthAdjustment (blastkar_partial.c:656-748, public domain): largest
// integer below the fixed point of  ell -> alpha/lambda * ln(K (m-ell)(n-N ell)) + beta.
int length_adjustment(double K, double logK, double a_over_l, double beta, int qlen, int64_t dblen, int nseq)
{
  const double m = qlen, n = double(dblen), N = nseq;
  double lower = 0, upper, ell = 0, ell_next = 0;
  bool converged = false;
  {
    const double qa = N, qmb = m * N + n, qc = n * m - (m > n ? m : n) / K;
    if (qc < 0) return 0;
    upper = 2 * qc / (qmb + std::sqrt(qmb * qmb - 4 * qa * qc));
  }
  for (int iter = 1; iter <= 20; ++iter) {
    ell = ell_next;
    const double ss = (m - ell) * (n - N * ell);
    const double ell_bar = a_over_l * (logK + std::log(ss)) + beta;
    if (ell_bar >= ell) {
      lower = ell;
      if (ell_bar - lower <= 1.0) { converged = true; break; }
      if (lower >= upper) break;
    } else {
      upper = ell;
    }
    if (lower <= ell_bar && ell_bar <= upper) ell_next = ell_bar;
    else ell_next = iter == 1 ? upper : (lower + upper) / 2;
  }
  int adj = int(lower);
  if (converged) {
    const double c = std::ceil(lower);
    if (c <= upper) {
      const double ss = (m - c) * (n - N * c);
      if (a_over_l * (logK + std::log(ss)) + beta >= c) adj = int(c);
    }
  }
  return adj;
}
}  // namespace

extern "C" int swa_matrix_builtin(const char* name, int64_t* matrix)
try {
  if (!name || !matrix) return swa::fail(SWA_EINVAL, "swa_matrix_builtin: null argument");
  for (int k = 0; k < REFDATA_NMATRICES; ++k)
    if (strcasecmp(name, refdata_matrix_names[k]) == 0) {
      fill_default(matrix);
      for (int d = 0; d < 28; ++d)
        for (int q = 0; q < 28; ++q) matrix[(d << 5) | q] = refdata_matrices[k][d][q];
      return SWA_OK;
    }
  return swa::fail(SWA_EINVAL, "unknown score matrix name");
} SWA_CATCH

extern "C" int swa_matrix_nucleotide(int64_t match, int64_t mismatch, int64_t* matrix)
try {
  if (!matrix) return swa::fail(SWA_EINVAL, "swa_matrix_nucleotide: null argument");
  fill_default(matrix);
  for (int d = 1; d < 16; ++d)                                 // matrices.cc:533-538: exact equality of base masks
    for (int q = 1; q < 16; ++q) matrix[(d << 5) | q] = d == q ? match : mismatch;
  return SWA_OK;
} SWA_CATCH

extern "C" int swa_matrix_parse(const char* text, int64_t* matrix)
try {
  // matrices.cc:352-430: '#'/blank lines skipped, a line starting with blank or tab names the
  // columns, any other line is "<row letter> score score ..." -> [(row << 5) | column]
  if (!text || !matrix) return swa::fail(SWA_EINVAL, "swa_matrix_parse: null argument");
  fill_default(matrix);
  int columns[256];
  int ncol = 0;
  const char* p = text;
  while (*p) {
    const char* e = std::strchr(p, '\n');
    const size_t len = e ? size_t(e - p) : std::strlen(p);
    if (len && p[0] != '#') {
      if (p[0] == ' ' || p[0] == '\t') {
        int k = 0;
        for (size_t i = 1; i < len; ++i)
          if (!std::isspace((unsigned char)p[i])) { columns[k++ & 255] = stdaa_code((unsigned char)p[i]); ++ncol; }
      } else {
        const int row = stdaa_code((unsigned char)p[0]);
        const char* c = p + 1;
        const char* end = p + len;
        for (int i = 0; i < ncol; ++i) {
          while (c < end && std::isspace((unsigned char)*c)) ++c;
          if (c >= end) break;
          char* stop;
          const long v = std::strtol(c, &stop, 10);
          if (stop == c) return swa::fail(SWA_EINVAL, "Problem parsing score matrix file.");
          const int col = columns[i & 255];
          if (row >= 0 && col >= 0) matrix[(row << 5) | col] = v;
          c = stop;
        }
      }
    }
    p += len + (e ? 1 : 0);
  }
  return SWA_OK;
} SWA_CATCH

extern "C" int swa_default_gaps(const char* matrixname, int64_t* gapopen, int64_t* gapextend)
try {
  if (!matrixname || !gapopen || !gapextend) return swa::fail(SWA_EINVAL, "swa_default_gaps: null argument");
  for (int t = 0; t < REFDATA_NKA; ++t)
    if (strcasecmp(matrixname, refdata_ka_tables[t].name) == 0)
      for (int i = 0; i < refdata_ka_tables[t].n; ++i)
        if (refdata_ka_tables[t].rows[i].best) {                                          // stats.cc:314-321
          *gapopen = int64_t(refdata_ka_tables[t].rows[i].go);
          *gapextend = int64_t(refdata_ka_tables[t].rows[i].ge);
          return SWA_OK;
        }
  return swa::fail(SWA_EINVAL, "no default gap penalties for this matrix");
} SWA_CATCH

// (long) of a double as the reference's x86-64 binary performs it (hits.cc:491, 497: cvttsd2si): anything outside the range
// of a long - an empty query makes Kmn 0 and the threshold -inf - becomes the "integer indefinite" value LONG_MIN.  In C++
// that conversion is undefined behaviour (UBSan found it on the empty query of the multi-query fixture); spelled out, the
// result is the same on every compiler.
static int64_t to_long(double x)
{
  if (!(x >= -9223372036854775808.0 && x < 9223372036854775808.0)) return INT64_MIN;
  return int64_t(x);
}

extern "C" int swa_stats_init(int symtype, const char* matrixname, int64_t match, int64_t mismatch,
                              int64_t gapopen, int64_t gapextend, int64_t qlen,
                              int64_t db_seqcount, int64_t db_symcount, int64_t effdbsize,
                              int64_t minscore, int64_t maxscore, double minexpect, double expect,
                              swa_stats_t* out)
try {
  if (!out) return swa::fail(SWA_EINVAL, "swa_stats_init: null output");
  std::memset(out, 0, sizeof *out);
  out->scorethreshold = minscore;                                                         // hits.cc:486-487
  out->upperscorethreshold = maxscore;
  if (symtype < 0 || symtype > 4) return swa::fail(SWA_EINVAL, "swa_stats_init: symtype must be 0..4");
  ka_params ka{};
  // translated against translated (-p 4) uses the matrix's ungapped row (hits.cc:401-410)
  const bool ok = symtype == SWA_SYMTYPE_NUCLEOTIDE ? lookup_nucleotide(match, mismatch, gapopen, gapextend, ka)
                  : symtype == SWA_SYMTYPE_TRANSLATED_BOTH ? (matrixname && lookup_protein(matrixname, 32767, 32767, ka))
                                                           : (matrixname && lookup_protein(matrixname, gapopen, gapextend, ka));
  // lengths in codons where the query / database is nucleotide (hits.cc:436-449)
  if (symtype == SWA_SYMTYPE_TRANSLATED_QUERY || symtype == SWA_SYMTYPE_TRANSLATED_BOTH) qlen /= 3;
  if ((symtype == SWA_SYMTYPE_TRANSLATED_DB || symtype == SWA_SYMTYPE_TRANSLATED_BOTH) && effdbsize <= 0) db_symcount /= 3;
  out->available = ok ? 1 : 0;
  if (!ok) return SWA_OK;
  out->lambda = ka.lambda; out->K = ka.K; out->H = ka.H; out->alpha = ka.alpha; out->beta = ka.beta;
  out->logK = std::log(ka.K);                                                             // hits.cc:370-372 / 439-441
  out->lambda_d_log2 = ka.lambda / std::log(2.0);
  out->logK_d_log2 = out->logK / std::log(2.0);
  const int seqcount = int(db_seqcount);                                                  // "int seqcount", hits.cc:330
  const int64_t dlen = effdbsize > 0 ? effdbsize : db_symcount;
  const int adj = length_adjustment(ka.K, out->logK, ka.alpha / ka.lambda, ka.beta, int(qlen), dlen, seqcount);
  out->lenadj = adj;
  out->m = qlen - adj;
  out->n = effdbsize > 0 ? effdbsize : dlen - int64_t(seqcount) * adj;
  out->Kmn = ka.K * double(out->m) * double(out->n);
  const int64_t by_expect = to_long(std::ceil(-std::log(expect / out->Kmn) / ka.lambda));  // hits.cc:491
  if (by_expect > minscore) out->scorethreshold = by_expect;
  if (minexpect > 0.0) {
    const int64_t by_min = to_long(std::floor(-std::log(minexpect / out->Kmn) / ka.lambda));
    if (by_min < maxscore) out->upperscorethreshold = by_min;
  }
  return SWA_OK;
} SWA_CATCH

extern "C" double swa_evalue(const swa_stats_t* st, int64_t score)
{
  return st->Kmn * std::exp(-st->lambda * score);                                         // hits.cc:1777
}

extern "C" double swa_bits(const swa_stats_t* st, int64_t score)
{
  return st->lambda_d_log2 * score - st->logK_d_log2;                                     // hits.cc:1779
}

// ---- translated searches: genetic codes (query.cc:118-170) and six-frame translation --------------
extern "C" const char* swa_gencode_name(int gencode)
{
  if (gencode < 1 || gencode > 23 || !refdata_gencode[gencode - 1][0]) return nullptr;
  return refdata_gencode_names[gencode - 1];
}

extern "C" int swa_translate_table(int gencode, uint8_t* table)
try {
  if (!table) return swa::fail(SWA_EINVAL, "swa_translate_table: null output");
  if (!swa_gencode_name(gencode)) return swa::fail(SWA_EINVAL, "Illegal genetic code specified.");
  const char* code = refdata_gencode[gencode - 1];           // 64 letters, codon positions in T,C,A,G order
  static const char stdaa[] = "-ABCDEFGHIKLMNPQRSTVWXYZU*OJ";
  // the set of amino acids each of the 4096 nibble triples can stand for, then translate_createtable's
  // merging rule (query.cc:377-455): one letter -> itself, {D,N,B} -> B, {E,Q,Z} -> Z, anything else -> X
  for (int t = 0; t < 4096; ++t) {
    const int n[3] = {t >> 8, (t >> 4) & 15, t & 15};
    unsigned seen = 0;                                       // bit per stdaa code
    static const int tcag[4] = {2, 1, 3, 0};                 // nibble bit (A,C,G,T) -> index in T,C,A,G
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j)
        for (int k = 0; k < 4; ++k)
          if (((n[0] >> i) & 1) && ((n[1] >> j) & 1) && ((n[2] >> k) & 1)) {
            const char* p = std::strchr(stdaa, code[16 * tcag[i] + 4 * tcag[j] + tcag[k]]);
            seen |= 1u << (p ? int(p - stdaa) : 21);
          }
    auto bit = [](char c) { return 1u << int(std::strchr(stdaa, c) - stdaa); };
    int out;
    if (seen == 0) out = 21;                                  // X
    else if ((seen & (seen - 1)) == 0) out = __builtin_ctz(seen);
    else if ((seen & ~(bit('D') | bit('N') | bit('B'))) == 0) out = 2;    // B
    else if ((seen & ~(bit('E') | bit('Q') | bit('Z'))) == 0) out = 23;   // Z
    else out = 21;
    table[t] = uint8_t(out);
  }
  return SWA_OK;
} SWA_CATCH

extern "C" int swa_translate(const uint8_t* dna, int64_t dlen, int strand, int frame, const uint8_t* table,
                             uint8_t* prot, int64_t* plen)
try {
  if (!table || !plen || dlen < 0 || (dlen > 0 && !dna) || frame < 0 || frame > 2 || strand < 0 || strand > 1)
    return swa::fail(SWA_EINVAL, "swa_translate: bad argument");
  const int64_t n = dlen - frame > 0 ? (dlen - frame) / 3 : 0;
  *plen = n;
  if (n > 0 && !prot) return swa::fail(SWA_EINVAL, "swa_translate: null output");
  static const uint8_t compl4[16] = {0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15};
  for (int64_t k = 0; k < n; ++k) {
    int a, b, c;
    if (!strand) {
      const uint8_t* p = dna + frame + 3 * k;
      a = p[0] & 15; b = p[1] & 15; c = p[2] & 15;
    } else {
      const uint8_t* p = dna + dlen - 1 - frame - 3 * k;
      a = compl4[p[0] & 15]; b = compl4[p[-1] & 15]; c = compl4[p[-2] & 15];
    }
    prot[k] = table[256 * a + 16 * b + c];
  }
  return SWA_OK;
} SWA_CATCH
