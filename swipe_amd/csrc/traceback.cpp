// Alignment phase, host part: from the end point of a local alignment to its start point and the
// edit script, for the few hits (-b, at most a few hundred) that get an alignment.
//
// Mirrors align() of the reference (align.cc:469-519) as hits_align calls it (hits.cc:546-618):
//   1. end cell: given (the search16s end point found on the GPU by swa_endpoints_kernel) or, when the
//      reference would not trust a hint (hits.cc:587), the first strict maximum of a forward
//      query-row-major sweep (align.cc:70-106);
//   2. start cell: backward sweep from the end cell, anchored there, stopping at the first cell (query
//      row descending, then database column descending) whose value reaches the score (align.cc:111-154);
//   3. edit script between the two cells in linear space (Myers & Miller 1988 as arranged by Huang,
//      Hardison & Miller 1990 - align.cc:236-467), with the reference's tie rules: the join that passes
//      through a gap wins ties against the plain join and the last such column wins, the plain join keeps
//      the first best column, and a lone query symbol prefers the leftmost best partner.
// The recursion of the reference is replaced by an explicit work stack; results are identical.
//
// Costs: a gap of k symbols costs open + k*extend; substitution scores are matrix[(d << 5) + q].
#include "traceback.h"

#include <algorithm>

namespace swa {
namespace {

struct Sweeper {
  const uint8_t* a;       // query
  const uint8_t* b;       // database sequence
  const int32_t* mat;     // [(d << 5) + q]
  int64_t open, ext;

  int64_t sub(int64_t ai, int64_t bj) const { return mat[(int(b[bj]) << 5) + a[ai]]; }

  // `rows` query rows against n database columns, global recurrence; `edge` = cost of opening the gap
  // that runs down column 0.  forward: rows a0.., columns b0..; backward: mirrored from the far corner of
  // the (rows_total x n) block.  best[j]: any ending, gap[j]: ending with query symbols against a gap
  void run(int64_t a0, int64_t b0, int64_t rows_total, int64_t n, int64_t rows, bool forward, int64_t edge,
           int64_t* best, int64_t* gap) const
  {
    best[0] = 0;
    int64_t t = -open;
    for (int64_t j = 1; j <= n; ++j) {
      t -= ext;
      best[j] = t;
      gap[j] = t - open;
    }
    t = -edge;
    for (int64_t i = 1; i <= rows; ++i) {
      const int64_t ai = forward ? a0 + i - 1 : a0 + rows_total - i;
      int64_t diag = best[0];
      t -= ext;
      best[0] = t;
      int64_t h = t, f = t - open;
      for (int64_t j = 1; j <= n; ++j) {
        const int64_t bj = forward ? b0 + j - 1 : b0 + n - j;
        f = std::max(f, h - open) - ext;
        const int64_t e = std::max(gap[j], best[j] - open) - ext;
        gap[j] = e;
        h = std::max(std::max(diag + sub(ai, bj), f), e);
        diag = best[j];
        best[j] = h;
      }
    }
    gap[0] = best[0];
  }
};

struct Task {
  int64_t a0, b0, m, n, lead, trail;
  int64_t literal;   // > 0: emit that many 'D' instead of solving a block
};

void append(std::vector<EditOp>& ops, char kind, int64_t count)
{
  if (count <= 0) return;
  if (!ops.empty() && ops.back().kind == kind) ops.back().count += count;
  else ops.push_back({kind, count});
}

void one_query_symbol(const Sweeper& s, const Task& k, std::vector<EditOp>& ops)      // align.cc:260-328
{
  const int64_t n = k.n;
  int64_t at, top;
  if (k.lead <= k.trail) { at = -1; top = -k.lead - (1 + n) * s.ext - s.open; }
  else                   { at = n;  top = -s.open - (1 + n) * s.ext - k.trail; }
  for (int64_t j = 0; j < n; ++j) {
    int64_t v = s.sub(k.a0, k.b0 + j) - s.ext * (n - 1);
    if (j > 0) v -= s.open;
    if (j + 1 < n) v -= s.open;
    if (v > top) { top = v; at = j; }
  }
  if (at < 0) { append(ops, 'D', 1); append(ops, 'I', n); }
  else if (at == n) { append(ops, 'I', n); append(ops, 'D', 1); }
  else { append(ops, 'I', at); append(ops, 'M', 1); append(ops, 'I', n - 1 - at); }
}

}  // namespace

void edit_script(const uint8_t* query, const uint8_t* dseq, const int32_t* matrix, int64_t gapopen, int64_t gapextend,
                 int64_t q_start, int64_t d_start, int64_t q_end, int64_t d_end, std::vector<EditOp>& ops)
{
  const Sweeper s{query, dseq, matrix, gapopen, gapextend};
  std::vector<Task> stack;
  stack.push_back({q_start, d_start, q_end - q_start + 1, d_end - d_start + 1, gapopen, gapopen, 0});
  std::vector<int64_t> buf;
  while (!stack.empty()) {
    const Task k = stack.back();
    stack.pop_back();
    if (k.literal > 0) { append(ops, 'D', k.literal); continue; }
    if (k.n == 0) { append(ops, 'D', k.m); continue; }
    if (k.m == 0) { append(ops, 'I', k.n); continue; }
    if (k.m == 1) { one_query_symbol(s, k, ops); continue; }
    const int64_t half = k.m / 2, w = k.n + 1;
    buf.resize(size_t(4 * w));
    int64_t *fb = buf.data(), *fg = fb + w, *rb = fg + w, *rg = rb + w;
    s.run(k.a0, k.b0, k.m, k.n, half, true, k.lead, fb, fg);
    s.run(k.a0, k.b0, k.m, k.n, k.m - half, false, k.trail, rb, rg);
    int64_t cut = 0, top = fb[0] + rb[k.n];
    bool via_gap = false;
    for (int64_t j = 1; j <= k.n; ++j)                      // align.cc:423-432
      if (fb[j] + rb[k.n - j] > top) { top = fb[j] + rb[k.n - j]; cut = j; }
    for (int64_t j = 0; j <= k.n; ++j)                      // align.cc:437-446
      if (fg[j] + rg[k.n - j] + s.open >= top) { top = fg[j] + rg[k.n - j] + s.open; cut = j; via_gap = true; }
    // children go on the stack right part first, so the left part is spelled out first
    if (!via_gap) {
      stack.push_back({k.a0 + half, k.b0 + cut, k.m - half, k.n - cut, s.open, k.trail, 0});
      stack.push_back({k.a0, k.b0, half, cut, k.lead, s.open, 0});
    } else {
      stack.push_back({k.a0 + half + 1, k.b0 + cut, k.m - half - 1, k.n - cut, 0, k.trail, 0});
      stack.push_back({0, 0, 0, 0, 0, 0, 2});
      stack.push_back({k.a0, k.b0, half - 1, cut, k.lead, 0, 0});
    }
  }
}

int64_t forward_end(const uint8_t* query, int64_t qlen, const uint8_t* dseq, int64_t dlen, const int32_t* matrix,
                    int64_t gapopen, int64_t gapextend, int64_t* q_end, int64_t* d_end)
{
  std::vector<int64_t> hh(size_t(dlen), 0), ee(size_t(dlen), -gapopen);
  int64_t score = 0;
  for (int64_t i = 0; i < qlen; ++i) {
    int64_t h = 0, diag = 0, f = -gapopen;
    for (int64_t j = 0; j < dlen; ++j) {
      f = std::max(f, h - gapopen) - gapextend;
      const int64_t e = std::max(ee[size_t(j)], hh[size_t(j)] - gapopen) - gapextend;
      ee[size_t(j)] = e;
      h = std::max<int64_t>(diag + matrix[(int(dseq[j]) << 5) + query[i]], 0);
      h = std::max(std::max(h, f), e);
      diag = hh[size_t(j)];
      hh[size_t(j)] = h;
      if (h > score) { score = h; *q_end = i; *d_end = j; }
    }
  }
  return score;
}

bool backward_start(const uint8_t* query, const uint8_t* dseq, const int32_t* matrix, int64_t gapopen,
                    int64_t gapextend, int64_t score, int64_t q_end, int64_t d_end, int64_t* q_start, int64_t* d_start)
{
  std::vector<int64_t> hh(size_t(d_end + 1), -1), ee(size_t(d_end + 1), -1);
  int64_t reached = 0;
  for (int64_t i = q_end; i >= 0; --i) {
    int64_t h = -1, f = -1, diag = (i == q_end) ? 0 : -1;
    for (int64_t j = d_end; j >= 0; --j) {
      f = std::max(f, h - gapopen) - gapextend;
      const int64_t e = std::max(ee[size_t(j)], hh[size_t(j)] - gapopen) - gapextend;
      ee[size_t(j)] = e;
      h = std::max(std::max(diag + matrix[(int(dseq[j]) << 5) + query[i]], f), e);
      diag = hh[size_t(j)];
      hh[size_t(j)] = h;
      if (h > reached) {
        reached = h;
        *q_start = i;
        *d_start = j;
        if (reached >= score) return true;
      }
    }
  }
  return false;
}

void count_columns(const uint8_t* query, const uint8_t* dseq, const int32_t* matrix, int64_t q_start, int64_t d_start,
                   const std::vector<EditOp>& ops, int64_t* identities, int64_t* positives, int64_t* indels,
                   int64_t* aligned, int64_t* gaps)
{
  int64_t qi = q_start, dj = d_start;
  *identities = *positives = *indels = *aligned = *gaps = 0;
  for (const EditOp& op : ops) {                              // count_align, hits.cc:1073-1109
    *aligned += op.count;
    if (op.kind == 'D') { *gaps += 1; *indels += op.count; qi += op.count; }
    else if (op.kind == 'I') { *gaps += 1; *indels += op.count; dj += op.count; }
    else {
      for (int64_t k = 0; k < op.count; ++k, ++qi, ++dj) {
        const int qs = query[qi], ds = dseq[dj];
        if (qs == ds) { ++*identities; ++*positives; }
        else if (matrix[32 * qs + ds] > 0) ++*positives;      // index order as at hits.cc:1105
      }
    }
  }
}

}  // namespace swa
