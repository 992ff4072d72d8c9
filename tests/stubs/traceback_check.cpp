// TEST DRIVER, not product code: the host half of the alignment phase (swipe_amd/csrc/traceback.cpp: forward / backward
// sweeps of region(), Myers-Miller diff(), align.cc:70-467) under AddressSanitizer + UndefinedBehaviorSanitizer on
// random sequence pairs with planted, mutated and gapped copies.  Checked without any second implementation: the edit
// script, re-scored column by column with the affine gap cost, must give exactly the score of the forward sweep; it
// must start and end on a matched column pair and span the two cells the sweeps returned.
//   usage: traceback_check <pairs> <seed>
#include "../../swipe_amd/csrc/traceback.h"

#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

int main(int argc, char** argv)
{
  const int pairs = argc > 1 ? std::atoi(argv[1]) : 2000;
  std::mt19937_64 rng(argc > 2 ? uint64_t(std::atoll(argv[2])) : 1);
  int bad = 0, aligned_pairs = 0;
  for (int t = 0; t < pairs; ++t) {
    // a random scoring system: diagonal 1..9, off-diagonal -6..2, sometimes asymmetric; gaps 1..14 + 1..4
    std::vector<int32_t> M(1024, -1);
    const int nsym = 4 + int(rng() % 21);
    const bool asym = rng() % 4 == 0;
    for (int a = 1; a <= nsym; ++a)
      for (int b = 1; b <= (asym ? nsym : a); ++b) {
        const int32_t v = a == b ? 1 + int32_t(rng() % 9) : int32_t(rng() % 9) - 6;
        M[(a << 5) | b] = v;
        if (!asym) M[(b << 5) | a] = v;
      }
    const int64_t go = 1 + int64_t(rng() % 14), ge = 1 + int64_t(rng() % 4);
    const int64_t qlen = 1 + int64_t(rng() % 120);
    std::vector<uint8_t> q(static_cast<size_t>(qlen)), d;
    for (auto& c : q) c = uint8_t(1 + rng() % nsym);
    // database sequence: noise + a mutated copy of a stretch of the query with insertions and deletions + noise
    for (int i = int(rng() % 30); i > 0; --i) d.push_back(uint8_t(1 + rng() % nsym));
    const int64_t from = int64_t(rng() % qlen), to = from + int64_t(rng() % (qlen - from + 1));
    for (int64_t i = from; i < to; ++i) {
      const unsigned r = unsigned(rng() % 100);
      if (r < 6) continue;                                                  // deletion
      if (r < 12) for (int k = 1 + int(rng() % 5); k > 0; --k) d.push_back(uint8_t(1 + rng() % nsym));   // insertion
      d.push_back(r < 25 ? uint8_t(1 + rng() % nsym) : q[size_t(i)]);
    }
    for (int i = int(rng() % 30); i > 0; --i) d.push_back(uint8_t(1 + rng() % nsym));
    const int64_t dlen = int64_t(d.size());
    int64_t qe = -1, de = -1, qs = -1, ds = -1;
    const int64_t score = swa::forward_end(q.data(), qlen, d.data(), dlen, M.data(), go, ge, &qe, &de);
    if (score <= 0) continue;                                               // nothing to align (the reference reports an internal error)
    if (!swa::backward_start(q.data(), d.data(), M.data(), go, ge, score, qe, de, &qs, &ds)) { ++bad; std::fprintf(stderr, "pair %d: no start cell\n", t); continue; }
    std::vector<swa::EditOp> ops;
    swa::edit_script(q.data(), d.data(), M.data(), go, ge, qs, ds, qe, de, ops);
    int64_t s = 0, qi = qs, di = ds;
    bool ok = !ops.empty() && ops.front().kind == 'M' && ops.back().kind == 'M';
    for (const swa::EditOp& op : ops) {
      if (op.count < 1) ok = false;
      if (op.kind == 'M') for (int64_t k = 0; k < op.count && qi < qlen && di < dlen; ++k) s += M[size_t((d[size_t(di++)] << 5) | q[size_t(qi++)])];
      else if (op.kind == 'D') { s -= go + op.count * ge; qi += op.count; }
      else if (op.kind == 'I') { s -= go + op.count * ge; di += op.count; }
      else ok = false;
    }
    if (!ok || s != score || qi != qe + 1 || di != de + 1 || qs < 0 || ds < 0 || qe >= qlen || de >= dlen) {
      ++bad;
      std::fprintf(stderr, "pair %d: forward score %ld, script re-scores %ld; cells (%ld,%ld)-(%ld,%ld), script ends (%ld,%ld)\n", t, long(score), long(s),
                   long(qs), long(ds), long(qe), long(de), long(qi), long(di));
    }
    int64_t ident = 0, pos = 0, indels = 0, al = 0, gaps = 0;
    swa::count_columns(q.data(), d.data(), M.data(), qs, ds, ops, &ident, &pos, &indels, &al, &gaps);
    if (ident > pos) { ++bad; std::fprintf(stderr, "pair %d: more identities than positives (the diagonal is positive)\n", t); }
    if (al < ident || indels < gaps || al != (qe - qs + 1) + (de - ds + 1) - (al - indels)) { ++bad; std::fprintf(stderr, "pair %d: column counts inconsistent\n", t); }
    ++aligned_pairs;
  }
  std::printf("traceback check: %d pairs aligned, %d bad\n", aligned_pairs, bad);
  return bad == 0 && aligned_pairs > pairs / 2 ? 0 : 1;
}
