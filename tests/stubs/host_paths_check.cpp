// TEST DRIVER, not product code: the host paths of the single-shard C ABI that the command-line driver does not reach -
// window views of long sequences, multi-pass launches, the re-queue variants, inclusion subsets, streamed handles, end
// points, the alignment phase with a too-small text buffer, two strands / two queries per pass, translated shards -
// exercised through include/swipe_amd.h on a GPU and checked WITHOUT an oracle, by self-consistency: every option
// variant must reproduce the scores of the default search, every hit list must be the ordered top of those scores, a
// streamed handle must answer like the resident one.  Built for running under AddressSanitizer + UBSan (statically
// linked with the instrumented host objects by `make -C swipe_amd/csrc asan`, tools/asan_cli.sh runs it); it links
// against the plain library too.  Exit 0 = every comparison held (and, instrumented, no sanitizer report).
#include "../../include/swipe_amd.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

static int bad = 0, checks = 0;
#define EXPECT(c) do { ++checks; if (!(c)) { std::fprintf(stderr, "FAILED %s:%d: %s   [%s]\n", __FILE__, __LINE__, #c, swa_last_error()); ++bad; } } while (0)

struct Db { std::vector<uint8_t> res; std::vector<int64_t> off; int64_t nseq() const { return int64_t(off.size()) - 1; } };

static void append(Db& d, const std::vector<uint8_t>& s) { d.res.insert(d.res.end(), s.begin(), s.end()); d.off.push_back(int64_t(d.res.size())); }
static std::vector<uint8_t> random_seq(std::mt19937_64& rng, size_t n, bool protein)
{
  std::vector<uint8_t> s(n);
  static const uint8_t aa[20] = {1, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 22};   // the 20 standard letters
  for (auto& c : s) c = protein ? aa[rng() % 20] : uint8_t(1u << (rng() % 4));
  return s;
}
// a copy of q[from, to) with substitutions, placed inside noise
static std::vector<uint8_t> homolog(std::mt19937_64& rng, const std::vector<uint8_t>& q, size_t from, size_t to, size_t before, size_t after, bool protein)
{
  std::vector<uint8_t> s = random_seq(rng, before, protein);
  for (size_t i = from; i < to; ++i) s.push_back(rng() % 7 == 0 ? random_seq(rng, 1, protein)[0] : q[i]);
  const std::vector<uint8_t> t = random_seq(rng, after, protein);
  s.insert(s.end(), t.begin(), t.end());
  return s;
}

static std::vector<swa_hit_t> top_of(const std::vector<int64_t>& scores, int64_t first, int64_t keep, int64_t lo, int64_t hi, int64_t* total, int64_t* obvious)
{
  std::vector<swa_hit_t> all;
  *total = *obvious = 0;
  for (size_t i = 0; i < scores.size(); ++i) {
    if (scores[i] > hi) ++*obvious;                               // hits.cc:174-178
    if (scores[i] >= lo) ++*total;
    if (scores[i] >= lo && scores[i] <= hi) all.push_back(swa_hit_t{first + int64_t(i), scores[i]});
  }
  std::sort(all.begin(), all.end(), [](const swa_hit_t& a, const swa_hit_t& b) { return a.score > b.score || (a.score == b.score && a.seqno > b.seqno); });
  if (int64_t(all.size()) > keep) all.resize(size_t(keep));
  return all;
}
static bool same_hits(const std::vector<swa_hit_t>& a, const swa_hit_t* b, int64_t nb)
{
  if (int64_t(a.size()) != nb) return false;
  for (size_t i = 0; i < a.size(); ++i) if (a[i].seqno != b[i].seqno || a[i].score != b[i].score) return false;
  return true;
}

static std::vector<int64_t> all_scores(swa_db* db, const std::vector<uint8_t>& q, int64_t n)
{
  std::vector<int64_t> s(size_t(n) + 1, -99);
  swa_counters_t c;
  EXPECT(swa_search(db, q.data(), int64_t(q.size()), s.data(), &c) == SWA_OK);
  EXPECT(s[size_t(n)] == -99);                                    // nothing written past the shard
  s.resize(size_t(n));
  return s;
}

static void check_topk(swa_db* db, const std::vector<uint8_t>& q, const std::vector<int64_t>& scores, int64_t first, int64_t keep, int64_t lo, int64_t hi)
{
  std::vector<swa_hit_t> got(size_t(keep) + 1);
  int64_t n = -1, tot = -1, obv = -1, wt, wo;
  swa_counters_t c;
  EXPECT(swa_search_topk(db, q.data(), int64_t(q.size()), keep, lo, hi, got.data(), &n, &tot, &obv, &c) == SWA_OK);
  const std::vector<swa_hit_t> want = top_of(scores, first, keep, lo, hi, &wt, &wo);
  EXPECT(same_hits(want, got.data(), n) && tot == wt && obv == wo);
}

static void protein_shard(int device, std::mt19937_64& rng)
{
  const std::vector<uint8_t> q = random_seq(rng, 200, true), q2 = random_seq(rng, 150, true);
  Db d;
  d.off.push_back(0);
  for (int s = 0; s < 3000; ++s) {
    if (s % 97 == 5) append(d, homolog(rng, q, rng() % 50, 120 + rng() % 80, rng() % 100, rng() % 100, true));
    else if (s % 211 == 7) append(d, homolog(rng, q2, 0, 150, rng() % 60, rng() % 60, true));
    else if (s == 1000) append(d, std::vector<uint8_t>());           // an empty sequence
    else append(d, random_seq(rng, 10 + rng() % 590, true));
  }
  for (int k = 0; k < 3; ++k) append(d, homolog(rng, q, 0, 200, 9000 + 3000 * size_t(k), 4000, true));   // long: windows
  const int64_t n = d.nseq(), first = 1000;
  int64_t M[1024];
  EXPECT(swa_matrix_builtin("BLOSUM62", M) == SWA_OK);

  swa_db* db = nullptr;
  EXPECT(swa_db_from_memory(d.res.data(), d.off.data(), n, SWA_SYMTYPE_PROTEIN, device, first, 0, 0, &db) == SWA_OK);
  if (!db) return;
  std::vector<int64_t> tmp(static_cast<size_t>(n), 0);
  EXPECT(swa_search(db, q.data(), 200, tmp.data(), nullptr) == SWA_ESTATE);                 // search before set_scoring
  EXPECT(swa_set_scoring(db, M, 12, 1) == SWA_OK);
  swa_db_info_t info;
  EXPECT(swa_db_info(db, &info) == SWA_OK && info.seqcount == n && info.first_seqno == first && info.symcount == d.off.back());
  const std::vector<int64_t> s0 = all_scores(db, q, n);
  EXPECT(*std::max_element(s0.begin(), s0.end()) > 300);                                      // the planted copies are found

  // every option variant reproduces the default scores
  const char* variants[][2] = {{"window", "0"}, {"window", "3000"}, {"window", "1500"}, {"force_mp", "1"}, {"requeue_host", "1"},
                               {"requeue_follow", "0"}, {"wave_requeue", "0"}, {"narrow_variant", "1"}, {"lanes", "16"}, {"lanes", "4"},
                               {"pipe", "0"}, {"blocks_per_cu", "2"}, {"boundary_mb", "1"}};
  for (auto& v : variants) {
    EXPECT(swa_set_option(db, v[0], v[1]) == SWA_OK);
    const std::vector<int64_t> s = all_scores(db, q, n);
    if (s != s0) { ++bad; std::fprintf(stderr, "FAILED: option %s=%s changes scores\n", v[0], v[1]); }
    ++checks;
    check_topk(db, q, s0, first, 40, 45, 1000000);
    EXPECT(swa_set_option(db, v[0], nullptr) == SWA_OK);
  }
  EXPECT(swa_set_option(db, "no_such_key", "1") == SWA_EINVAL && swa_set_option(db, "window", "x") == SWA_EINVAL);
  // window views together with the multi-pass kernel (the hand-over buffer of a view)
  EXPECT(swa_set_option(db, "window", "2000") == SWA_OK && swa_set_option(db, "force_mp", "1") == SWA_OK);
  EXPECT(all_scores(db, q, n) == s0);
  EXPECT(swa_set_option(db, "window", nullptr) == SWA_OK && swa_set_option(db, "force_mp", nullptr) == SWA_OK);

  // hit lists: auto, bound build forced, exact first pass forced; keep 0 / 1 / more than there are; a score window
  for (const char* b : {"-1", "1", "0"}) {
    EXPECT(swa_set_option(db, "bound", b) == SWA_OK);
    check_topk(db, q, s0, first, 50, 40, 1000000);
    check_topk(db, q, s0, first, 1, 60, 1000000);
    check_topk(db, q, s0, first, 0, 40, 1000000);
    check_topk(db, q, s0, first, 5000, 25, 1000000);               // permissive: more hits than the eager candidates
    check_topk(db, q, s0, first, 30, 40, 400);                      // "obvious" hits above the window
  }
  EXPECT(swa_set_option(db, "bound", nullptr) == SWA_OK);

  // two different queries in one pass = two searches
  {
    const std::vector<int64_t> sb = all_scores(db, q2, n);
    std::vector<swa_hit_t> h1(60), h2(60);
    int64_t n1, n2, t1, t2, o1, o2, wt, wo;
    swa_counters_t c;
    EXPECT(swa_search_pair_topk(db, q.data(), 200, q2.data(), 150, 50, 40, 1000000, 20, 35, 1000000, h1.data(), &n1, &t1, &o1, h2.data(), &n2, &t2, &o2, &c) == SWA_OK);
    EXPECT(same_hits(top_of(s0, first, 50, 40, 1000000, &wt, &wo), h1.data(), n1) && t1 == wt);
    EXPECT(same_hits(top_of(sb, first, 20, 35, 1000000, &wt, &wo), h2.data(), n2) && t2 == wt);
  }

  // inclusion subsets
  {
    std::vector<uint8_t> inc(static_cast<size_t>(n));
    for (auto& x : inc) x = rng() % 3 != 0;
    EXPECT(swa_db_set_inclusion(db, inc.data(), n) == SWA_OK);
    EXPECT(swa_db_set_inclusion(db, inc.data(), n - 1) == SWA_EINVAL);
    const std::vector<int64_t> s = all_scores(db, q, n);
    bool ok = true;
    for (size_t i = 0; i < s.size(); ++i) ok = ok && s[i] == (inc[i] ? s0[i] : -1);
    EXPECT(ok);
    std::vector<int64_t> masked = s0;
    for (size_t i = 0; i < masked.size(); ++i) if (!inc[i]) masked[i] = -1;
    check_topk(db, q, masked, first, 50, 40, 1000000);
    EXPECT(swa_db_set_inclusion(db, nullptr, 0) == SWA_OK);
    EXPECT(all_scores(db, q, n) == s0);
  }

  // end points and the alignment phase of the best hits; a text buffer that is too small reports what it needs
  {
    int64_t wt, wo;
    const std::vector<swa_hit_t> top = top_of(s0, first, 12, 40, 1000000, &wt, &wo);
    std::vector<int64_t> ids, sc(top.size()), bp(top.size()), bq(top.size());
    for (const swa_hit_t& h : top) ids.push_back(h.seqno);
    EXPECT(swa_search_endpoints(db, q.data(), 200, ids.data(), int64_t(ids.size()), sc.data(), bp.data(), bq.data()) == SWA_OK);
    for (size_t i = 0; i < top.size(); ++i) {
      const int64_t len = d.off[size_t(ids[i] - first) + 1] - d.off[size_t(ids[i] - first)];
      EXPECT(sc[i] == top[i].score && bp[i] >= 0 && bp[i] < len && bq[i] >= 0 && bq[i] < 200);
    }
    const int64_t outside = first + n;
    EXPECT(swa_search_endpoints(db, q.data(), 200, &outside, 1, sc.data(), bp.data(), bq.data()) != SWA_OK);
    std::vector<swa_alignment_t> al(top.size());
    char tiny[4];
    int64_t used = 0;
    EXPECT(swa_align_hits(db, q.data(), 200, ids.data(), nullptr, nullptr, int64_t(ids.size()), al.data(), tiny, sizeof tiny, &used) == SWA_ERANGE && used > 4);
    std::vector<char> text(static_cast<size_t>(used));
    EXPECT(swa_align_hits(db, q.data(), 200, ids.data(), nullptr, nullptr, int64_t(ids.size()), al.data(), text.data(), used, &used) == SWA_OK);
    for (size_t i = 0; i < top.size(); ++i) {
      EXPECT(al[i].score == top[i].score && al[i].seqno == ids[i] && al[i].q_start <= al[i].q_end && al[i].q_end < 200 && al[i].d_start <= al[i].d_end);
      const bool inside = al[i].cigar_offset >= 0 && al[i].cigar_len > 0 && al[i].cigar_offset + al[i].cigar_len < used;
      EXPECT(inside && text[size_t(al[i].cigar_offset + al[i].cigar_len)] == '\0' && text[size_t(al[i].cigar_offset)] == 'M');
    }
    // db_getsequence out of the shard
    std::vector<uint8_t> buf(64);
    int64_t len = 0;
    if (ids.empty()) { ++bad; swa_db_close(db); return; }
    const size_t k = size_t(ids[0] - first);
    EXPECT(swa_db_sequence(db, ids[0], 0, 0, buf.data(), 8, &len, nullptr) == SWA_ERANGE && len == d.off[k + 1] - d.off[k]);
    buf.resize(size_t(std::max<int64_t>(len, 1)));
    EXPECT(swa_db_sequence(db, ids[0], 0, 0, buf.data(), int64_t(buf.size()), &len, nullptr) == SWA_OK && len == d.off[k + 1] - d.off[k] && !std::memcmp(buf.data(), d.res.data() + d.off[k], size_t(len)));
  }

  // queries of other shapes on the same handle: one row, a lane's worth, more than one pass of the 16-lane kernel
  for (size_t ql : {size_t(1), size_t(48), size_t(49), size_t(1100)}) {
    const std::vector<uint8_t> qq = ql <= 200 ? std::vector<uint8_t>(q.begin(), q.begin() + long(ql)) : random_seq(rng, ql, true);
    const std::vector<int64_t> a = all_scores(db, qq, n);
    EXPECT(swa_set_option(db, "window", "0") == SWA_OK && swa_set_option(db, "lanes", "16") == SWA_OK);
    EXPECT(all_scores(db, qq, n) == a);
    EXPECT(swa_set_option(db, "window", nullptr) == SWA_OK && swa_set_option(db, "lanes", nullptr) == SWA_OK);
    check_topk(db, qq, a, first, 25, ql == 1 ? 1 : 30, 1000000);
  }

  // a streamed handle over the same arrays answers like the resident one
  {
    Db big = d;
    while (big.res.size() < (40u << 20)) append(big, random_seq(rng, 100 + rng() % 500, true));
    const int64_t nb = big.nseq();
    swa_db *res = nullptr, *str = nullptr;
    EXPECT(swa_db_from_memory(big.res.data(), big.off.data(), nb, SWA_SYMTYPE_PROTEIN, device, 0, 0, 0, &res) == SWA_OK);
    EXPECT(swa_db_from_memory_streamed(big.res.data(), big.off.data(), nb, SWA_SYMTYPE_PROTEIN, device, 0, 0, 0, int64_t(32) << 20, &str) == SWA_OK);
    if (res && str) {
      EXPECT(swa_set_scoring(res, M, 12, 1) == SWA_OK && swa_set_scoring(str, M, 12, 1) == SWA_OK);
      const std::vector<int64_t> a = all_scores(res, q, nb);
      swa_db_info_t ri, si;
      EXPECT(swa_db_info(res, &ri) == SWA_OK && swa_db_info(str, &si) == SWA_OK && si.seqcount == nb && ri.seqcount == nb);
      EXPECT(all_scores(str, q, nb) == a);
      check_topk(str, q, a, 0, 50, 40, 1000000);
      check_topk(str, q, a, 0, 3, 40, 300);
      std::vector<swa_hit_t> h1(60), h2(60);
      int64_t n1, n2, t1, t2, o1, o2, wt, wo;
      swa_counters_t c;
      EXPECT(swa_search_pair_topk(str, q.data(), 200, q2.data(), 150, 50, 40, 1000000, 20, 35, 1000000, h1.data(), &n1, &t1, &o1, h2.data(), &n2, &t2, &o2, &c) == SWA_OK);
      EXPECT(same_hits(top_of(a, 0, 50, 40, 1000000, &wt, &wo), h1.data(), n1) && t1 == wt);
      // a streamed handle answers the entry points that name sequences too (round 4: the owning part is bound to a slot)
      int64_t ids[3] = {0, nb / 2, nb - 1}, s1[3], p1[3], q1[3], s2[3], p2[3], q2x[3];
      EXPECT(swa_search_endpoints(res, q.data(), 200, ids, 3, s1, p1, q1) == SWA_OK);
      EXPECT(swa_search_endpoints(str, q.data(), 200, ids, 3, s2, p2, q2x) == SWA_OK);
      for (int k = 0; k < 3; ++k) EXPECT(s1[k] == s2[k] && p1[k] == p2[k] && q1[k] == q2x[k]);
      // ... and takes an inclusion set (round 5): the even sequences only
      std::vector<uint8_t> inc(static_cast<size_t>(nb));
      for (int64_t k = 0; k < nb; ++k) inc[size_t(k)] = uint8_t(k % 2 == 0);
      EXPECT(swa_db_set_inclusion(res, inc.data(), nb) == SWA_OK && swa_db_set_inclusion(str, inc.data(), nb) == SWA_OK);
      std::vector<swa_hit_t> g1(40), g2(40);
      int64_t m1 = 0, m2 = 0, u1 = 0, u2 = 0, v1 = 0, v2 = 0;
      EXPECT(swa_search_topk(res, q.data(), 200, 40, 35, 1000000, g1.data(), &m1, &u1, &v1, &c) == SWA_OK);
      EXPECT(swa_search_topk(str, q.data(), 200, 40, 35, 1000000, g2.data(), &m2, &u2, &v2, &c) == SWA_OK);
      EXPECT(m1 == m2 && u1 == u2 && same_hits(std::vector<swa_hit_t>(g1.begin(), g1.begin() + m1), g2.data(), m2));
      for (int64_t k = 0; k < m2; ++k) EXPECT(g2[size_t(k)].seqno % 2 == 0);
    }
    swa_db_close(res);
    swa_db_close(str);
  }
  swa_db_close(db);

  // arguments that must be refused
  swa_db* none = nullptr;
  std::vector<uint8_t> wrong = d.res;
  wrong[wrong.size() / 2] = 40;                                                               // not a residue code
  EXPECT(swa_db_from_memory(wrong.data(), d.off.data(), n, SWA_SYMTYPE_PROTEIN, device, 0, 0, 0, &none) == SWA_EINVAL && !none);
  std::vector<int64_t> down = d.off;
  std::swap(down[10], down[11]);
  EXPECT(swa_db_from_memory(d.res.data(), down.data(), n, SWA_SYMTYPE_PROTEIN, device, 0, 0, 0, &none) == SWA_EINVAL && !none);
  EXPECT(swa_db_from_memory(d.res.data(), d.off.data(), n, SWA_SYMTYPE_PROTEIN, 9999, 0, 0, 0, &none) != SWA_OK && !none);
  const int64_t zero = 0;
  EXPECT(swa_db_from_memory(nullptr, &zero, 0, SWA_SYMTYPE_PROTEIN, device, 0, 0, 0, &none) == SWA_OK && none);   // an empty shard searches
  if (none) {
    EXPECT(swa_set_scoring(none, M, 12, 1) == SWA_OK);
    std::vector<swa_hit_t> h(4);
    int64_t nh = -1, t = -1, o = -1;
    EXPECT(swa_search_topk(none, q.data(), 200, 4, 1, 1000, h.data(), &nh, &t, &o, nullptr) == SWA_OK && nh == 0 && t == 0);
    swa_db_close(none);
  }
}

static void nucleotide_shard(int device, std::mt19937_64& rng)
{
  const std::vector<uint8_t> q = random_seq(rng, 300, false);
  std::vector<uint8_t> rc(q.rbegin(), q.rend());
  static const uint8_t compl4[16] = {0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15};
  for (auto& c : rc) c = compl4[c];
  Db d;
  d.off.push_back(0);
  for (int s = 0; s < 2500; ++s) {
    if (s % 83 == 3) append(d, homolog(rng, q, 20, 280, rng() % 90, rng() % 90, false));
    else if (s % 89 == 4) append(d, homolog(rng, rc, 10, 250, rng() % 90, rng() % 90, false));
    else append(d, random_seq(rng, 1 + rng() % 800, false));
  }
  append(d, homolog(rng, q, 0, 300, 150000, 50000, false));                                    // a long subject: windows on both strands
  const int64_t n = d.nseq();
  int64_t M[1024];
  EXPECT(swa_matrix_nucleotide(1, -3, M) == SWA_OK);
  swa_db* db = nullptr;
  EXPECT(swa_db_from_memory(d.res.data(), d.off.data(), n, SWA_SYMTYPE_NUCLEOTIDE, device, 0, 0, 0, &db) == SWA_OK);
  if (!db) return;
  EXPECT(swa_set_scoring(db, M, 7, 2) == SWA_OK);
  const std::vector<int64_t> a = all_scores(db, q, n), b = all_scores(db, rc, n);
  std::vector<int64_t> s1(static_cast<size_t>(n)), s2(static_cast<size_t>(n));
  swa_counters_t c;
  EXPECT(swa_search2(db, q.data(), rc.data(), 300, s1.data(), s2.data(), &c) == SWA_OK && s1 == a && s2 == b);
  EXPECT(swa_set_option(db, "window", "0") == SWA_OK);
  EXPECT(swa_search2(db, q.data(), rc.data(), 300, s1.data(), s2.data(), &c) == SWA_OK && s1 == a && s2 == b);
  EXPECT(swa_set_option(db, "window", nullptr) == SWA_OK);
  // both strands in one list: entries of one (score, seqno) keep the order plus, minus
  std::vector<swa_hit_t> h(80);
  std::vector<int32_t> which(80);
  int64_t nh = 0, tot = 0, obv = 0, ta, tb, oa, ob;
  EXPECT(swa_search2_topk(db, q.data(), rc.data(), 300, 80, 30, 1000000, h.data(), which.data(), &nh, &tot, &obv, &c) == SWA_OK);
  top_of(a, 0, 0, 30, 1000000, &ta, &oa);
  top_of(b, 0, 0, 30, 1000000, &tb, &ob);
  EXPECT(tot == ta + tb && nh == std::min<int64_t>(80, tot));
  bool ok = true;
  for (int64_t i = 0; i < nh; ++i) {
    if (h[size_t(i)].seqno < 0 || h[size_t(i)].seqno >= n) { ok = false; break; }
    ok = ok && h[size_t(i)].score == (which[size_t(i)] ? b : a)[size_t(h[size_t(i)].seqno)];
    if (i) ok = ok && (h[size_t(i - 1)].score > h[size_t(i)].score || (h[size_t(i - 1)].score == h[size_t(i)].score && h[size_t(i - 1)].seqno >= h[size_t(i)].seqno));
  }
  EXPECT(ok);
  // minus-strand end points: the reverse complement of the subject against the plus query
  int64_t wt, wo;
  const std::vector<swa_hit_t> topb = top_of(b, 0, 5, 30, 1000000, &wt, &wo);
  for (const swa_hit_t& x : topb) {
    const int32_t one = 1;
    int64_t sc = -1, bp = -1, bq = -1;
    EXPECT(swa_search_endpoints_strand(db, q.data(), 300, &x.seqno, &one, nullptr, 1, &sc, &bp, &bq) == SWA_OK && sc == x.score);
  }
  swa_db_close(db);

  // the same shard as six translated frames: 6 scores per sequence, frame hits consistent with them
  const std::vector<uint8_t> pq = random_seq(rng, 60, true);
  Db t;
  t.off.push_back(0);
  for (int s = 0; s < 400; ++s) append(t, random_seq(rng, s == 7 ? 0 : 2 + rng() % 900, false));
  int64_t B[1024];
  EXPECT(swa_matrix_builtin("BLOSUM62", B) == SWA_OK);
  swa_db* tr = nullptr;
  EXPECT(swa_db_from_memory_translated(t.res.data(), t.off.data(), t.nseq(), 1, device, 50, 0, 0, &tr) == SWA_OK);
  if (!tr) return;
  EXPECT(swa_set_scoring(tr, B, 12, 1) == SWA_OK);
  const std::vector<int64_t> f = all_scores(tr, pq, 6 * t.nseq());
  const uint8_t* qs[1] = {pq.data()};
  const int64_t ql[1] = {60};
  const int32_t tag[1] = {0};
  std::vector<swa_fhit_t> fh(40);
  int64_t nf = 0, tf = 0, of = 0;
  EXPECT(swa_search_frames_topk(tr, 1, qs, ql, tag, 40, 20, 1000000, fh.data(), &nf, &tf, &of, &c) == SWA_OK);
  int64_t want_total = 0;
  for (int64_t v : f) want_total += v >= 20;
  EXPECT(tf == want_total && nf == std::min<int64_t>(40, want_total));
  ok = true;
  for (int64_t i = 0; i < nf; ++i) {
    const swa_fhit_t& x = fh[size_t(i)];
    const int64_t at = 6 * (x.seqno - 50) + 3 * x.dstrand + x.dframe;
    if (at < 0 || at >= int64_t(f.size())) { ok = false; break; }
    ok = ok && x.score == f[size_t(at)];
    if (i) ok = ok && fh[size_t(i - 1)].score >= x.score;
  }
  EXPECT(ok);
  swa_db_close(tr);
}

// the same arrays through BLAST v4 volumes on disk: whole, a range, streamed, six translated frames
static void files_shard(int device, std::mt19937_64& rng, const std::string& dir)
{
  const std::vector<uint8_t> q = random_seq(rng, 120, true);
  Db d;
  d.off.push_back(0);
  for (int s = 0; s < 60000; ++s) append(d, s % 611 == 2 ? homolog(rng, q, 0, 120, rng() % 40, rng() % 40, true) : random_seq(rng, s == 9 ? 0 : 5 + rng() % 800, true));   // 24 MB: streams under a 24 MiB budget
  const int64_t n = d.nseq();
  const std::string base = dir + "/hp_aa";
  EXPECT(swa_blastdb_write(base.c_str(), SWA_SYMTYPE_PROTEIN, d.res.data(), d.off.data(), n, 1, "host paths") == SWA_OK);
  int64_t M[1024];
  EXPECT(swa_matrix_builtin("BLOSUM62", M) == SWA_OK);
  swa_db *mem = nullptr, *whole = nullptr, *part = nullptr, *str = nullptr, *none = nullptr;
  EXPECT(swa_db_from_memory(d.res.data(), d.off.data(), n, SWA_SYMTYPE_PROTEIN, device, 0, 0, 0, &mem) == SWA_OK);
  EXPECT(swa_db_open(base.c_str(), SWA_SYMTYPE_PROTEIN, device, 0, -1, &whole) == SWA_OK);
  EXPECT(swa_db_open(base.c_str(), SWA_SYMTYPE_PROTEIN, device, 300, 899, &part) == SWA_OK);
  EXPECT(swa_db_open_streamed(base.c_str(), SWA_SYMTYPE_PROTEIN, device, 0, -1, int64_t(24) << 20, &str) == SWA_OK);
  EXPECT(swa_db_open_streamed(base.c_str(), SWA_SYMTYPE_PROTEIN, device, 0, -1, int64_t(1) << 20, &none) == SWA_ENOMEM && !none);   // no room for one sequence
  EXPECT(swa_db_open((base + "_missing").c_str(), SWA_SYMTYPE_PROTEIN, device, 0, -1, &none) == SWA_EIO && !none);
  if (mem && whole && part && str) {
    for (swa_db* h : {mem, whole, part, str}) EXPECT(swa_set_scoring(h, M, 12, 1) == SWA_OK);
    const std::vector<int64_t> a = all_scores(mem, q, n);
    EXPECT(all_scores(whole, q, n) == a);
    EXPECT(all_scores(str, q, n) == a);
    const std::vector<int64_t> p = all_scores(part, q, 600);
    EXPECT(std::equal(p.begin(), p.end(), a.begin() + 300));
    swa_db_info_t pi;
    EXPECT(swa_db_info(part, &pi) == SWA_OK && pi.first_seqno == 300 && pi.seqcount == 600 && pi.total_seqcount == n);
    check_topk(part, q, p, 300, 20, 35, 1000000);
    check_topk(str, q, a, 0, 20, 35, 1000000);
  }
  for (swa_db* h : {mem, whole, part, str}) swa_db_close(h);

  // nucleotide volumes opened as six translated frames = the arrays translated in memory
  Db t;
  t.off.push_back(0);
  for (int s = 0; s < 300; ++s) append(t, random_seq(rng, 3 + rng() % 700, false));
  const std::string nb = dir + "/hp_nt";
  EXPECT(swa_blastdb_write(nb.c_str(), SWA_SYMTYPE_NUCLEOTIDE, t.res.data(), t.off.data(), t.nseq(), 1, "host paths nt") == SWA_OK);
  swa_db *tm = nullptr, *tf = nullptr;
  EXPECT(swa_db_from_memory_translated(t.res.data(), t.off.data(), t.nseq(), 11, device, 0, 0, 0, &tm) == SWA_OK);
  EXPECT(swa_db_open_translated(nb.c_str(), 11, device, 0, -1, &tf) == SWA_OK);
  if (tm && tf) {
    EXPECT(swa_set_scoring(tm, M, 12, 1) == SWA_OK && swa_set_scoring(tf, M, 12, 1) == SWA_OK);
    const std::vector<uint8_t> pq = random_seq(rng, 40, true);
    EXPECT(all_scores(tm, pq, 6 * t.nseq()) == all_scores(tf, pq, 6 * t.nseq()));
  }
  swa_db_close(tm);
  swa_db_close(tf);
  for (const char* e : {".pin", ".psq", ".phr"}) std::remove((base + e).c_str());
  for (const char* e : {".nin", ".nsq", ".nhr"}) std::remove((nb + e).c_str());
}

int main(int argc, char** argv)
{
  const int device = argc > 1 ? std::atoi(argv[1]) : 0;
  const uint64_t seed = argc > 2 ? uint64_t(std::atoll(argv[2])) : 20260930;
  const int rounds = argc > 3 ? std::atoi(argv[3]) : 1;
  const std::string dir = argc > 4 ? argv[4] : "/tmp";
  if (swa_device_count() < 1) { std::fprintf(stderr, "host paths check: no HIP device (there is no CPU fallback)\n"); return 2; }
  for (int r = 0; r < rounds; ++r) {
    std::mt19937_64 rng(seed + uint64_t(r));
    protein_shard(device, rng);
    nucleotide_shard(device, rng);
    files_shard(device, rng, dir);
  }
  std::printf("host paths check: %d rounds from seed %llu, %d comparisons, %d bad\n", rounds, (unsigned long long)seed, checks, bad);
  return bad == 0 ? 0 : 1;
}
