// TEST DRIVER (see fake_shard.cpp): groups of N stand-in shards against the one-shard group, from several caller
// threads at once.  Exit 0 = every comparison held; ThreadSanitizer makes the process fail on any data race.
#include "../../include/swipe_amd.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#include <atomic>
static std::atomic<int> bad{0};
#define EXPECT(c) do { if (!(c)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); ++bad; } } while (0)

static void run(int nseq, int nshards, int gencode, unsigned seed)
{
  std::mt19937_64 rng(seed);
  std::vector<int64_t> off(size_t(nseq) + 1, 5);
  for (int s = 0; s < nseq; ++s) off[size_t(s) + 1] = off[size_t(s)] + int64_t(rng() % 300);
  std::vector<uint8_t> res(size_t(off[size_t(nseq)]) + 1, 1);
  std::vector<int> dev(size_t(nshards), 0);
  const int one = 0;
  swa_group *g = nullptr, *ref = nullptr;
  EXPECT(swa_group_from_memory(res.data(), off.data(), nseq, 1, gencode, nshards, dev.data(), 100, 0, 0, &g) == SWA_OK);
  EXPECT(swa_group_from_memory(res.data(), off.data(), nseq, 1, gencode, 1, &one, 100, 0, 0, &ref) == SWA_OK);
  if (!g || !ref) return;
  swa_db_info_t gi, ri;
  int ns = 0;
  EXPECT(swa_group_info(g, &gi, &ns) == SWA_OK && swa_group_info(ref, &ri, nullptr) == SWA_OK);
  EXPECT(gi.seqcount == ri.seqcount && gi.symcount == ri.symcount && gi.longest == ri.longest && gi.first_seqno == 100 && ns >= 1 && ns <= nshards);
  int64_t M[1024] = {3};
  EXPECT(swa_group_set_scoring(g, M, 12, 1) == SWA_OK && swa_group_set_scoring(ref, M, 12, 1) == SWA_OK);
  EXPECT(swa_group_set_option(g, "no_such_option", "1") == SWA_EINVAL);
  {
    int64_t a = -1, b = -1;
    int32_t r = -1, t = -1;
    EXPECT(swa_group_wait(g) == SWA_OK && swa_group_load_progress(g, &a, &b, &r, &t) == SWA_OK && a == 0 && b == 0 && r == 0 && t == 0);
    EXPECT(swa_group_wait(nullptr) == SWA_EINVAL);
  }
  const int frames = gencode ? 6 : 1;
  for (int round = 0; round < 6; ++round) {
    std::vector<uint8_t> q(size_t(5 + rng() % 40)), q2(size_t(5 + rng() % 40));
    for (auto& c : q) c = uint8_t(rng() % 25);
    for (auto& c : q2) c = uint8_t(rng() % 25);
    const int64_t keep = round == 0 ? 1 : round == 1 ? 0 : int64_t(1 + rng() % 300), lo = int64_t(rng() % 39), hi = round == 2 ? 30 : 1000;
    if (round == 3 && nseq > 0) {          // an inclusion set that empties the first half: shards contributing no hit
      std::vector<uint8_t> inc(size_t(nseq), 1);
      for (int s = 0; s < nseq / 2; ++s) inc[size_t(s)] = 0;
      EXPECT(swa_group_set_inclusion(g, inc.data(), nseq) == SWA_OK && swa_group_set_inclusion(ref, inc.data(), nseq) == SWA_OK);
      EXPECT(swa_group_set_inclusion(g, inc.data(), nseq + 1) == SWA_EINVAL);
    }
    // all scores
    std::vector<int64_t> sa(size_t(nseq * frames) + 1, -7), sb(sa);
    swa_counters_t ca, cb;
    EXPECT(swa_group_search(g, q.data(), int64_t(q.size()), sa.data(), &ca) == SWA_OK);
    EXPECT(swa_group_search(ref, q.data(), int64_t(q.size()), sb.data(), &cb) == SWA_OK);
    EXPECT(sa == sb && ca.cells == cb.cells && ca.narrow == cb.narrow);
    // top-K
    std::vector<swa_hit_t> ha(size_t(keep) + 1), hb(size_t(keep) + 1);
    int64_t na = -1, nb = -1, ta = 0, tb = 0, oa = 0, ob = 0;
    EXPECT(swa_group_search_topk(g, q.data(), int64_t(q.size()), keep, lo, hi, ha.data(), &na, &ta, &oa, &ca) == SWA_OK);
    EXPECT(swa_group_search_topk(ref, q.data(), int64_t(q.size()), keep, lo, hi, hb.data(), &nb, &tb, &ob, &cb) == SWA_OK);
    EXPECT(na == nb && ta == tb && oa == ob && !std::memcmp(ha.data(), hb.data(), size_t(na) * sizeof(swa_hit_t)));
    // pair
    std::vector<swa_hit_t> h2a(size_t(keep) + 1), h2b(size_t(keep) + 1);
    int64_t n2a = 0, n2b = 0, t2a = 0, t2b = 0, o2a = 0, o2b = 0;
    EXPECT(swa_group_search_pair_topk(g, q.data(), int64_t(q.size()), q2.data(), int64_t(q2.size()), keep, lo, hi, keep, lo + 1, hi, ha.data(), &na, &ta, &oa, h2a.data(), &n2a, &t2a, &o2a, nullptr) == SWA_OK);
    EXPECT(swa_group_search_pair_topk(ref, q.data(), int64_t(q.size()), q2.data(), int64_t(q2.size()), keep, lo, hi, keep, lo + 1, hi, hb.data(), &nb, &tb, &ob, h2b.data(), &n2b, &t2b, &o2b, nullptr) == SWA_OK);
    EXPECT(na == nb && n2a == n2b && t2a == t2b && o2a == o2b && !std::memcmp(ha.data(), hb.data(), size_t(na) * sizeof(swa_hit_t)) &&
           !std::memcmp(h2a.data(), h2b.data(), size_t(n2a) * sizeof(swa_hit_t)));
    // frames
    const uint8_t* qs[3] = {q.data(), q2.data(), q.data()};
    const int64_t ql[3] = {int64_t(q.size()), int64_t(q2.size()), int64_t(q.size()) - 1};
    const int32_t tags[3] = {0, 1, 5};
    std::vector<swa_fhit_t> fa(size_t(keep) + 1), fb(size_t(keep) + 1);
    EXPECT(swa_group_search_frames_topk(g, 3, qs, ql, tags, keep, lo, hi, fa.data(), &na, &ta, &oa, nullptr) == SWA_OK);
    EXPECT(swa_group_search_frames_topk(ref, 3, qs, ql, tags, keep, lo, hi, fb.data(), &nb, &tb, &ob, nullptr) == SWA_OK);
    EXPECT(na == nb && ta == tb && oa == ob);
    for (int64_t i = 0; i < na && i < nb; ++i)
      EXPECT(fa[size_t(i)].seqno == fb[size_t(i)].seqno && fa[size_t(i)].score == fb[size_t(i)].score && fa[size_t(i)].qframe == fb[size_t(i)].qframe &&
             fa[size_t(i)].qstrand == fb[size_t(i)].qstrand && fa[size_t(i)].dframe == fb[size_t(i)].dframe && fa[size_t(i)].dstrand == fb[size_t(i)].dstrand);
    // alignment phase routed to the owning shards, scripts back in hit order (one buffer growth on the way)
    if (na > 0) {
      std::vector<int64_t> ids;
      std::vector<int32_t> ds, df;
      for (int64_t i = 0; i < na; ++i) { ids.push_back(fa[size_t(i)].seqno); ds.push_back(fa[size_t(i)].dstrand); df.push_back(fa[size_t(i)].dframe); }
      std::vector<swa_alignment_t> aa(static_cast<size_t>(na)), ab(static_cast<size_t>(na));
      std::vector<char> ta2(16), tb2(16);
      int64_t ua = 0, ub = 0;
      int rc = swa_group_align_hits(g, q.data(), int64_t(q.size()), ids.data(), ds.data(), df.data(), na, aa.data(), ta2.data(), 16, &ua);
      EXPECT(rc == SWA_ERANGE || rc == SWA_OK);
      ta2.resize(size_t(ua));
      EXPECT(swa_group_align_hits(g, q.data(), int64_t(q.size()), ids.data(), ds.data(), df.data(), na, aa.data(), ta2.data(), ua, &ua) == SWA_OK);
      rc = swa_group_align_hits(ref, q.data(), int64_t(q.size()), ids.data(), ds.data(), df.data(), na, ab.data(), tb2.data(), 16, &ub);
      tb2.resize(size_t(ub));
      EXPECT(swa_group_align_hits(ref, q.data(), int64_t(q.size()), ids.data(), ds.data(), df.data(), na, ab.data(), tb2.data(), ub, &ub) == SWA_OK);
      EXPECT(ua == ub && ta2 == tb2 && !std::memcmp(aa.data(), ab.data(), size_t(na) * sizeof(swa_alignment_t)));
      uint8_t buf[400];
      int64_t len = 0;
      EXPECT(swa_group_db_sequence(g, ids[0], 0, 0, buf, 400, &len, nullptr) == SWA_OK && (len == 0 || buf[0] == ids[0] % 251));
      EXPECT(swa_group_db_sequence(g, 100 + nseq, 0, 0, buf, 400, &len, nullptr) == SWA_EINVAL);
      EXPECT(swa_group_db_sequence(g, 99, 0, 0, buf, 400, &len, nullptr) == SWA_EINVAL);
    }
  }
  // a failing shard: the error surfaces on the caller's thread with the shard named, the group stays usable
  if (ns > 1) {
    EXPECT(swa_group_set_option(g, "fail_search", "1") == SWA_OK);
    uint8_t q[4] = {1, 2, 3, 4};
    swa_hit_t h[4];
    int64_t n = 0;
    EXPECT(swa_group_search_topk(g, q, 4, 4, 0, 100, h, &n, nullptr, nullptr, nullptr) == SWA_ENOMEM);
    EXPECT(std::strstr(swa_last_error(), "shard 1") != nullptr);
    EXPECT(swa_group_set_option(g, "fail_search", "0") == SWA_OK);
    EXPECT(swa_group_search_topk(g, q, 4, 4, 0, 100, h, &n, nullptr, nullptr, nullptr) == SWA_OK);
    // an exception on a shard's thread becomes a status on the caller's
    EXPECT(swa_group_set_option(g, "throw_search", "1") == SWA_OK);
    std::vector<int64_t> all(size_t(nseq * frames) + 1);
    EXPECT(swa_group_search(g, q, 4, all.data(), nullptr) == SWA_ENOMEM);
    EXPECT(std::strstr(swa_last_error(), "exception on a shard's thread") != nullptr);
    EXPECT(swa_group_set_option(g, "throw_search", "0") == SWA_OK);
    EXPECT(swa_group_search(g, q, 4, all.data(), nullptr) == SWA_OK);
  }
  swa_group_close(g);
  swa_group_close(ref);
}

int main()
{
  // shard bounds: cuts are monotone, cover everything, and balance residues
  {
    std::vector<int64_t> off = {0, 10, 10, 10, 500, 510, 520, 1000};
    int64_t cuts[5];
    EXPECT(swa_shard_bounds(off.data(), 7, 4, cuts) == SWA_OK && cuts[0] == 0 && cuts[4] == 7);
    for (int r = 0; r < 4; ++r) EXPECT(cuts[r] <= cuts[r + 1]);
  }
  const int sizes[] = {0, 1, 7, 600, 2500};
  const int shards[] = {1, 2, 3, 4, 8, 13};
  std::vector<std::thread> callers;
  unsigned seed = 1;
  for (int n : sizes)
    for (int s : shards) {
      const unsigned sd = seed++;
      callers.emplace_back(run, n, s, (sd % 3 == 0) ? 1 : 0, sd);      // several groups alive and searching at once
      if (callers.size() == 6) { for (auto& t : callers) t.join(); callers.clear(); }
    }
  for (auto& t : callers) t.join();
  if (bad) { std::fprintf(stderr, "%d comparisons failed\n", bad.load()); return 1; }
  std::printf("group check ok\n");
  return 0;
}
