// swa_group: one database over several devices behind one handle (include/swipe_amd.h, "several devices").
//
// The reference's counterpart is the thread / MPI layer of swipe.cc: run_threads + worker (1599-1699) give chunks of
// sequence numbers to -a N pthreads that enter one hit list under hitsmutex; mpiswipe's master hands chunks to
// workers and re-enters their reported hits (1812-2160).  Here the unit is a DEVICE: shard i of the database is
// resident on devices[i] and owned by one host thread for the group's whole life; a search is posted to every
// worker, each reduces its shard to a top-K on its device (swa_search_*topk), and the caller's thread merges the N
// short lists with the reference comparator.  Nothing in this file touches HIP: it is host orchestration over the
// single-shard C ABI, which is what keeps it testable under ThreadSanitizer with stand-in shards (tests/).
#include "../../include/swipe_amd.h"
#include "host_util.h"

#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using swa::fail;

namespace {
bool hit_before(const swa_hit_t& a, const swa_hit_t& b)   // hits.cc:188-190: score desc, then seqno desc
{
  return a.score > b.score || (a.score == b.score && a.seqno > b.seqno);
}

// One host thread bound to one shard: jobs run in posting order, the poster waits for the result.
struct Worker {
  std::thread th;
  std::mutex m;
  std::condition_variable cv;
  std::function<int()> job;
  bool has_job = false, done = false, stop = false;
  int rc = SWA_OK;
  std::string err;

  Worker() { th = std::thread([this] { loop(); }); }
  ~Worker()
  {
    { std::lock_guard<std::mutex> l(m); stop = true; }
    cv.notify_all();
    if (th.joinable()) th.join();
  }
  void loop()
  {
    std::unique_lock<std::mutex> l(m);
    for (;;) {
      cv.wait(l, [this] { return has_job || stop; });
      if (!has_job) return;                                // stop, nothing pending
      std::function<int()> f = std::move(job);
      has_job = false;
      l.unlock();
      int r;
      try { r = f(); }                                     // the C ABI does not throw; a stand-in shard of the tests may
      catch (const std::exception& x) { r = fail(SWA_ENOMEM, std::string("exception on a shard's thread: ") + x.what()); }
      catch (...) { r = fail(SWA_ENOMEM, "exception on a shard's thread"); }   // anything else would end the process (std::terminate)
      std::string e = r == SWA_OK ? std::string() : std::string(swa_last_error());   // thread-local on THIS thread
      l.lock();
      rc = r;
      err = std::move(e);
      done = true;
      cv.notify_all();
    }
  }
  void post(std::function<int()> f)
  {
    std::lock_guard<std::mutex> l(m);
    job = std::move(f);
    has_job = true;
    done = false;
    cv.notify_all();
  }
  int wait(std::string* e)
  {
    std::unique_lock<std::mutex> l(m);
    cv.wait(l, [this] { return done; });
    done = false;
    if (rc != SWA_OK && e) *e = err;
    return rc;
  }
};
}  // namespace

struct swa_group {
  std::vector<swa_db*> shard;
  std::vector<int> device;
  std::vector<int64_t> first, count;                      // global number of a shard's first sequence, its sequences
  std::vector<std::unique_ptr<Worker>> worker;
  std::mutex call;                                        // the contract is one caller at a time; two are serialised, not mixed up
  int frames = 1;

  // fn(i) on every shard's own thread, all at once; the first failure (by shard number) is reported
  int run_all(const std::function<int(int)>& fn)
  {
    std::lock_guard<std::mutex> one_caller(call);
    const int n = int(worker.size());
    for (int i = 0; i < n; ++i) worker[size_t(i)]->post([&fn, i] { return fn(i); });
    int rc = SWA_OK;
    std::string msg;
    for (int i = 0; i < n; ++i) {
      std::string e;
      const int r = worker[size_t(i)]->wait(&e);
      if (r != SWA_OK && rc == SWA_OK) { rc = r; msg = "shard " + std::to_string(i) + ": " + e; }
    }
    return rc == SWA_OK ? SWA_OK : fail(rc, msg);
  }
  int run_one(int i, const std::function<int()>& fn)
  {
    std::lock_guard<std::mutex> one_caller(call);
    worker[size_t(i)]->post(fn);
    std::string e;
    const int r = worker[size_t(i)]->wait(&e);
    return r == SWA_OK ? SWA_OK : fail(r, "shard " + std::to_string(i) + ": " + e);
  }
  int owner(int64_t seqno) const
  {
    const size_t k = size_t(std::upper_bound(first.begin(), first.end(), seqno) - first.begin());
    if (k == 0) return -1;
    return seqno < first[k - 1] + count[k - 1] ? int(k - 1) : -1;
  }
};

namespace {
void sum_counters(const std::vector<swa_counters_t>& c, swa_counters_t* out)
{
  if (!out) return;
  *out = swa_counters_t{};
  for (size_t i = 0; i < c.size(); ++i) {
    out->narrow += c[i].narrow;
    out->wide += c[i].wide;
    out->full += c[i].full;
    out->cells += c[i].cells;
    out->kernel_ms = std::max(out->kernel_ms, c[i].kernel_ms);     // the shards run side by side
    out->total_ms = std::max(out->total_ms, c[i].total_ms);
    if (i == 0 || c[i].narrow_rows > out->narrow_rows) { out->narrow_rows = c[i].narrow_rows; out->narrow_shifted = c[i].narrow_shifted; }
    out->loading_parts += c[i].loading_parts;
    out->requeue_form = std::max(out->requeue_form, c[i].requeue_form);
  }
}

int check_devices(int nshards, const int* devices)
{
  if (nshards < 1 || nshards > 1024 || !devices) return fail(SWA_EINVAL, "between 1 and 1024 shards and their devices expected");
  const int ndev = swa_device_count();
  if (ndev < 1) return fail(SWA_ENODEV, "no HIP device (swipe_amd has no CPU fallback)");
  for (int i = 0; i < nshards; ++i)
    if (devices[i] < 0 || devices[i] >= ndev) return fail(SWA_ENODEV, "no such HIP device: " + std::to_string(devices[i]));
  return SWA_OK;
}

// the non-empty shards of `cuts` become workers; open_one(shard index r of cuts, first, last-exclusive, device, &db)
int make_group(const std::vector<int64_t>& cuts, const int* devices, int64_t base_seqno, int frames,
               const std::function<int(int64_t, int64_t, int, swa_db**)>& open_one, swa_group** out)
{
  std::unique_ptr<swa_group> g(new swa_group);
  g->frames = frames;
  for (size_t r = 0; r + 1 < cuts.size(); ++r) {
    if (cuts[r + 1] <= cuts[r]) continue;
    g->first.push_back(base_seqno + cuts[r]);
    g->count.push_back(cuts[r + 1] - cuts[r]);
    g->device.push_back(devices[r]);
  }
  if (g->first.empty()) {                                  // an empty database still answers searches (with no hits)
    g->first.push_back(base_seqno);
    g->count.push_back(0);
    g->device.push_back(devices[0]);
  }
  const size_t n = g->first.size();
  g->shard.assign(n, nullptr);
  for (size_t i = 0; i < n; ++i) g->worker.emplace_back(new Worker);
  swa_group* gp = g.get();
  const int rc = gp->run_all([&](int i) {
    return open_one(gp->first[size_t(i)] - base_seqno, gp->first[size_t(i)] - base_seqno + gp->count[size_t(i)], gp->device[size_t(i)],
                    &gp->shard[size_t(i)]);
  });
  if (rc != SWA_OK) {
    const std::string keep = swa_last_error();
    swa_group_close(g.release());
    return fail(rc, keep);
  }
  *out = g.release();
  return SWA_OK;
}
}  // namespace

extern "C" int swa_shard_bounds(const int64_t* offsets, int64_t nseq, int nshards, int64_t* cuts)
try {
  if (!offsets || nseq < 0 || nshards < 1 || !cuts) return fail(SWA_EINVAL, "bad argument");
  const int64_t total = offsets[nseq] - offsets[0];
  cuts[0] = 0;
  for (int r = 1; r < nshards; ++r) {
    // the first sequence starting at or beyond r / nshards of the residues (128-bit product: 2^63 residues x 1024 shards)
    const int64_t target = offsets[0] + int64_t((__int128)total * r / nshards);
    int64_t c = int64_t(std::lower_bound(offsets, offsets + nseq + 1, target) - offsets);
    c = std::min(std::max(c, cuts[r - 1]), nseq);
    cuts[r] = c;
  }
  cuts[nshards] = nseq;
  return SWA_OK;
} SWA_CATCH

extern "C" int swa_blastdb_shard_bounds(const char* basename, int symtype, int nshards, int64_t* cuts)
try {
  if (nshards < 1 || !cuts) return fail(SWA_EINVAL, "bad argument");
  if (nshards == 1) {                                      // one shard: everything; no need to walk the index for the lengths
    cuts[0] = 0;
    return swa::read_blast_totals(basename, symtype, &cuts[1], nullptr);
  }
  std::vector<int64_t> off;
  const int rc = swa::read_blast_lengths(basename, symtype, off);
  if (rc != SWA_OK) return rc;
  return swa_shard_bounds(off.data(), int64_t(off.size()) - 1, nshards, cuts);
} SWA_CATCH

extern "C" int swa_group_open(const char* basename, int symtype, int db_gencode, int nshards, const int* devices, swa_group** out)
try {
  if (!out) return fail(SWA_EINVAL, "null output handle");
  *out = nullptr;
  if (!basename) return fail(SWA_EINVAL, "null database name");
  int rc = check_devices(nshards, devices);
  if (rc != SWA_OK) return rc;
  const int filetype = db_gencode ? SWA_SYMTYPE_NUCLEOTIDE : symtype;
  std::vector<int64_t> cuts(size_t(nshards) + 1);
  rc = swa_blastdb_shard_bounds(basename, filetype, nshards, cuts.data());
  if (rc != SWA_OK) return rc;
  const std::string base(basename);
  return make_group(cuts, devices, 0, db_gencode ? 6 : 1, [&](int64_t lo, int64_t hi, int dev, swa_db** db) {
    if (hi <= lo) {                                        // empty database: a handle with no sequences
      const int64_t zero = 0;
      return db_gencode ? swa_db_from_memory_translated(nullptr, &zero, 0, db_gencode, dev, 0, 0, 0, db)
                        : swa_db_from_memory(nullptr, &zero, 0, symtype, dev, 0, 0, 0, db);
    }
    // every shard streams into its device behind this call (swa_db_open_async): the group is usable at once, its first
    // search follows the loaders
    return db_gencode ? swa_db_open_translated(base.c_str(), db_gencode, dev, lo, hi - 1, db)
                      : swa_db_open_async(base.c_str(), symtype, dev, lo, hi - 1, db);
  }, out);
} SWA_CATCH

// shards that may not be resident: every shard is opened with its own HBM budget (swa_db_open_streamed) and walks its parts
// through two device slots per search; the group layer above is unchanged
extern "C" int swa_group_open_streamed(const char* basename, int symtype, int nshards, const int* devices, int64_t hbm_budget_bytes,
                                       swa_group** out)
try {
  if (!out) return fail(SWA_EINVAL, "null output handle");
  *out = nullptr;
  if (!basename) return fail(SWA_EINVAL, "null database name");
  int rc = check_devices(nshards, devices);
  if (rc != SWA_OK) return rc;
  std::vector<int64_t> cuts(size_t(nshards) + 1);
  rc = swa_blastdb_shard_bounds(basename, symtype, nshards, cuts.data());
  if (rc != SWA_OK) return rc;
  const std::string base(basename);
  return make_group(cuts, devices, 0, 1, [&](int64_t lo, int64_t hi, int dev, swa_db** db) {
    if (hi <= lo) {
      const int64_t zero = 0;
      return swa_db_from_memory(nullptr, &zero, 0, symtype, dev, 0, 0, 0, db);
    }
    return swa_db_open_streamed(base.c_str(), symtype, dev, lo, hi - 1, hbm_budget_bytes, db);
  }, out);
} SWA_CATCH

extern "C" int swa_group_from_memory(const uint8_t* residues, const int64_t* offsets, int64_t nseq, int symtype, int db_gencode,
                                     int nshards, const int* devices, int64_t first_seqno, int64_t total_seqcount,
                                     int64_t total_symcount, swa_group** out)
try {
  if (!out) return fail(SWA_EINVAL, "null output handle");
  *out = nullptr;
  if (nseq < 0 || !offsets) return fail(SWA_EINVAL, "bad database arrays");
  int rc = check_devices(nshards, devices);
  if (rc != SWA_OK) return rc;
  std::vector<int64_t> cuts(size_t(nshards) + 1);
  rc = swa_shard_bounds(offsets, nseq, nshards, cuts.data());
  if (rc != SWA_OK) return rc;
  const int64_t tseq = total_seqcount > 0 ? total_seqcount : nseq;
  const int64_t tsym = total_symcount > 0 ? total_symcount : offsets[nseq] - offsets[0];
  return make_group(cuts, devices, first_seqno, db_gencode ? 6 : 1, [&](int64_t lo, int64_t hi, int dev, swa_db** db) {
    return db_gencode ? swa_db_from_memory_translated(residues, offsets + lo, hi - lo, db_gencode, dev, first_seqno + lo, tseq, tsym, db)
                      : swa_db_from_memory(residues, offsets + lo, hi - lo, symtype, dev, first_seqno + lo, tseq, tsym, db);
  }, out);
} SWA_CATCH

extern "C" void swa_group_close(swa_group* g)
{
  if (!g) return;
  // every shard is closed by the thread that served it, then the threads end
  for (size_t i = 0; i < g->worker.size(); ++i) {
    swa_db* db = g->shard[i];
    g->worker[i]->post([db] { swa_db_close(db); return SWA_OK; });
  }
  for (auto& w : g->worker) w->wait(nullptr);
  delete g;
}

extern "C" int swa_group_info(const swa_group* g, swa_db_info_t* info, int* nshards)
try {
  if (!g || !info) return fail(SWA_EINVAL, "null argument");
  *info = swa_db_info_t{};
  for (size_t i = 0; i < g->shard.size(); ++i) {
    swa_db_info_t s;
    const int rc = swa_db_info(g->shard[i], &s);
    if (rc != SWA_OK) return rc;
    info->seqcount += s.seqcount;
    info->symcount += s.symcount;
    info->longest = std::max(info->longest, s.longest);
    info->hbm_bytes += s.hbm_bytes;
    if (i == 0) { info->first_seqno = s.first_seqno; info->total_seqcount = s.total_seqcount; info->total_symcount = s.total_symcount; info->frames = s.frames; }
  }
  if (nshards) *nshards = int(g->shard.size());
  return SWA_OK;
} SWA_CATCH

extern "C" int swa_group_shard(const swa_group* g, int i, swa_db** db)
try {
  if (!g || !db || i < 0 || i >= int(g->shard.size())) return fail(SWA_EINVAL, "no such shard");
  *db = g->shard[size_t(i)];
  return SWA_OK;
} SWA_CATCH

extern "C" int swa_group_set_scoring(swa_group* g, const int64_t* matrix, int64_t gapopenextend, int64_t gapextend)
try {
  if (!g || !matrix) return fail(SWA_EINVAL, "null argument");
  return g->run_all([&](int i) { return swa_set_scoring(g->shard[size_t(i)], matrix, gapopenextend, gapextend); });
} SWA_CATCH

// Shards of swa_group_open stream in behind the call (swa_db_open_async): wait for all of them, first error wins.
extern "C" int swa_group_wait(swa_group* g)
try {
  if (!g) return fail(SWA_EINVAL, "null group handle");
  return g->run_all([&](int i) { return swa_db_wait(g->shard[size_t(i)]); });
} SWA_CATCH

// sums over the shards (all zero once every shard is resident); a failed load is reported here as by swa_db_load_progress
extern "C" int swa_group_load_progress(swa_group* g, int64_t* bytes_loaded, int64_t* bytes_total, int32_t* parts_ready, int32_t* parts_total)
try {
  if (!g) return fail(SWA_EINVAL, "null group handle");
  int64_t done = 0, total = 0;
  int32_t ready = 0, parts = 0;
  for (swa_db* db : g->shard) {
    int64_t d = 0, t = 0;
    int32_t r = 0, p = 0;
    const int rc = swa_db_load_progress(db, &d, &t, &r, &p);
    if (rc != SWA_OK) return rc;
    done += d; total += t; ready += r; parts += p;
  }
  if (bytes_loaded) *bytes_loaded = done;
  if (bytes_total) *bytes_total = total;
  if (parts_ready) *parts_ready = ready;
  if (parts_total) *parts_total = parts;
  return SWA_OK;
} SWA_CATCH

extern "C" int swa_group_set_option(swa_group* g, const char* key, const char* value)
try {
  if (!g || !key) return fail(SWA_EINVAL, "null argument");
  return g->run_all([&](int i) { return swa_set_option(g->shard[size_t(i)], key, value); });
} SWA_CATCH

extern "C" int swa_group_set_inclusion(swa_group* g, const uint8_t* include, int64_t n)
try {
  if (!g) return fail(SWA_EINVAL, "null group handle");
  int64_t total = 0;
  for (int64_t c : g->count) total += c;
  if (include && n != total) return fail(SWA_EINVAL, "inclusion array must have one entry per sequence of the group");
  return g->run_all([&](int i) {
    const int64_t lo = g->first[size_t(i)] - g->first[0];
    return swa_db_set_inclusion(g->shard[size_t(i)], include ? include + lo : nullptr, g->count[size_t(i)]);
  });
} SWA_CATCH

extern "C" int swa_group_search(swa_group* g, const uint8_t* query, int64_t qlen, int64_t* scores, swa_counters_t* counters)
try {
  if (!g) return fail(SWA_EINVAL, "null group handle");
  std::vector<swa_counters_t> c(g->shard.size());
  const int rc = g->run_all([&](int i) {
    const int64_t lo = (g->first[size_t(i)] - g->first[0]) * g->frames;
    return swa_search(g->shard[size_t(i)], query, qlen, scores ? scores + lo : nullptr, &c[size_t(i)]);
  });
  if (rc == SWA_OK) sum_counters(c, counters);
  return rc;
} SWA_CATCH

extern "C" int swa_group_search_topk(swa_group* g, const uint8_t* query, int64_t qlen, int64_t keep, int64_t minscore,
                                     int64_t maxscore, swa_hit_t* hits, int64_t* nhits, int64_t* totalhits, int64_t* obvious,
                                     swa_counters_t* counters)
try {
  if (!g) return fail(SWA_EINVAL, "null group handle");
  if (keep < 0 || (keep > 0 && !hits) || !nhits) return fail(SWA_EINVAL, "bad hit buffer");
  *nhits = 0;
  const size_t n = g->shard.size(), stride = size_t(std::max<int64_t>(keep, 1));
  std::vector<swa_hit_t> lists(n * stride);
  std::vector<int64_t> cnt(n, 0), tot(n, 0), obv(n, 0);
  std::vector<swa_counters_t> c(n);
  int rc = g->run_all([&](int i) {
    return swa_search_topk(g->shard[size_t(i)], query, qlen, keep, minscore, maxscore, lists.data() + size_t(i) * stride, &cnt[size_t(i)],
                           &tot[size_t(i)], &obv[size_t(i)], &c[size_t(i)]);
  });
  if (rc != SWA_OK) return rc;
  rc = swa_hits_merge(lists.data(), cnt.data(), int(n), int64_t(stride), keep, hits, nhits);
  if (rc != SWA_OK) return rc;
  int64_t t = 0, o = 0;
  for (size_t i = 0; i < n; ++i) { t += tot[i]; o += obv[i]; }
  if (totalhits) *totalhits = t;
  if (obvious) *obvious = o;
  sum_counters(c, counters);
  return SWA_OK;
} SWA_CATCH

extern "C" int swa_group_search_pair_topk(swa_group* g, const uint8_t* query1, int64_t qlen1, const uint8_t* query2, int64_t qlen2,
                                          int64_t keep1, int64_t minscore1, int64_t maxscore1, int64_t keep2, int64_t minscore2,
                                          int64_t maxscore2, swa_hit_t* hits1, int64_t* nhits1, int64_t* totalhits1,
                                          int64_t* obvious1, swa_hit_t* hits2, int64_t* nhits2, int64_t* totalhits2,
                                          int64_t* obvious2, swa_counters_t* counters)
try {
  if (!g) return fail(SWA_EINVAL, "null group handle");
  if (keep1 < 0 || keep2 < 0 || (keep1 > 0 && !hits1) || (keep2 > 0 && !hits2) || !nhits1 || !nhits2)
    return fail(SWA_EINVAL, "bad hit buffer");
  *nhits1 = *nhits2 = 0;
  const size_t n = g->shard.size(), s1 = size_t(std::max<int64_t>(keep1, 1)), s2 = size_t(std::max<int64_t>(keep2, 1));
  std::vector<swa_hit_t> l1(n * s1), l2(n * s2);
  std::vector<int64_t> c1(n, 0), c2(n, 0), t1(n, 0), t2(n, 0), o1(n, 0), o2(n, 0);
  std::vector<swa_counters_t> c(n);
  int rc = g->run_all([&](int i) {
    const size_t k = size_t(i);
    return swa_search_pair_topk(g->shard[k], query1, qlen1, query2, qlen2, keep1, minscore1, maxscore1, keep2, minscore2, maxscore2,
                                l1.data() + k * s1, &c1[k], &t1[k], &o1[k], l2.data() + k * s2, &c2[k], &t2[k], &o2[k], &c[k]);
  });
  if (rc != SWA_OK) return rc;
  rc = swa_hits_merge(l1.data(), c1.data(), int(n), int64_t(s1), keep1, hits1, nhits1);
  if (rc == SWA_OK) rc = swa_hits_merge(l2.data(), c2.data(), int(n), int64_t(s2), keep2, hits2, nhits2);
  if (rc != SWA_OK) return rc;
  int64_t a = 0, b = 0, x = 0, y = 0;
  for (size_t i = 0; i < n; ++i) { a += t1[i]; b += o1[i]; x += t2[i]; y += o2[i]; }
  if (totalhits1) *totalhits1 = a;
  if (obvious1) *obvious1 = b;
  if (totalhits2) *totalhits2 = x;
  if (obvious2) *obvious2 = y;
  sum_counters(c, counters);
  return SWA_OK;
} SWA_CATCH

extern "C" int swa_group_search_frames_topk(swa_group* g, int nq, const uint8_t* const* queries, const int64_t* qlens,
                                            const int32_t* qtags, int64_t keep, int64_t minscore, int64_t maxscore,
                                            swa_fhit_t* hits, int64_t* nhits, int64_t* totalhits, int64_t* obvious,
                                            swa_counters_t* counters)
try {
  if (!g) return fail(SWA_EINVAL, "null group handle");
  if (keep < 0 || (keep > 0 && !hits) || !nhits) return fail(SWA_EINVAL, "bad hit buffer");
  *nhits = 0;
  const size_t n = g->shard.size(), stride = size_t(std::max<int64_t>(keep, 1));
  std::vector<swa_fhit_t> lists(n * stride);
  std::vector<int64_t> cnt(n, 0), tot(n, 0), obv(n, 0);
  std::vector<swa_counters_t> c(n);
  int rc = g->run_all([&](int i) {
    const size_t k = size_t(i);
    return swa_search_frames_topk(g->shard[k], nq, queries, qlens, qtags, keep, minscore, maxscore, lists.data() + k * stride, &cnt[k],
                                  &tot[k], &obv[k], &c[k]);
  });
  if (rc != SWA_OK) return rc;
  rc = swa_fhits_merge(lists.data(), cnt.data(), int(n), int64_t(stride), keep, hits, nhits);
  if (rc != SWA_OK) return rc;
  int64_t t = 0, o = 0;
  for (size_t i = 0; i < n; ++i) { t += tot[i]; o += obv[i]; }
  if (totalhits) *totalhits = t;
  if (obvious) *obvious = o;
  sum_counters(c, counters);
  return SWA_OK;
} SWA_CATCH

extern "C" int swa_group_align_hits(swa_group* g, const uint8_t* query, int64_t qlen, const int64_t* seqnos,
                                    const int32_t* dstrands, const int32_t* dframes, int64_t n, swa_alignment_t* out, char* text,
                                    int64_t text_cap, int64_t* text_used)
try {
  if (!g) return fail(SWA_EINVAL, "null group handle");
  if (n < 0 || text_cap < 0 || !text_used || (n > 0 && (!seqnos || !out)) || (text_cap > 0 && !text))
    return fail(SWA_EINVAL, "bad argument");
  *text_used = 0;
  const size_t ns = g->shard.size();
  // hits by owning shard, in hit order
  std::vector<std::vector<int64_t>> idx(ns);
  for (int64_t i = 0; i < n; ++i) {
    const int o = g->owner(seqnos[i]);
    if (o < 0) return fail(SWA_EINVAL, "sequence number outside the group's shards");
    idx[size_t(o)].push_back(i);
  }
  struct Part { std::vector<int64_t> seq; std::vector<int32_t> ds, df; std::vector<swa_alignment_t> al; std::vector<char> text; };
  std::vector<Part> part(ns);
  for (size_t s = 0; s < ns; ++s)
    for (int64_t i : idx[s]) {
      part[s].seq.push_back(seqnos[i]);
      part[s].ds.push_back(dstrands ? dstrands[i] : 0);
      part[s].df.push_back(dframes ? dframes[i] : 0);
    }
  const int rc = g->run_all([&](int si) {
    Part& p = part[size_t(si)];
    const int64_t m = int64_t(p.seq.size());
    if (m == 0) return int(SWA_OK);
    p.al.resize(size_t(m));
    p.text.resize(1 << 16);
    int64_t used = 0;
    int r = swa_align_hits(g->shard[size_t(si)], query, qlen, p.seq.data(), p.ds.data(), p.df.data(), m, p.al.data(), p.text.data(),
                           int64_t(p.text.size()), &used);
    if (r == SWA_ERANGE) {
      p.text.resize(size_t(used));
      r = swa_align_hits(g->shard[size_t(si)], query, qlen, p.seq.data(), p.ds.data(), p.df.data(), m, p.al.data(), p.text.data(),
                         int64_t(p.text.size()), &used);
    }
    p.text.resize(r == SWA_OK ? size_t(used) : 0);
    return r;
  });
  if (rc != SWA_OK) return rc;
  // back into hit order; the edit scripts are laid out in that order too
  std::vector<const swa_alignment_t*> src(static_cast<size_t>(n), nullptr);
  std::vector<const char*> script(static_cast<size_t>(n), nullptr);
  for (size_t s = 0; s < ns; ++s)
    for (size_t k = 0; k < idx[s].size(); ++k) {
      src[size_t(idx[s][k])] = &part[s].al[k];
      script[size_t(idx[s][k])] = part[s].text.data() + part[s].al[k].cigar_offset;
    }
  int64_t need = 0;
  for (int64_t i = 0; i < n; ++i) need += src[size_t(i)]->cigar_len + 1;
  int64_t at = 0;
  for (int64_t i = 0; i < n; ++i) {
    out[i] = *src[size_t(i)];
    out[i].cigar_offset = at;
    if (need <= text_cap) {
      std::memcpy(text + at, script[size_t(i)], size_t(out[i].cigar_len));
      text[at + out[i].cigar_len] = '\0';
    }
    at += out[i].cigar_len + 1;
  }
  *text_used = need;
  if (need > text_cap) return fail(SWA_ERANGE, "text buffer too small for the edit scripts");
  return SWA_OK;
} SWA_CATCH

extern "C" int swa_group_db_sequence(swa_group* g, int64_t seqno, int dstrand, int dframe, uint8_t* buf, int64_t cap,
                                     int64_t* len, int64_t* ntlen)
try {
  if (!g) return fail(SWA_EINVAL, "null group handle");
  const int o = g->owner(seqno);
  if (o < 0) return fail(SWA_EINVAL, "sequence number outside the group's shards");
  return g->run_one(o, [&] { return swa_db_sequence(g->shard[size_t(o)], seqno, dstrand, dframe, buf, cap, len, ntlen); });
} SWA_CATCH

// ---- merging per-shard lists (tag_search_report re-entered through hits_enter, swipe.cc:1951-1974) -----------------
// the same for frame-tagged hits: entries of one sequence all come from the shard that holds it, already in the
// reference's order (query frame, then database frame), which a stable sort on (score, seqno) preserves
extern "C" int swa_fhits_merge(const swa_fhit_t* lists, const int64_t* counts, int nlists, int64_t stride, int64_t keep,
                               swa_fhit_t* out, int64_t* nout)
try {
  if (!lists || !counts || nlists < 0 || keep < 0 || stride < 0 || !nout || (keep > 0 && !out)) return fail(SWA_EINVAL, "bad argument");
  std::vector<swa_fhit_t> all;
  for (int l = 0; l < nlists; ++l)
    if (counts[l] < 0 || counts[l] > stride) return fail(SWA_EINVAL, "a list longer than its stride");
  for (int l = 0; l < nlists; ++l)
    for (int64_t i = 0; i < counts[l]; ++i) all.push_back(lists[int64_t(l) * stride + i]);
  std::stable_sort(all.begin(), all.end(), [](const swa_fhit_t& a, const swa_fhit_t& b) {
    if (a.score != b.score) return a.score > b.score;
    return a.seqno > b.seqno;
  });
  const size_t k = std::min<size_t>(size_t(keep), all.size());
  for (size_t i = 0; i < k; ++i) out[i] = all[i];
  *nout = int64_t(k);
  return SWA_OK;
} SWA_CATCH

extern "C" int swa_hits_merge(const swa_hit_t* lists, const int64_t* counts, int nlists, int64_t stride,
                              int64_t keep, swa_hit_t* out, int64_t* nout)
try {
  if (!lists || !counts || nlists < 0 || keep < 0 || stride < 0 || !nout || (keep > 0 && !out)) return fail(SWA_EINVAL, "bad argument");
  std::vector<swa_hit_t> all;
  for (int l = 0; l < nlists; ++l)
    if (counts[l] < 0 || counts[l] > stride) return fail(SWA_EINVAL, "a list longer than its stride");
  for (int l = 0; l < nlists; ++l)
    for (int64_t i = 0; i < counts[l]; ++i) all.push_back(lists[int64_t(l) * stride + i]);
  std::stable_sort(all.begin(), all.end(), hit_before);
  const size_t k = std::min<size_t>(size_t(keep), all.size());
  for (size_t i = 0; i < k; ++i) out[i] = all[i];
  *nout = int64_t(k);
  return SWA_OK;
} SWA_CATCH
