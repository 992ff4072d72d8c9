// TEST SCAFFOLDING, not product code: a stand-in for the single-shard C ABI so that swipe_amd/csrc/group.cpp (threads,
// job hand-over, routing, merging) can run under ThreadSanitizer on a machine without a GPU.  A "shard" here scores
// sequence s against a query with a hash of (s, query, scoring) - no alignment is computed anywhere - so a group of N
// stand-in shards must return exactly what one stand-in shard holding every sequence returns.
#include "../../include/swipe_amd.h"
#include "../../swipe_amd/csrc/host_util.h"

#include <algorithm>
#include <chrono>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace swa {
static thread_local std::string g_err;
int fail(int code, const std::string& msg) { g_err = msg; return code; }
int read_blast_lengths(const char*, int, std::vector<int64_t>&) { return fail(SWA_EIO, "stand-in: no files"); }
int read_blast_totals(const char*, int, int64_t*, int64_t*) { return fail(SWA_EIO, "stand-in: no files"); }
}  // namespace swa

struct swa_db {
  int64_t first = 0, nseq = 0, tseq = 0, tsym = 0;
  std::vector<int64_t> off;
  int frames = 1;
  uint64_t scoring = 0;
  std::vector<uint8_t> include;
  bool fail_search = false, throw_search = false;
};

static uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; return x ^ (x >> 33); }
static uint64_t qhash(const uint8_t* q, int64_t n) { uint64_t h = 1469598103934665603ULL; for (int64_t i = 0; i < n; ++i) h = (h ^ q[i]) * 1099511628211ULL; return h ^ uint64_t(n); }
// few distinct values: ties across shard boundaries are the interesting case
static int64_t fake_score(const swa_db* db, int64_t seqno, int frame, uint64_t qh) { return int64_t(mix(uint64_t(seqno) * 6 + uint64_t(frame) + qh + db->scoring) % 40); }

extern "C" {
const char* swa_last_error(void) { return swa::g_err.c_str(); }
int swa_device_count(void) { return 1; }
int swa_db_from_memory(const uint8_t*, const int64_t* offsets, int64_t nseq, int, int, int64_t first_seqno, int64_t tseq, int64_t tsym, swa_db** out)
{
  swa_db* d = new swa_db;
  d->first = first_seqno; d->nseq = nseq; d->tseq = tseq; d->tsym = tsym;
  d->off.assign(offsets, offsets + nseq + 1);
  std::this_thread::sleep_for(std::chrono::microseconds(200 * (first_seqno % 5)));
  *out = d;
  return SWA_OK;
}
int swa_db_from_memory_translated(const uint8_t* r, const int64_t* o, int64_t n, int, int dev, int64_t f, int64_t ts, int64_t ty, swa_db** out)
{
  const int rc = swa_db_from_memory(r, o, n, 0, dev, f, ts, ty, out);
  if (rc == SWA_OK) (*out)->frames = 6;
  return rc;
}
int swa_db_open(const char*, int, int, int64_t, int64_t, swa_db**) { return swa::fail(SWA_EIO, "stand-in: no files"); }
int swa_db_open_async(const char*, int, int, int64_t, int64_t, swa_db**) { return swa::fail(SWA_EIO, "stand-in: no files"); }
int swa_db_wait(swa_db* d) { return d ? SWA_OK : swa::fail(SWA_EINVAL, "null database handle"); }          // a stand-in shard is resident
int swa_db_load_progress(swa_db* d, int64_t* a, int64_t* b, int32_t* r, int32_t* t)
{
  if (!d) return swa::fail(SWA_EINVAL, "null database handle");
  if (a) *a = 0;
  if (b) *b = 0;
  if (r) *r = 0;
  if (t) *t = 0;
  return SWA_OK;
}
int swa_db_open_streamed(const char*, int, int, int64_t, int64_t, int64_t, swa_db**) { return swa::fail(SWA_EIO, "stand-in: no files"); }
int swa_db_open_translated(const char*, int, int, int64_t, int64_t, swa_db**) { return swa::fail(SWA_EIO, "stand-in: no files"); }
void swa_db_close(swa_db* d) { delete d; }
int swa_db_info(const swa_db* d, swa_db_info_t* i)
{
  *i = swa_db_info_t{};
  i->seqcount = d->nseq; i->symcount = d->off[size_t(d->nseq)] - d->off[0]; i->first_seqno = d->first; i->frames = d->frames;
  i->total_seqcount = d->tseq; i->total_symcount = d->tsym; i->hbm_bytes = 1;
  for (int64_t s = 0; s < d->nseq; ++s) i->longest = std::max(i->longest, d->off[size_t(s) + 1] - d->off[size_t(s)]);
  return SWA_OK;
}
int swa_set_scoring(swa_db* d, const int64_t* m, int64_t goe, int64_t ge) { d->scoring = uint64_t(m[0] * 31 + goe * 7 + ge); return SWA_OK; }
int swa_set_option(swa_db* d, const char* key, const char* value)
{
  if (!std::strcmp(key, "fail_search")) { d->fail_search = value && value[0] == '1' && d->first > 100; return SWA_OK; }   // every shard but the first
  if (!std::strcmp(key, "throw_search")) { d->throw_search = value && value[0] == '1' && d->first > 100; return SWA_OK; }
  return std::strcmp(key, "bound") ? swa::fail(SWA_EINVAL, std::string("unknown option ") + key) : SWA_OK;
}
int swa_db_set_inclusion(swa_db* d, const uint8_t* inc, int64_t n)
{
  if (inc && n != d->nseq) return swa::fail(SWA_EINVAL, "inclusion array must have one entry per sequence of the shard");
  if (inc) d->include.assign(inc, inc + n); else d->include.clear();
  return SWA_OK;
}
static bool included(const swa_db* d, int64_t local) { return d->include.empty() || d->include[size_t(local)]; }
int swa_search(swa_db* d, const uint8_t* q, int64_t qlen, int64_t* scores, swa_counters_t* c)
{
  if (d->fail_search) return swa::fail(SWA_ENOMEM, "stand-in failure");
  if (d->throw_search) throw std::bad_alloc();           // the real C ABI never throws (SWA_CATCH); the group survives one that does
  const uint64_t qh = qhash(q, qlen);
  for (int64_t s = 0; s < d->nseq && scores; ++s)
    for (int f = 0; f < d->frames; ++f) scores[s * d->frames + f] = included(d, s) ? fake_score(d, d->first + s, f, qh) : -1;
  if (c) { *c = swa_counters_t{}; c->narrow = d->nseq; c->cells = (d->off[size_t(d->nseq)] - d->off[0]) * qlen; c->total_ms = 1 + double(d->first % 3); c->kernel_ms = 1; }
  return SWA_OK;
}
struct C6 { int64_t seqno, score; int which, dtag; };
static bool c6_before(const C6& a, const C6& b)
{
  if (a.score != b.score) return a.score > b.score;
  if (a.seqno != b.seqno) return a.seqno > b.seqno;
  if (a.which != b.which) return a.which < b.which;
  return a.dtag < b.dtag;
}
static void collect(const swa_db* d, const uint8_t* q, int64_t qlen, int which, int64_t lo, int64_t hi, std::vector<C6>& out, int64_t* tot, int64_t* obv)
{
  const uint64_t qh = qhash(q, qlen);
  for (int64_t s = 0; s < d->nseq; ++s)
    for (int f = 0; f < d->frames && included(d, s); ++f) {
      const int64_t v = fake_score(d, d->first + s, f, qh);
      if (v >= lo) ++*tot;
      if (v > hi) ++*obv;
      if (v >= lo && v <= hi) out.push_back({d->first + s, v, which, f});
    }
}
int swa_search_topk(swa_db* d, const uint8_t* q, int64_t qlen, int64_t keep, int64_t lo, int64_t hi, swa_hit_t* hits, int64_t* n, int64_t* tot, int64_t* obv, swa_counters_t* c)
{
  if (d->fail_search) return swa::fail(SWA_ENOMEM, "stand-in failure");
  std::vector<C6> v;
  int64_t t = 0, o = 0;
  collect(d, q, qlen, 0, lo, hi, v, &t, &o);
  std::sort(v.begin(), v.end(), c6_before);
  *n = std::min<int64_t>(keep, int64_t(v.size()));
  for (int64_t i = 0; i < *n; ++i) hits[i] = {v[size_t(i)].seqno, v[size_t(i)].score};
  if (tot) *tot = t;
  if (obv) *obv = o;
  return swa_search(d, q, qlen, nullptr, c);
}
int swa_search_pair_topk(swa_db* d, const uint8_t* q1, int64_t l1, const uint8_t* q2, int64_t l2, int64_t k1, int64_t lo1, int64_t hi1, int64_t k2, int64_t lo2, int64_t hi2,
                         swa_hit_t* h1, int64_t* n1, int64_t* t1, int64_t* o1, swa_hit_t* h2, int64_t* n2, int64_t* t2, int64_t* o2, swa_counters_t* c)
{
  int rc = swa_search_topk(d, q1, l1, k1, lo1, hi1, h1, n1, t1, o1, c);
  if (rc == SWA_OK) rc = swa_search_topk(d, q2, l2, k2, lo2, hi2, h2, n2, t2, o2, nullptr);
  return rc;
}
int swa_search_frames_topk(swa_db* d, int nq, const uint8_t* const* qs, const int64_t* ql, const int32_t* tags, int64_t keep, int64_t lo, int64_t hi, swa_fhit_t* hits, int64_t* n,
                           int64_t* tot, int64_t* obv, swa_counters_t* c)
{
  if (d->fail_search) return swa::fail(SWA_ENOMEM, "stand-in failure");
  std::vector<C6> v;
  int64_t t = 0, o = 0;
  for (int i = 0; i < nq; ++i) collect(d, qs[i], ql[i], i, lo, hi, v, &t, &o);
  std::sort(v.begin(), v.end(), c6_before);
  *n = std::min<int64_t>(keep, int64_t(v.size()));
  for (int64_t i = 0; i < *n; ++i) {
    const C6& x = v[size_t(i)];
    const int tag = tags ? tags[x.which] : 0;
    hits[i] = {x.seqno, x.score, tag / 3, tag % 3, x.dtag / 3, x.dtag % 3};
  }
  if (tot) *tot = t;
  if (obv) *obv = o;
  return swa_search(d, qs[0], ql[0], nullptr, c);
}
int swa_align_hits(swa_db* d, const uint8_t* q, int64_t qlen, const int64_t* seqnos, const int32_t* ds, const int32_t* df, int64_t n, swa_alignment_t* out, char* text, int64_t cap, int64_t* used)
{
  const uint64_t qh = qhash(q, qlen);
  std::string all;
  for (int64_t i = 0; i < n; ++i) {
    if (seqnos[i] < d->first || seqnos[i] >= d->first + d->nseq) return swa::fail(SWA_EINVAL, "sequence number outside this shard");
    std::memset(&out[i], 0, sizeof out[i]);
    out[i].seqno = seqnos[i]; out[i].dstrand = ds ? ds[i] : 0; out[i].dframe = df ? df[i] : 0;
    out[i].score = fake_score(d, seqnos[i], 3 * out[i].dstrand + out[i].dframe, qh);
    const std::string s = "M" + std::to_string(seqnos[i] % 977 + 1) + std::string(size_t(seqnos[i] % 3 == 0 ? 40000 : 0), 'I');   // some scripts outgrow the first buffer
    out[i].cigar_offset = int64_t(all.size()); out[i].cigar_len = int64_t(s.size());
    all += s; all += '\0';
  }
  *used = int64_t(all.size());
  if (*used > cap) return swa::fail(SWA_ERANGE, "text buffer too small for the edit scripts");
  std::memcpy(text, all.data(), all.size());
  return SWA_OK;
}
int swa_db_sequence(swa_db* d, int64_t seqno, int, int, uint8_t* buf, int64_t cap, int64_t* len, int64_t* ntlen)
{
  if (seqno < d->first || seqno >= d->first + d->nseq) return swa::fail(SWA_EINVAL, "sequence number outside this shard");
  const int64_t l = seqno - d->first;
  *len = d->off[size_t(l) + 1] - d->off[size_t(l)];
  if (ntlen) *ntlen = 0;
  if (*len > cap) return swa::fail(SWA_ERANGE, "sequence buffer too small");
  std::memset(buf, int(seqno % 251), size_t(*len));
  return SWA_OK;
}
}
