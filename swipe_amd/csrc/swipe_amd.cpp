// swipe_amd host library: the C ABI of include/swipe_amd.h on top of the gfx950 kernels.
//
// Replaces the reference's search_chunk() seam (swipe.cc:1365-1596): where the reference
// walks a chunk of sequence numbers through search7 -> search16 -> fullsw and calls
// hits_enter() per sequence, this library keeps the whole shard resident in HBM in two forms
//   raw      residues[] + offsets[]               (database order; re-queue passes gather from it)
//   stream   batch-interleaved residue pairs      (what the first-pass kernel streams)
// runs the packed-f16 kernel over every batch, re-queues the flagged sequences to the 32-bit
// and 64-bit kernels, and reduces to the hit list on device.
#include "../../include/swipe_amd.h"
#include "host_util.h"
#include "sw_device.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

extern "C" {
int swa_narrow_rows_for(int qlen);
int swa_wide_rows_for(int qlen);
hipError_t swa_launch_narrow(int K, const swa_narrow_params* p, int blocks, hipStream_t st);
hipError_t swa_launch_wide(int K, int bits, const swa_wide_params* p, int blocks, hipStream_t st);
hipError_t swa_launch_format(const uint8_t* residues, const int64_t* offsets, const int32_t* slots,
                             const swa_batch* batches, int nbatches, uint16_t* stream, hipStream_t st);
hipError_t swa_launch_filter(const int* scores, const long long* scores64, int n, long long minscore,
                             long long maxscore, int* cand_count, int cand_cap, int* cand_idx,
                             long long* cand_score, unsigned long long* tallies, hipStream_t st);
}

namespace swa {
static thread_local std::string g_last_error;
int fail(int code, const std::string& msg)
{
  g_last_error = msg;
  return code;
}
}  // namespace swa
using swa::fail;

#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess)                                                                     \
      return fail(e_ == hipErrorOutOfMemory ? SWA_ENOMEM : SWA_ENODEV,                       \
                  std::string(#expr) + ": " + hipGetErrorString(e_));                        \
  } while (0)

namespace {

template <typename T> struct DevBuf {
  T* p = nullptr;
  size_t cap = 0;
  ~DevBuf() { if (p) (void)hipFree(p); }
  hipError_t reserve(size_t n)
  {
    if (n <= cap) return hipSuccess;
    if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), (n ? n : 1) * sizeof(T));
    if (e == hipSuccess) cap = n;
    return e;
  }
  size_t bytes() const { return cap * sizeof(T); }
};

// A set of batches over some sequences + its formatted stream
struct BatchSet {
  DevBuf<int32_t> slots;
  DevBuf<swa_batch> batches;
  DevBuf<uint16_t> stream;
  int nbatches = 0;
  int64_t chunks = 0;
};

uint16_t f16_bits(float f)
{
  _Float16 h = (_Float16)f;
  uint16_t b;
  std::memcpy(&b, &h, 2);
  return b;
}
}  // namespace

struct swa_db {
  int device = 0;
  int symtype = SWA_SYMTYPE_PROTEIN;
  int cus = 256;
  int64_t nseq = 0, nsym = 0, longest = 0, first_seqno = 0, total_seq = 0, total_sym = 0;
  std::vector<int64_t> h_offsets;
  std::vector<int32_t> h_order;            // sequence indices by descending length
  hipStream_t stream = nullptr;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};

  DevBuf<uint8_t> residues;
  DevBuf<int64_t> offsets;
  BatchSet main;                           // all sequences, two per DPP row
  BatchSet scratch;                        // re-queued sequences, one per DPP row
  DevBuf<int32_t> scores;
  DevBuf<long long> scores64;
  DevBuf<int32_t> ovf_list;
  DevBuf<int32_t> ctl;                     // [0] work counter, [1] overflow count, [2] candidate count
  DevBuf<unsigned long long> tallies;      // totalhits, obvious
  DevBuf<int32_t> cand_idx;
  DevBuf<long long> cand_score;
  DevBuf<uint8_t> qseq;
  DevBuf<int32_t> matrix;
  DevBuf<swa_query> query;

  bool scoring_set = false;
  int32_t h_matrix[1024];
  int64_t goe = 0, ge = 0, hi = 0, lo = 0;
  bool searched = false;
  int narrow_variant = 0;                  // 0 auto, 1 plain, 2 row-shifted (SWA_NARROW_VARIANT, for A/B runs)

  ~swa_db()
  {
    for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e);
    if (stream) (void)hipStreamDestroy(stream);
  }
  size_t hbm_bytes() const
  {
    return residues.bytes() + offsets.bytes() + main.slots.bytes() + main.batches.bytes() + main.stream.bytes() +
           scratch.slots.bytes() + scratch.batches.bytes() + scratch.stream.bytes() + scores.bytes() +
           scores64.bytes() + ovf_list.bytes() + cand_idx.bytes() + cand_score.bytes();
  }
};

namespace {

// Lay `ids` (already ordered by descending length) out as batches with `per_row` sequences per
// DPP row (2 = packed pairs for the f16 kernel, 1 = slot A only for the wide kernels), upload,
// and run the formatting kernel.
int build_batches(swa_db* db, const int32_t* ids, int64_t n, int per_row, BatchSet& bs)
{
  const int per_batch = 4 * per_row;
  const int64_t nb = (n + per_batch - 1) / per_batch;
  if (nb > 0x7fffffff) return fail(SWA_EINVAL, "too many batches for one shard");
  std::vector<int32_t> slots(size_t(nb) * SWA_SLOTS, -1);
  std::vector<swa_batch> batches(static_cast<size_t>(nb));
  uint64_t chunk_total = 0;
  for (int64_t b = 0; b < nb; ++b) {
    int64_t longest = 0;
    for (int j = 0; j < per_batch; ++j) {
      const int64_t i = b * per_batch + j;
      if (i >= n) break;
      const int32_t id = ids[i];
      const int row = j / per_row, half = j % per_row;
      slots[size_t(b) * SWA_SLOTS + row * 2 + half] = id;
      const int64_t len = db->h_offsets[id + 1] - db->h_offsets[id];
      if (len > longest) longest = len;
    }
    const int64_t steps = (longest + 1) & ~int64_t(1);
    const int64_t nchunks = (steps + 15) / 16;
    if (chunk_total > 0xffffffffull || steps > 0x7ffffff0) return fail(SWA_EINVAL, "residue stream exceeds 2^32 chunks");
    batches[size_t(b)].offset = uint32_t(chunk_total);
    batches[size_t(b)].steps = int32_t(steps);
    chunk_total += uint64_t(nchunks);
  }
  HIP_TRY(bs.slots.reserve(slots.size()));
  HIP_TRY(bs.batches.reserve(batches.size()));
  HIP_TRY(bs.stream.reserve(size_t(chunk_total) * 64));
  if (nb) {
    HIP_TRY(hipMemcpyAsync(bs.slots.p, slots.data(), slots.size() * sizeof(int32_t), hipMemcpyHostToDevice, db->stream));
    HIP_TRY(hipMemcpyAsync(bs.batches.p, batches.data(), batches.size() * sizeof(swa_batch), hipMemcpyHostToDevice, db->stream));
    HIP_TRY(swa_launch_format(db->residues.p, db->offsets.p, bs.slots.p, bs.batches.p, int(nb), bs.stream.p, db->stream));
    HIP_TRY(hipStreamSynchronize(db->stream));     // host vectors go out of scope
  }
  bs.nbatches = int(nb);
  bs.chunks = int64_t(chunk_total);
  return SWA_OK;
}

int persistent_blocks(const swa_db* db, int nbatches)
{
  int blocks = (nbatches + 3) / 4;                   // 4 waves per block, one batch per wave at a time
  const int cap = db->cus * 8;
  if (blocks > cap) blocks = cap;
  return blocks < 1 ? 1 : blocks;
}

// sort sequence indices by (length desc, index asc): counting sort when lengths are modest
void order_by_length(const std::vector<int64_t>& off, const int32_t* ids, int64_t n, std::vector<int32_t>& out)
{
  out.resize(size_t(n));
  int64_t longest = 0;
  for (int64_t i = 0; i < n; ++i) {
    const int32_t id = ids ? ids[i] : int32_t(i);
    longest = std::max(longest, off[id + 1] - off[id]);
  }
  if (longest <= (int64_t(1) << 24) && n > 1024) {
    std::vector<int64_t> count(size_t(longest) + 2, 0);
    for (int64_t i = 0; i < n; ++i) {
      const int32_t id = ids ? ids[i] : int32_t(i);
      ++count[size_t(longest - (off[id + 1] - off[id])) + 1];
    }
    for (size_t k = 1; k < count.size(); ++k) count[k] += count[k - 1];
    for (int64_t i = 0; i < n; ++i) {
      const int32_t id = ids ? ids[i] : int32_t(i);
      out[size_t(count[size_t(longest - (off[id + 1] - off[id]))]++)] = id;
    }
  } else {
    for (int64_t i = 0; i < n; ++i) out[size_t(i)] = ids ? ids[i] : int32_t(i);
    std::stable_sort(out.begin(), out.end(), [&](int32_t a, int32_t b) {
      return off[a + 1] - off[a] > off[b + 1] - off[b];
    });
  }
}

int ingest(swa_db* db, const uint8_t* residues, const int64_t* offsets, int64_t nseq)
{
  if (nseq > 0x7ffffff0) return fail(SWA_EINVAL, "more than 2^31 sequences in one shard; shard the database");
  db->nseq = nseq;
  db->h_offsets.assign(offsets, offsets + nseq + 1);
  const int64_t base = db->h_offsets[0];
  for (int64_t& o : db->h_offsets) o -= base;
  db->nsym = db->h_offsets[size_t(nseq)];
  db->longest = 0;
  for (int64_t s = 0; s < nseq; ++s) {
    const int64_t len = db->h_offsets[s + 1] - db->h_offsets[s];
    if (len < 0) return fail(SWA_EINVAL, "sequence offsets must be non-decreasing");
    db->longest = std::max(db->longest, len);
  }
  HIP_TRY(hipSetDevice(db->device));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, db->device));
  db->cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  HIP_TRY(hipStreamCreate(&db->stream));
  for (hipEvent_t& e : db->ev) HIP_TRY(hipEventCreate(&e));
  HIP_TRY(db->residues.reserve(size_t(db->nsym) + 16));
  HIP_TRY(db->offsets.reserve(size_t(nseq) + 1));
  if (db->nsym) HIP_TRY(hipMemcpyAsync(db->residues.p, residues + base, size_t(db->nsym), hipMemcpyHostToDevice, db->stream));
  HIP_TRY(hipMemcpyAsync(db->offsets.p, db->h_offsets.data(), (size_t(nseq) + 1) * sizeof(int64_t), hipMemcpyHostToDevice, db->stream));
  HIP_TRY(db->scores.reserve(size_t(nseq)));
  HIP_TRY(db->ovf_list.reserve(size_t(nseq)));
  HIP_TRY(db->ctl.reserve(8));
  HIP_TRY(db->tallies.reserve(2));
  HIP_TRY(db->matrix.reserve(1024));
  HIP_TRY(db->query.reserve(1));
  order_by_length(db->h_offsets, nullptr, nseq, db->h_order);
  return build_batches(db, db->h_order.data(), nseq, 2, db->main);
}

struct SearchTimes { float narrow_ms = 0, total_ms = 0; };

// the escalation loop: packed f16 -> 32 bit -> 64 bit.  Scores end up in db->scores / scores64.
int run_search(swa_db* db, const uint8_t* query, int64_t qlen, swa_counters_t* counters)
{
  if (!db) return fail(SWA_EINVAL, "null database handle");
  if (!db->scoring_set) return fail(SWA_ESTATE, "swa_set_scoring must be called before searching");
  if (qlen < 0 || (qlen > 0 && !query)) return fail(SWA_EINVAL, "bad query");
  if (qlen > 1024) return fail(SWA_EINVAL, "queries longer than 1024 residues are not supported yet");
  for (int64_t i = 0; i < qlen; ++i)
    if (query[i] >= 32) return fail(SWA_EINVAL, "query symbol code out of range (must be < 32)");
  HIP_TRY(hipSetDevice(db->device));
  hipStream_t st = db->stream;
  swa_counters_t c{};
  c.cells = db->nsym * qlen;
  HIP_TRY(hipEventRecord(db->ev[0], st));
  if (qlen == 0 || db->nseq == 0) {
    if (db->nseq) HIP_TRY(hipMemsetAsync(db->scores.p, 0, size_t(db->nseq) * sizeof(int32_t), st));
    HIP_TRY(hipEventRecord(db->ev[1], st));
    HIP_TRY(hipEventRecord(db->ev[2], st));
    HIP_TRY(hipEventRecord(db->ev[3], st));
    HIP_TRY(hipStreamSynchronize(st));
    db->searched = true;
    if (counters) *counters = c;
    return SWA_OK;
  }
  HIP_TRY(db->qseq.reserve(size_t(qlen)));
  HIP_TRY(hipMemcpyAsync(db->qseq.p, query, size_t(qlen), hipMemcpyHostToDevice, st));
  swa_query hq{db->qseq.p, db->matrix.p, int32_t(qlen)};
  HIP_TRY(hipMemcpyAsync(db->query.p, &hq, sizeof hq, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemsetAsync(db->ctl.p, 0, 8 * sizeof(int32_t), st));

  // f16 pairs are exact while every value stays within +-2048: needs modest scores and penalties
  const bool narrow_ok = db->hi >= 0 && db->hi < 1024 && db->lo > -1024 && db->goe >= 0 && db->goe <= 1024 &&
                         db->ge >= 0 && db->ge <= 1024 && swa_narrow_rows_for(int(qlen)) > 0;
  std::vector<int32_t> requeue;
  HIP_TRY(hipEventRecord(db->ev[1], st));
  if (narrow_ok) {
    swa_narrow_params p{};
    p.query = db->query.p;
    p.stream = db->main.stream.p;
    p.batches = db->main.batches.p;
    p.slots = db->main.slots.p;
    p.nbatches = db->main.nbatches;
    p.counter = db->ctl.p + 0;
    p.scores = db->scores.p;
    p.limit = int32_t(2048 - db->hi);
    p.ovf_count = db->ctl.p + 1;
    p.ovf_list = db->ovf_list.p;
    const int K = swa_narrow_rows_for(int(qlen));
    auto pair = [](float v) { const uint32_t b = f16_bits(v); return b | b << 16; };
    p.negQ = pair(-float(db->goe));
    p.negR = pair(-float(db->ge));
    // row-shifted form: values carry up to K*R extra, so its exact range ends K*R earlier
    const int64_t shifted_limit = 2048 - db->hi - int64_t(K + 1) * db->ge;
    p.shifted = (db->narrow_variant != 1 && K <= 48 && db->goe >= db->ge && shifted_limit >= 1024) ? 1 : 0;
    if (db->narrow_variant == 2 && !p.shifted) return fail(SWA_EINVAL, "row-shifted kernel not applicable to this scoring");
    if (p.shifted) {
      p.limit = int32_t(shifted_limit);
      p.gapextend_f = float(db->ge);
      p.negQR = pair(-float(db->goe - db->ge));
      p.negKR = pair(-float(int64_t(K) * db->ge));
      for (int r = 0; r <= K + 1; ++r) p.rowc[r] = pair(float(int64_t(r) * db->ge));
    }
    int blocks = persistent_blocks(db, p.nbatches);
    c.narrow_rows = K;
    c.narrow_shifted = p.shifted;
    HIP_TRY(swa_launch_narrow(K, &p, blocks, st));
    HIP_TRY(hipEventRecord(db->ev[2], st));
    int32_t novf = 0;
    HIP_TRY(hipMemcpyAsync(&novf, db->ctl.p + 1, sizeof novf, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    c.narrow = db->nseq;
    if (novf > 0) {
      requeue.resize(size_t(novf));
      HIP_TRY(hipMemcpy(requeue.data(), db->ovf_list.p, size_t(novf) * sizeof(int32_t), hipMemcpyDeviceToHost));
      std::sort(requeue.begin(), requeue.end());       // deterministic order whatever the wave timing was
    }
  } else {
    HIP_TRY(hipEventRecord(db->ev[2], st));
    requeue.assign(db->h_order.begin(), db->h_order.end());
  }

  for (int bits = 32; !requeue.empty() && bits <= 64; bits += 32) {
    std::vector<int32_t> ordered;
    order_by_length(db->h_offsets, requeue.data(), int64_t(requeue.size()), ordered);
    int rc = build_batches(db, ordered.data(), int64_t(ordered.size()), 1, db->scratch);
    if (rc != SWA_OK) return rc;
    if (bits == 64) HIP_TRY(db->scores64.reserve(size_t(db->nseq)));
    HIP_TRY(hipMemsetAsync(db->ctl.p, 0, 2 * sizeof(int32_t), st));
    swa_wide_params p{};
    p.query = db->query.p;
    p.stream = db->scratch.stream.p;
    p.batches = db->scratch.batches.p;
    p.slots = db->scratch.slots.p;
    p.nbatches = db->scratch.nbatches;
    p.counter = db->ctl.p + 0;
    p.scores = db->scores.p;
    p.scores64 = db->scores64.p;
    p.limit = (1ll << 31) - db->hi;
    p.ovf_count = db->ctl.p + 1;
    p.ovf_list = db->ovf_list.p;
    p.gapopenextend = db->goe;
    p.gapextend = db->ge;
    const int K = swa_wide_rows_for(int(qlen));
    if (!K) return fail(SWA_EINVAL, "query too long for the wide kernel");
    HIP_TRY(swa_launch_wide(K, bits, &p, persistent_blocks(db, p.nbatches), st));
    (bits == 32 ? c.wide : c.full) = int64_t(requeue.size());
    requeue.clear();
    if (bits == 32) {
      int32_t novf = 0;
      HIP_TRY(hipMemcpyAsync(&novf, db->ctl.p + 1, sizeof novf, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      if (novf > 0) {
        requeue.resize(size_t(novf));
        HIP_TRY(hipMemcpy(requeue.data(), db->ovf_list.p, size_t(novf) * sizeof(int32_t), hipMemcpyDeviceToHost));
        std::sort(requeue.begin(), requeue.end());
      }
    }
  }
  HIP_TRY(hipEventRecord(db->ev[3], st));
  HIP_TRY(hipStreamSynchronize(st));
  float ms = 0;
  HIP_TRY(hipEventElapsedTime(&ms, db->ev[1], db->ev[2]));
  c.kernel_ms = ms;
  HIP_TRY(hipEventElapsedTime(&ms, db->ev[0], db->ev[3]));
  c.total_ms = ms;
  db->searched = true;
  if (counters) *counters = c;
  return SWA_OK;
}

bool hit_before(const swa_hit_t& a, const swa_hit_t& b)   // hits.cc:188-190: score desc, then seqno desc
{
  return a.score > b.score || (a.score == b.score && a.seqno > b.seqno);
}
}  // namespace

extern "C" const char* swa_last_error(void) { return swa::g_last_error.c_str(); }

extern "C" int swa_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

extern "C" int swa_db_from_memory(const uint8_t* residues, const int64_t* offsets, int64_t nseq, int symtype,
                                  int device, int64_t first_seqno, int64_t total_seqcount,
                                  int64_t total_symcount, swa_db** out)
{
  if (!out) return fail(SWA_EINVAL, "null output handle");
  *out = nullptr;
  if (nseq < 0 || !offsets || (!residues && nseq > 0 && offsets[nseq] > offsets[0]))
    return fail(SWA_EINVAL, "bad database arrays");
  if (symtype != SWA_SYMTYPE_PROTEIN && symtype != SWA_SYMTYPE_NUCLEOTIDE)
    return fail(SWA_EINVAL, "symtype must be 0 (nucleotide) or 1 (protein)");
  if (device < 0 || device >= swa_device_count())
    return fail(SWA_ENODEV, "no such HIP device (swipe_amd has no CPU fallback)");
  swa_db* db = new (std::nothrow) swa_db;
  if (!db) return fail(SWA_ENOMEM, "out of host memory");
  db->device = device;
  db->symtype = symtype;
  if (const char* v = std::getenv("SWA_NARROW_VARIANT")) db->narrow_variant = std::atoi(v);
  db->first_seqno = first_seqno;
  const int rc = ingest(db, residues, offsets, nseq);
  if (rc != SWA_OK) { delete db; return rc; }
  db->total_seq = total_seqcount > 0 ? total_seqcount : db->nseq;
  db->total_sym = total_symcount > 0 ? total_symcount : db->nsym;
  *out = db;
  return SWA_OK;
}

extern "C" int swa_db_open(const char* basename, int symtype, int device, int64_t first_seqno,
                           int64_t last_seqno, swa_db** out)
{
  if (!out) return fail(SWA_EINVAL, "null output handle");
  *out = nullptr;
  swa::HostDb h;
  const int rc = swa::read_blast_db(basename, symtype, first_seqno, last_seqno, h);
  if (rc != SWA_OK) return rc;
  return swa_db_from_memory(h.residues.data(), h.offsets.data(), int64_t(h.offsets.size()) - 1, symtype, device,
                            h.first_seqno, h.total_seqcount, h.total_symcount, out);
}

extern "C" int swa_blastdb_read(const char* basename, int symtype, int64_t first_seqno, int64_t last_seqno,
                                uint8_t** residues, int64_t** offsets, int64_t* nseq, int64_t* total_seqcount,
                                int64_t* total_symcount, int64_t* longest)
{
  if (!residues || !offsets || !nseq) return fail(SWA_EINVAL, "null output");
  swa::HostDb h;
  const int rc = swa::read_blast_db(basename, symtype, first_seqno, last_seqno, h);
  if (rc != SWA_OK) return rc;
  *nseq = int64_t(h.offsets.size()) - 1;
  *residues = static_cast<uint8_t*>(std::malloc(h.residues.size() ? h.residues.size() : 1));
  *offsets = static_cast<int64_t*>(std::malloc(h.offsets.size() * sizeof(int64_t)));
  if (!*residues || !*offsets) { std::free(*residues); std::free(*offsets); return fail(SWA_ENOMEM, "out of host memory"); }
  if (!h.residues.empty()) std::memcpy(*residues, h.residues.data(), h.residues.size());
  std::memcpy(*offsets, h.offsets.data(), h.offsets.size() * sizeof(int64_t));
  if (total_seqcount) *total_seqcount = h.total_seqcount;
  if (total_symcount) *total_symcount = h.total_symcount;
  if (longest) *longest = h.longest;
  return SWA_OK;
}

extern "C" void swa_free(void* p) { std::free(p); }

extern "C" int swa_db_info(const swa_db* db, swa_db_info_t* info)
{
  if (!db || !info) return fail(SWA_EINVAL, "null argument");
  info->seqcount = db->nseq;
  info->symcount = db->nsym;
  info->longest = db->longest;
  info->first_seqno = db->first_seqno;
  info->total_seqcount = db->total_seq;
  info->total_symcount = db->total_sym;
  info->hbm_bytes = int64_t(db->hbm_bytes());
  return SWA_OK;
}

extern "C" void swa_db_close(swa_db* db)
{
  if (!db) return;
  (void)hipSetDevice(db->device);
  delete db;
}

extern "C" int swa_set_scoring(swa_db* db, const int64_t* matrix, int64_t gapopenextend, int64_t gapextend)
{
  if (!db || !matrix) return fail(SWA_EINVAL, "null argument");
  int64_t lo = 100, hi = -100;                       // matrices.cc:561-572
  for (int i = 0; i < 1024; ++i) {
    if (matrix[i] > 0x3fffffff || matrix[i] < -0x3fffffff) return fail(SWA_EINVAL, "matrix entry out of range");
    db->h_matrix[i] = int32_t(matrix[i]);
    lo = std::min(lo, matrix[i]);
    hi = std::max(hi, matrix[i]);
  }
  if (gapopenextend < 0 || gapextend < 0 || gapopenextend > 0x3fffffff || gapextend > 0x3fffffff)
    return fail(SWA_EINVAL, "gap penalties out of range");
  db->lo = lo;
  db->hi = hi;
  db->goe = gapopenextend;
  db->ge = gapextend;
  HIP_TRY(hipSetDevice(db->device));
  HIP_TRY(hipMemcpyAsync(db->matrix.p, db->h_matrix, sizeof db->h_matrix, hipMemcpyHostToDevice, db->stream));
  HIP_TRY(hipStreamSynchronize(db->stream));
  db->scoring_set = true;
  return SWA_OK;
}

extern "C" int swa_search(swa_db* db, const uint8_t* query, int64_t qlen, int64_t* scores, swa_counters_t* counters)
{
  const int rc = run_search(db, query, qlen, counters);
  if (rc != SWA_OK || !scores || db->nseq == 0) return rc;
  std::vector<int32_t> s32(size_t(db->nseq));
  HIP_TRY(hipMemcpy(s32.data(), db->scores.p, s32.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
  std::vector<long long> s64;
  for (int64_t i = 0; i < db->nseq; ++i) {
    if (s32[size_t(i)] == SWA_SCORE_IN_64) {
      if (s64.empty()) {
        s64.resize(size_t(db->nseq));
        HIP_TRY(hipMemcpy(s64.data(), db->scores64.p, s64.size() * sizeof(long long), hipMemcpyDeviceToHost));
      }
      scores[i] = s64[size_t(i)];
    } else {
      scores[i] = s32[size_t(i)];
    }
  }
  return SWA_OK;
}

extern "C" int swa_search_topk(swa_db* db, const uint8_t* query, int64_t qlen, int64_t keep, int64_t minscore,
                               int64_t maxscore, swa_hit_t* hits, int64_t* nhits, int64_t* totalhits,
                               int64_t* obvious, swa_counters_t* counters)
{
  if (keep < 0 || (keep > 0 && !hits) || !nhits) return fail(SWA_EINVAL, "bad hit buffer");
  const int rc = run_search(db, query, qlen, counters);
  if (rc != SWA_OK) return rc;
  *nhits = 0;
  if (totalhits) *totalhits = 0;
  if (obvious) *obvious = 0;
  if (db->nseq == 0) return SWA_OK;
  hipStream_t st = db->stream;
  const int cap = int(std::min<int64_t>(db->nseq, std::max<int64_t>(1 << 20, 8 * keep)));
  HIP_TRY(db->cand_idx.reserve(size_t(cap)));
  HIP_TRY(db->cand_score.reserve(size_t(cap)));
  HIP_TRY(hipMemsetAsync(db->ctl.p + 2, 0, sizeof(int32_t), st));
  HIP_TRY(hipMemsetAsync(db->tallies.p, 0, 2 * sizeof(unsigned long long), st));
  HIP_TRY(swa_launch_filter(db->scores.p, db->scores64.p, int(db->nseq), minscore, maxscore, db->ctl.p + 2, cap,
                            db->cand_idx.p, db->cand_score.p, db->tallies.p, st));
  int32_t ncand = 0;
  unsigned long long tl[2] = {0, 0};
  HIP_TRY(hipMemcpyAsync(&ncand, db->ctl.p + 2, sizeof ncand, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(tl, db->tallies.p, sizeof tl, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  if (totalhits) *totalhits = int64_t(tl[0]);
  if (obvious) *obvious = int64_t(tl[1]);
  std::vector<swa_hit_t> cand;
  if (ncand <= cap) {
    std::vector<int32_t> idx((size_t(ncand)));
    std::vector<long long> sc((size_t(ncand)));
    if (ncand) {
      HIP_TRY(hipMemcpy(idx.data(), db->cand_idx.p, idx.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
      HIP_TRY(hipMemcpy(sc.data(), db->cand_score.p, sc.size() * sizeof(long long), hipMemcpyDeviceToHost));
    }
    cand.resize(size_t(ncand));
    for (int i = 0; i < ncand; ++i) cand[size_t(i)] = {db->first_seqno + idx[size_t(i)], sc[size_t(i)]};
  } else {
    // more candidates than the compaction buffer: take every score to the host instead
    std::vector<int64_t> all(size_t(db->nseq));
    std::vector<int32_t> s32(size_t(db->nseq));
    HIP_TRY(hipMemcpy(s32.data(), db->scores.p, s32.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    std::vector<long long> s64;
    for (int64_t i = 0; i < db->nseq; ++i) {
      int64_t v = s32[size_t(i)];
      if (v == SWA_SCORE_IN_64) {
        if (s64.empty()) {
          s64.resize(size_t(db->nseq));
          HIP_TRY(hipMemcpy(s64.data(), db->scores64.p, s64.size() * sizeof(long long), hipMemcpyDeviceToHost));
        }
        v = s64[size_t(i)];
      }
      if (v >= minscore && v <= maxscore) cand.push_back({db->first_seqno + i, v});
    }
  }
  const size_t k = std::min<size_t>(size_t(keep), cand.size());
  std::partial_sort(cand.begin(), cand.begin() + k, cand.end(), hit_before);
  for (size_t i = 0; i < k; ++i) hits[i] = cand[i];
  *nhits = int64_t(k);
  return SWA_OK;
}

extern "C" int swa_hits_merge(const swa_hit_t* lists, const int64_t* counts, int nlists, int64_t stride,
                              int64_t keep, swa_hit_t* out, int64_t* nout)
{
  if (!lists || !counts || nlists < 0 || !nout || (keep > 0 && !out)) return fail(SWA_EINVAL, "bad argument");
  std::vector<swa_hit_t> all;
  for (int l = 0; l < nlists; ++l)
    for (int64_t i = 0; i < counts[l]; ++i) all.push_back(lists[int64_t(l) * stride + i]);
  std::stable_sort(all.begin(), all.end(), hit_before);
  const size_t k = std::min<size_t>(size_t(keep), all.size());
  for (size_t i = 0; i < k; ++i) out[i] = all[i];
  *nout = int64_t(k);
  return SWA_OK;
}
