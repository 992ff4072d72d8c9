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
#include "traceback.h"
#include "kernel_choice.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cctype>
#include <condition_variable>
#include <mutex>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <string>
#include <thread>
#include <vector>

extern "C" {
int swa_narrow_rows_for(int qlen);
hipError_t swa_launch_narrow_split(int G, int K, const swa_narrow_params* p, int blocks, hipStream_t st);
hipError_t swa_launch_narrow_one_a(int K, const swa_narrow_params* p, int blocks, hipStream_t st);
hipError_t swa_launch_narrow_one_b(int K, const swa_narrow_params* p, int blocks, hipStream_t st);
hipError_t swa_launch_one_bound_c(int K, const swa_narrow_params* p, int blocks, hipStream_t st);
hipError_t swa_launch_one_bound_d(int K, const swa_narrow_params* p, int blocks, hipStream_t st);
hipError_t swa_launch_one_bound_e(int K, const swa_narrow_params* p, int blocks, hipStream_t st);
hipError_t swa_launch_narrow_pass(int K, const swa_narrow_params* p, int blocks, hipStream_t st);
hipError_t swa_launch_dual_pass(int K, int nres, const swa_mp_params* p, int cus, hipStream_t st);
hipError_t swa_launch_dual_bound(int G, int K, const swa_mp_params* p, int cus, hipStream_t st);
hipError_t swa_launch_narrow_bound_pass(int K, const swa_narrow_params* p, int blocks, hipStream_t st);
hipError_t swa_launch_narrow_bound_g2(int K, const swa_narrow_params* p, int blocks, hipStream_t st);
hipError_t swa_launch_narrow_bound_g4(int K, const swa_narrow_params* p, int blocks, hipStream_t st);
hipError_t swa_launch_narrow_bound_g8(int K, const swa_narrow_params* p, int blocks, hipStream_t st);
hipError_t swa_launch_narrow_bound_long2(int K, const swa_narrow_params* p, int blocks, hipStream_t st);
hipError_t swa_launch_narrow_bound_long4(int K, const swa_narrow_params* p, int blocks, hipStream_t st);
hipError_t swa_launch_narrow_bound_long8(int K, const swa_narrow_params* p, int blocks, hipStream_t st);
hipError_t swa_launch_narrow_bound_g16(int K, const swa_narrow_params* p, int blocks, hipStream_t st);
int swa_bound_period(void);
hipError_t swa_launch_gather(const swa_seqs* sq, const int* ids, const int64_t* out_off, int n, uint8_t* out, hipStream_t st);
hipError_t swa_launch_narrow(int K, const swa_narrow_params* p, int blocks, hipStream_t st);
int swa_mp_waves(int mode, int K);
hipError_t swa_launch_endpoints(const swa_seqs* sq, const int32_t* ids,
                                const uint8_t* minus, int n, const uint8_t* qseq, int qlen, const int32_t* matrix, long long Q, long long R,
                                long long* Hs, long long* Es, long long* out, hipStream_t st);
int swa_endpoints_rows_for(int qlen);
hipError_t swa_launch_endpoints_wave(const swa_seqs* sq, const int32_t* ids,
                                     const uint8_t* minus, int n, const uint8_t* qseq, int qlen, const int32_t* matrix,
                                     int Q, int R, int* bh, int* bf, const int64_t* boff, long long* out, int* scores,
                                     hipStream_t st);
hipError_t swa_launch_mark_excluded(int* scores, const int* ids, int n, hipStream_t st);
hipError_t swa_launch_translate(const uint8_t* nt, int packed, const int64_t* ntoff, const int64_t* voff, int64_t nv,
                                const uint8_t* table, uint8_t* prot, int64_t total, hipStream_t st);
hipError_t swa_launch_dual(int K, int nres, int G, const swa_mp_params* p, int cus, hipStream_t st);
hipError_t swa_launch_dual_one(int K, int nres, const swa_mp_params* p, int cus, hipStream_t st);
hipError_t swa_launch_mp(int mode, int K, const swa_mp_params* p, int blocks, int threads, hipStream_t st);
hipError_t swa_launch_format(const swa_seqs* sq, const int32_t* slots, const swa_batch* batches, int nbatches, void* stream,
                             int nibbles, hipStream_t st);
hipError_t swa_launch_unterminate(const uint8_t* chunk, long long c0, long long c1, const int64_t* offsets, int s0, int n,
                                  uint8_t* residues, unsigned* flags, hipStream_t st);
hipError_t swa_launch_unpack_nt(const uint8_t* chunk, long long c0, const int64_t* raw, const int64_t* offsets, int s0, int n,
                                uint8_t* residues, hipStream_t st);
hipError_t swa_launch_filter(const int* scores, const long long* scores64, int n, int which, long long minscore,
                             long long maxscore, int* cand_count, int cand_cap, swa_cand* cand,
                             unsigned long long* tallies, hipStream_t st);
hipError_t swa_launch_rebase(const swa_batch* src, swa_batch* dst, int n, uint32_t delta, hipStream_t st);
hipError_t swa_launch_fold(int* scores, long long* scores64, const int32_t* parents, const int32_t* wfirst, int nparents, int nseq,
                           hipStream_t st);
int swa_requeue_block_rows_for(int qlen);
hipError_t swa_launch_requeue_block(const swa_seqs* sq, const int32_t* list, const int32_t* count, int cap, int32_t* work, const uint8_t* qseq, int qlen,
                                    const int32_t* matrix, int Q, int R, int* scores, int blocks, hipStream_t st);
hipError_t swa_launch_requeue_wave(const swa_seqs* sq, const int32_t* list, const int32_t* count,
                                   int cap, int32_t* work, const uint8_t* qseq, int qlen, const int32_t* matrix, int Q, int R,
                                   int* scores, int blocks, hipStream_t st);
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

// ---- red zones --------------------------------------------------------------------------------------
// The kernels index their buffers with hand-computed offsets (p.stream, p.boundary, rowc[], ovf_list, slots, scores ...).
// With SWA_REDZONES=1 in the environment when the library is first used, EVERY device allocation of the library gets a
// guard region of 4 KiB in front of it and one right behind its last requested byte, filled with 0xA5;
// swa_redzones_check() reads all guards of all live allocations back and reports every byte that changed.  A debugging
// facility, process-wide (allocations do not know their handle), off by default, no cost when off.
constexpr size_t REDZONE = 4096;
struct Redzones {
  std::mutex mu;
  struct Entry { unsigned char* raw; size_t bytes; size_t elem; };
  std::vector<Entry> live;
  static bool on()
  {
    static const bool v = [] { const char* e = std::getenv("SWA_REDZONES"); return e && *e && std::strcmp(e, "0") != 0; }();
    return v;
  }
  static Redzones& get() { static Redzones r; return r; }
};

template <typename T> struct DevBuf {
  T* p = nullptr;
  size_t cap = 0;
  unsigned char* raw = nullptr;            // the allocation itself when guarded (p = raw + REDZONE); else null
  ~DevBuf() { release(); }
  hipError_t reserve(size_t n)
  {
    if (n <= cap) return hipSuccess;
    release();
    const size_t bytes = (n ? n : 1) * sizeof(T);
    if (!Redzones::on()) {
      hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), bytes);
      if (e == hipSuccess) cap = n;
      return e;
    }
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&raw), bytes + 2 * REDZONE);
    if (e != hipSuccess) { raw = nullptr; return e; }
    e = hipMemset(raw, 0xA5, REDZONE);
    if (e == hipSuccess) e = hipMemset(raw + REDZONE + bytes, 0xA5, REDZONE);
    if (e != hipSuccess) { (void)hipFree(raw); raw = nullptr; return e; }
    p = reinterpret_cast<T*>(raw + REDZONE);
    cap = n;
    Redzones& r = Redzones::get();
    std::lock_guard<std::mutex> g(r.mu);
    r.live.push_back({raw, bytes, sizeof(T)});
    return hipSuccess;
  }
  size_t bytes() const { return cap * sizeof(T); }
  void swap(DevBuf& o) { std::swap(p, o.p); std::swap(cap, o.cap); std::swap(raw, o.raw); }
  void release()
  {
    if (raw) {
      Redzones& r = Redzones::get();
      {
        std::lock_guard<std::mutex> g(r.mu);
        for (size_t i = 0; i < r.live.size(); ++i)
          if (r.live[i].raw == raw) { r.live[i] = r.live.back(); r.live.pop_back(); break; }
      }
      (void)hipFree(raw);
    } else if (p) {
      (void)hipFree(p);
    }
    p = nullptr; raw = nullptr; cap = 0;
  }
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
};

// A set of batches over some sequences + its formatted stream
struct BatchSet {
  DevBuf<int32_t> slots;
  DevBuf<swa_batch> batches;
  DevBuf<uint16_t> stream;
  int nbatches = 0;
  int64_t chunks = 0;
  std::vector<int32_t> h_steps;             // steps of every batch (non-increasing: batches are cut from a length-sorted list)
  std::vector<int32_t> h_slots;             // host copy of the slot table when the set is NOT the shard's length order cut into
                                            // batches (a streamed-in shard's main set: its parts' batches merged by length); else empty
  bool nibbles = false;                     // one-sequence-per-row stream of a nucleotide shard at 4 bits per base (32-byte chunks)
  bool built = false;
  // a VIEW (see "windows"): tables of its own over two stream regions addressed from one base pointer
  const uint16_t* stream_base = nullptr;
  int nlong = 0;                            // batches [0, nlong) live in the view's own stream region
  int64_t off_long = 0, off_main = 0;       // chunk offset (from the base) of batch 0 / batch nlong
  const uint16_t* sp() const { return stream_base ? stream_base : stream.p; }
  void swap(BatchSet& o)
  {
    slots.swap(o.slots); batches.swap(o.batches); stream.swap(o.stream);
    std::swap(nbatches, o.nbatches); std::swap(chunks, o.chunks); h_steps.swap(o.h_steps); h_slots.swap(o.h_slots);
    std::swap(nibbles, o.nibbles); std::swap(built, o.built); std::swap(stream_base, o.stream_base);
    std::swap(nlong, o.nlong); std::swap(off_long, o.off_long); std::swap(off_main, o.off_main);
  }
};

// Tuning / test knobs of one handle (swa_set_option).  The defaults are what the measurements in DESIGN.md chose;
// none of them changes a result.  A handle takes its initial values from the environment ONCE, when it is created
// (SWA_<KEY>, upper case) - a convenience for the A/B tools; the search path itself never reads the environment.
struct Options {
  int64_t bound = -1;            // bound build of the first pass of top-K searches: -1 auto, 0 never, 1 whenever a build exists
  int64_t lanes = 0;             // lanes per sequence pair (2 / 4 / 8 / 16) if the query fits; 0 = by query length
  int64_t pipe = -1;             // profile-load build of the split kernel: -1 measured best, 0 staged, 1 pipelined, 2 across steps
  int64_t blocks_per_cu = 0;     // persistent blocks per CU of the first-pass kernel; 0 = 8
  int64_t force_mp = 0;          // 1: block-synchronous multi-pass kernel for any query length
  int64_t mp_k = 0, mp_w = 0;    // its rows per lane / waves per SIMD; 0 = default
  int64_t boundary_mb = -1;      // cap of the pass hand-over buffer in MiB; -1 = from free memory
  int64_t wave_requeue = -1;     // re-queued sequences one wave each: -1 when the list is short, 0 never
  int64_t dual_mp = 0;           // 1: multi-pass policy for two queries
  int64_t dual_kmax = 0;         // largest rows-per-lane of the single-pass two-query kernel; 0 = no cap
  int64_t narrow_variant = 0;    // 0 auto, 1 plain (8.5-instruction) form, 2 row-shifted form
  int64_t endpoints_thread = 0;  // 1: one-thread 64-bit end-point kernel instead of the wave kernel
  int64_t requeue_host = 0;      // 1: the host reads the re-queue list before launching the wide kernels (two extra syncs)
  int64_t concat = -1;           // bound builds on chains of lanes: sets of batches a chain works through back to back without draining (sw_cb_kernel.inc); -1 = 16; N > 1 = N; 1 or 0 = every set on its own (the round-3 kernel)
  int64_t concat_tail = -1;      // ... except the last concat_tail sets of the queue, which are handed out one at a time; -1 = four per resident wave
  int64_t twin = -1;             // bound builds at two waves per SIMD: blocks of 8 waves holding the profile twice, the second copy renormalising H on the way (sw_cb_kernel.inc TWIN); 0 = blocks of 4 waves, one copy (the round-3 form); -1 = on
  int64_t requeue_block = 0;     // device-driven re-queue, a BLOCK of four waves per sequence instead of one wave: 1 whenever the query fits (<= 1024 rows), else never. Off until tools/rq_probe.py has priced its barrier per DP step on hardware
  int64_t requeue_follow = 0;    // (rounds 2-3: a re-queue kernel beside the first pass on a second stream; gone - the key is accepted and ignored)
  int64_t window = -1;           // long database sequences as overlapping windows: -1 auto, 0 never, n > 0: every sequence longer than n
  int64_t window_step = 0;       // distance between window starts; 0 = from the query (the overlap is never a knob: it is what makes it exact)
  int64_t long_lanes = 1;        // bound build: chains of 2 / 4 / 8 lanes with up to 62 rows per lane where the query fits them (0: 48)
  int64_t watchdog_s = 0;        // > 0: a search whose stream does not drain within that many seconds fails with the control block in the message
  // the pipelined open (sw_loading.inc); read when the open begins
  int64_t pipelined = 1;         // 0: swa_db_open[_async] always takes the old reader (read everything, then upload)
  int64_t load_part = 0;         // bytes of sequence file per part; 0 = an eighth of the shard, at least 256 MiB
  int64_t load_chunk = 0;        // bytes per page-locked staging chunk; 0 = 64 MiB
  int64_t load_threads = 0;      // reader threads per chunk; 0 = half the hardware threads, 2..16
  int64_t load_delay_ms = 0;     // tests: the loader sleeps that long after every chunk
  int64_t load_trace = 0;        // 1: one line on stderr with the stages of the load
  int64_t stream_reserve = -1;   // budgeted shards (sw_streamed.inc): bytes of a slot set aside for buffers that do not scale with the part; -1 = 8 MiB (tests: small databases in several parts)
};
struct OptionKey { const char* key; int64_t Options::*field; };
const OptionKey kOptionKeys[] = {
  {"bound", &Options::bound}, {"lanes", &Options::lanes}, {"pipe", &Options::pipe},
  {"blocks_per_cu", &Options::blocks_per_cu}, {"force_mp", &Options::force_mp}, {"mp_k", &Options::mp_k},
  {"mp_w", &Options::mp_w}, {"boundary_mb", &Options::boundary_mb}, {"wave_requeue", &Options::wave_requeue},
  {"dual_mp", &Options::dual_mp}, {"dual_kmax", &Options::dual_kmax}, {"narrow_variant", &Options::narrow_variant},
  {"endpoints_thread", &Options::endpoints_thread}, {"requeue_host", &Options::requeue_host},
  {"requeue_follow", &Options::requeue_follow}, {"requeue_block", &Options::requeue_block}, {"concat", &Options::concat}, {"concat_tail", &Options::concat_tail}, {"twin", &Options::twin}, {"window", &Options::window}, {"window_step", &Options::window_step},
  {"long_lanes", &Options::long_lanes}, {"watchdog_s", &Options::watchdog_s}, {"pipelined", &Options::pipelined},
  {"load_part", &Options::load_part}, {"load_chunk", &Options::load_chunk}, {"load_threads", &Options::load_threads},
  {"load_delay_ms", &Options::load_delay_ms}, {"load_trace", &Options::load_trace}, {"stream_reserve", &Options::stream_reserve},
};
bool parse_option_value(const char* key, const char* value, int64_t* out)
{
  if (!value || !*value) return false;
  if (!std::strcmp(key, "endpoints_thread") && !std::strcmp(value, "thread")) { *out = 1; return true; }
  if (!std::strcmp(key, "endpoints_thread") && !std::strcmp(value, "wave")) { *out = 0; return true; }
  char* end = nullptr;
  const long long v = std::strtoll(value, &end, 10);
  if (end == value || *end) return false;
  *out = v;
  return true;
}
void options_from_environment(Options& o)
{
  for (const OptionKey& k : kOptionKeys) {
    std::string name = "SWA_";
    for (const char* c = k.key; *c; ++c) name += char(std::toupper(static_cast<unsigned char>(*c)));
    if (!std::strcmp(k.key, "endpoints_thread")) name = "SWA_ENDPOINTS";
    int64_t v = 0;
    if (const char* e = std::getenv(name.c_str())) if (parse_option_value(k.key, e, &v)) o.*(k.field) = v;
  }
}

uint16_t f16_bits(float f)
{
  _Float16 h = (_Float16)f;
  uint16_t b;
  std::memcpy(&b, &h, 2);
  return b;
}
}  // namespace

struct Streamed;                          // sw_streamed.inc: a database walked through two device slots, part by part
struct Loading;                           // sw_loading.inc: a shard whose residues are still on their way into HBM
struct swa_db {
  std::shared_ptr<Streamed> streamed;     // set on the FRONT handle of a streamed database (it owns no device memory itself)
  std::shared_ptr<Loading> loading;       // set from swa_db_open_async until the first call that finds the loader through
  int device = 0;
  int symtype = SWA_SYMTYPE_PROTEIN;
  int cus = 256;
  int64_t nseq = 0, nsym = 0, longest = 0, first_seqno = 0, total_seq = 0, total_sym = 0;
  // a translated shard (frames == 6) holds 6 virtual protein sequences per nucleotide sequence, virtual
  // index 6 * (seqno - first_seqno) + 3 * dstrand + dframe; nseq / nsym / h_offsets then describe the
  // virtual sequences, nt_* the nucleotide sequences they came from
  int frames = 1;
  // sequences left out by an OID mask / taxid list (swa_db_set_inclusion): not in any batch, score -1
  DevBuf<int32_t> excluded;
  int64_t n_excluded = 0, active_sym = 0;
  std::vector<int32_t> h_stage;            // host staging for score downloads
  std::vector<int64_t> h_ntlen;
  int64_t nt_sym = 0, nt_longest = 0;
  std::vector<int64_t> h_offsets;
  std::vector<int32_t> h_order;            // sequence indices by descending length
  hipStream_t stream = nullptr;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  hipStream_t stream2 = nullptr;           // second stream of a part-wise first pass (a shard that is still loading)
  hipEvent_t ev2[2] = {nullptr, nullptr};  // [0] stream -> stream2: inputs uploaded; [1] stream2 -> stream: its launches through

  DevBuf<uint8_t> residues;                // one byte per residue; nucleotide shards: two per byte (packed), low nibble first
  bool packed = false;
  DevBuf<int64_t> offsets;
  BatchSet main;                           // all sequences, two per DPP row (nucleotide shards: built on first use)
  BatchSet scratch;                        // re-queued sequences, one per DPP row
  BatchSet single;                         // all sequences, one per DPP row (built on first use)
  BatchSet single4;                        // the same at 4 bits per base: what the two-query kernel streams for nucleotide shards
  // windows of long sequences (ids nseq + v), see "windows" below
  DevBuf<int64_t> wstart;
  DevBuf<int32_t> wlen;
  std::vector<int64_t> h_wstart;
  std::vector<int32_t> h_wlen;
  BatchSet view;                           // the set a search runs over when long sequences are cut into windows
  const BatchSet* view_of = nullptr;       // ... built over this set
  int64_t view_O = -1, view_W = 0, view_Lmax = 0;
  DevBuf<int32_t> wparents, wfirst;        // fold: parent p of windows [wfirst[i], wfirst[i + 1])
  int nparents = 0;
  int64_t nwin = 0;
  int64_t len_of(int64_t id) const { return id < nseq ? h_offsets[size_t(id) + 1] - h_offsets[size_t(id)] : h_wlen[size_t(id - nseq)]; }
  swa_seqs seqs() const { return swa_seqs{residues.p, offsets.p, packed ? 1 : 0, int32_t(nseq), wstart.p, wlen.p}; }
  DevBuf<int32_t> scores;
  DevBuf<long long> scores64;
  DevBuf<long long> scores64b;              // 64-bit scores of the second query of a dual search
  DevBuf<int32_t> ovf_list;
  DevBuf<int32_t> ovf_list2;
  DevBuf<int32_t> scores2;                 // second query of a dual search
  DevBuf<uint8_t> qseq2;
  DevBuf<int32_t> rq_ids;                  // re-queue list of the wave-per-sequence path, and its pass hand-over
  DevBuf<int> rq_bh, rq_bf;
  DevBuf<int64_t> rq_boff;
  DevBuf<unsigned char> boundary;          // per-wave pass hand-over columns of the multi-pass kernel
  DevBuf<int32_t> ctl;                     // counters + candidate records of the hit filter (layout: "control block" below)
  int cand_cap = 0;                        // candidate records ctl has room for
  DevBuf<uint8_t> qseq;                    // query of the alignment-phase entry points
  DevBuf<uint8_t> qblock;                  // [swa_query | query 1 | query 2] of the search in flight, one upload
  const uint8_t* qseq_p = nullptr;         // -> query 1 / 2 inside qblock
  const uint8_t* qseq2_p = nullptr;
  DevBuf<int32_t> matrix;

  bool scoring_set = false;
  int32_t h_matrix[1024];
  int64_t goe = 0, ge = 0, hi = 0, lo = 0;
  Options opt;                             // swa_set_option
  // Queries for which the bound build sent back more than 2 % of the shard under the current scoring system (a family-rich
  // query: its threshold sits inside the bulk of its scores): THAT query takes the exact first pass at once from then on.
  // Keyed by the query itself (hash of its residues) so that one such query does not switch the build off for every later
  // query of its length - results never depend on this, but throughput must not depend on the order of a query file
  // either; the newest 64 are remembered.
  struct BoundOff { uint64_t qhash; int64_t qlen, minscore; };
  std::vector<BoundOff> bound_off;
  uint64_t cur_qhash = 0;                  // hash of the query (pair) of the search in flight
  static uint64_t hash_query(const uint8_t* q1, int64_t n1, const uint8_t* q2, int64_t n2)
  {
    uint64_t h = 1469598103934665603ull;
    for (int64_t i = 0; i < n1; ++i) h = (h ^ q1[i]) * 1099511628211ull;
    h = (h ^ 0xff) * 1099511628211ull;
    for (int64_t i = 0; q2 && i < n2; ++i) h = (h ^ q2[i]) * 1099511628211ull;
    return h;
  }
  bool bound_is_off(int64_t qlen, int64_t minscore) const
  {
    for (const BoundOff& b : bound_off) if (b.qhash == cur_qhash && b.qlen == qlen && minscore <= b.minscore) return true;
    return false;
  }
  void note_bound_off(int64_t qlen, int64_t minscore)
  {
    if (bound_off.size() >= 64) bound_off.erase(bound_off.begin());
    bound_off.push_back({cur_qhash, qlen, minscore});
  }
  // page-locked staging block for everything a search reads back (counters, tallies, the first candidates): one
  // asynchronous copy and ONE stream synchronisation per search
  unsigned char* pin = nullptr;
  size_t pin_bytes = 0;

  ~swa_db()
  {
    loading.reset();                       // stops and joins the loader before anything it uses goes away
    for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : ev2) if (e) (void)hipEventDestroy(e);
    if (stream2) (void)hipStreamDestroy(stream2);
    if (stream) (void)hipStreamDestroy(stream);
    if (pin) (void)hipHostFree(pin);
  }
  size_t hbm_bytes() const
  {
    return residues.bytes() + offsets.bytes() + main.slots.bytes() + main.batches.bytes() + main.stream.bytes() +
           scratch.slots.bytes() + scratch.batches.bytes() + scratch.stream.bytes() + single.slots.bytes() +
           single.batches.bytes() + single.stream.bytes() + single4.slots.bytes() + single4.batches.bytes() +
           single4.stream.bytes() + scores2.bytes() + boundary.bytes() + scores.bytes() +
           scores64.bytes() + scores64b.bytes() + ovf_list.bytes() + ovf_list2.bytes() + ctl.bytes() + view.slots.bytes() +
           view.batches.bytes() + view.stream.bytes() + wstart.bytes() + wlen.bytes() + wparents.bytes() + wfirst.bytes() +
           qblock.bytes() + matrix.bytes();
  }
};

namespace {
int streamed_set_scoring(swa_db* front, const int64_t* matrix, int64_t goe, int64_t ge);
int streamed_set_inclusion(swa_db* front, const uint8_t* include, int64_t n);
size_t streamed_hbm(const swa_db* front);
int streamed_wait(swa_db* front);
int streamed_progress(swa_db* front, int64_t* done, int64_t* total, int32_t* ready, int32_t* nparts);
int streamed_candidates(swa_db* front, const uint8_t* query, int64_t qlen, int64_t keep, int64_t minscore, int64_t maxscore,
                        std::vector<struct Cand>& cand, int64_t* tot, int64_t* obv, swa_counters_t* counters,
                        const uint8_t* query2 = nullptr, int32_t tag1 = 0, struct Pair* pair = nullptr, int32_t tag0 = 0);
int streamed_all_scores(swa_db* front, const uint8_t* query, int64_t qlen, int64_t* scores, swa_counters_t* counters,
                        const uint8_t* query2 = nullptr, int64_t* scores2 = nullptr);
int streamed_by_owner(swa_db* front, const int64_t* seqnos, int64_t n, const std::function<int(swa_db*, const std::vector<int64_t>&)>& fn);
int settle_loading(swa_db* db, bool wait, bool* still, bool explicit_wait = false);
void release_loader_leftovers(swa_db* db);
int apply_inclusion(swa_db* db, const uint8_t* include, int64_t n, bool release_now = true);
size_t loading_hbm(const swa_db* db);
// entry points that want ONE resident shard: not for streamed handles; a shard that is still loading is waited for
int loaded(swa_db* db)
{
  return db && db->loading ? settle_loading(db, true, nullptr) : SWA_OK;
}

// fn(lo, hi) over [0, n) on a few host threads (disjoint ranges; at least `grain` items per thread)
template <typename Fn> void parallel_for(int64_t n, int64_t grain, const Fn& fn, int64_t max_threads = 16)
{
  const int64_t nthreads = std::max<int64_t>(1, std::min<int64_t>({int64_t(std::thread::hardware_concurrency()), max_threads, n / std::max<int64_t>(1, grain)}));
  if (nthreads == 1) { fn(int64_t(0), n); return; }
  std::vector<std::thread> pool;
  for (int64_t t = 0; t < nthreads; ++t) pool.emplace_back([&, t]() { fn(n * t / nthreads, n * (t + 1) / nthreads); });
  for (std::thread& t : pool) t.join();
}

// Host half of a batch set: `ids` (ordered by descending length) laid out as batches with `per_row` sequences per DPP row
// (2 = packed pairs for the f16 kernel, 1 = slot A only for the wide kernels) - the tables the format kernel and the
// search kernels read.  Needs lengths only, no residues: the pipelined open makes its plans while the residues travel.
struct PlannedSet {
  std::vector<int32_t> slots;
  std::vector<swa_batch> batches;
  std::vector<int32_t> h_steps;
  uint64_t chunk_total = 0;
};
template <typename LenOf>
int plan_batches(const LenOf& len_of, const int32_t* ids, int64_t n, int per_row, PlannedSet& ps, int64_t max_threads = 16)
{
  const int per_batch = 4 * per_row;
  const int64_t nb = (n + per_batch - 1) / per_batch;
  if (nb > 0x7fffffff) return fail(SWA_EINVAL, "too many batches for one shard");
  ps.slots.assign(size_t(nb) * SWA_SLOTS, -1);
  ps.batches.resize(static_cast<size_t>(nb));
  ps.h_steps.resize(size_t(nb));
  parallel_for(nb, 1 << 15, [&](int64_t lo, int64_t hi) {
    for (int64_t b = lo; b < hi; ++b) {
      int64_t longest = 0;
      for (int j = 0; j < per_batch; ++j) {
        const int64_t i = b * per_batch + j;
        if (i >= n) break;
        const int32_t id = ids[i];
        ps.slots[size_t(b) * SWA_SLOTS + (j / per_row) * 2 + j % per_row] = id;
        longest = std::max<int64_t>(longest, len_of(id));
      }
      // even, and at least one full chunk; lengths beyond the 31-bit step count are caught below
      ps.h_steps[size_t(b)] = int32_t(std::min<int64_t>(0x7ffffffe, std::max<int64_t>(16, (longest + 1) & ~int64_t(1))));
    }
  }, max_threads);
  uint64_t chunk_total = 0;
  for (int64_t b = 0; b < nb; ++b) {
    const int64_t steps = ps.h_steps[size_t(b)];
    if (chunk_total > 0xffffffffull || steps > 0x7ffffff0) return fail(SWA_EINVAL, "residue stream exceeds 2^32 chunks");
    ps.batches[size_t(b)].offset = uint32_t(chunk_total);
    ps.batches[size_t(b)].steps = int32_t(steps);
    chunk_total += uint64_t((steps + 15) / 16);
  }
  ps.chunk_total = chunk_total;
  return SWA_OK;
}
// Device half: tables up, format kernel enqueued on `st`.  The copies read ps's vectors: the caller keeps them alive until
// the stream has passed this point.
int format_planned(swa_db* db, const PlannedSet& ps, BatchSet& bs, bool nibbles, hipStream_t st)
{
  const size_t nb = ps.batches.size();
  HIP_TRY(bs.slots.reserve(ps.slots.size()));
  HIP_TRY(bs.batches.reserve(nb));
  HIP_TRY(bs.stream.reserve(size_t(ps.chunk_total) * (nibbles ? 16 : 64)));      // 32 / 128 bytes per 16-column chunk
  if (nb) {
    const swa_seqs sq = db->seqs();
    HIP_TRY(hipMemcpyAsync(bs.slots.p, ps.slots.data(), ps.slots.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(bs.batches.p, ps.batches.data(), nb * sizeof(swa_batch), hipMemcpyHostToDevice, st));
    HIP_TRY(swa_launch_format(&sq, bs.slots.p, bs.batches.p, int(nb), bs.stream.p, nibbles ? 1 : 0, st));
  }
  bs.nibbles = nibbles;
  bs.nbatches = int(nb);
  bs.chunks = int64_t(ps.chunk_total);
  return SWA_OK;
}
// both halves, synchronously
int build_batches(swa_db* db, const int32_t* ids, int64_t n, int per_row, BatchSet& bs, bool nibbles = false)
{
  PlannedSet ps;
  int rc = plan_batches([&](int32_t id) { return db->len_of(id); }, ids, n, per_row, ps);
  if (rc == SWA_OK) rc = format_planned(db, ps, bs, nibbles, db->stream);
  if (rc != SWA_OK) return rc;
  HIP_TRY(hipStreamSynchronize(db->stream));       // ps goes out of scope
  bs.h_steps.swap(ps.h_steps);
  std::vector<int32_t>().swap(bs.h_slots);           // the set is `ids` cut into batches
  bs.built = true;
  return SWA_OK;
}

int persistent_blocks(const swa_db* db, int nbatches)
{
  int blocks = (nbatches + 3) / 4;                   // 4 waves per block, one batch per wave at a time
  const int cap = db->cus * 8;
  if (blocks > cap) blocks = cap;
  return blocks < 1 ? 1 : blocks;
}

// sort sequence indices by (length desc, index asc): counting sort when lengths are modest - on a few threads when the
// list is long (each counts and scatters its own contiguous slice of the list, so the order within a length is the list's)
template <typename LenOf>
void order_by_length(const LenOf& len_of, const int32_t* ids, int64_t n, std::vector<int32_t>& out, int64_t max_threads = 16)
{
  out.resize(size_t(n));
  int64_t longest = 0;
  {
    std::mutex mu;
    parallel_for(n, 1 << 18, [&](int64_t lo, int64_t hi) {
      int64_t lg = 0;
      for (int64_t i = lo; i < hi; ++i) lg = std::max<int64_t>(lg, len_of(ids ? ids[i] : int32_t(i)));
      std::lock_guard<std::mutex> g(mu);
      longest = std::max(longest, lg);
    }, max_threads);
  }
  if (longest <= (int64_t(1) << 24) && n > 1024) {
    const size_t bins = size_t(longest) + 1;
    // slices: few enough that their private histograms stay small next to the list
    const int64_t T = std::max<int64_t>(1, std::min<int64_t>({int64_t(std::thread::hardware_concurrency()), max_threads, n >> 18,
                                                               int64_t(8 * size_t(n) / bins) + 1}));
    std::vector<std::vector<int64_t>> count(static_cast<size_t>(T), std::vector<int64_t>(bins, 0));
    auto slice = [&](int64_t t, int64_t* lo, int64_t* hi) { *lo = n * t / T; *hi = n * (t + 1) / T; };
    auto tally = [&](int64_t t) {
      int64_t lo, hi;
      slice(t, &lo, &hi);
      std::vector<int64_t>& c = count[size_t(t)];
      for (int64_t i = lo; i < hi; ++i) ++c[size_t(longest - len_of(ids ? ids[i] : int32_t(i)))];
    };
    auto scatter = [&](int64_t t) {
      int64_t lo, hi;
      slice(t, &lo, &hi);
      std::vector<int64_t>& c = count[size_t(t)];
      for (int64_t i = lo; i < hi; ++i) {
        const int32_t id = ids ? ids[i] : int32_t(i);
        out[size_t(c[size_t(longest - len_of(id))]++)] = id;
      }
    };
    auto run = [&](const std::function<void(int64_t)>& fn) {
      if (T == 1) { fn(0); return; }
      std::vector<std::thread> pool;
      for (int64_t t = 0; t < T; ++t) pool.emplace_back(fn, t);
      for (std::thread& th : pool) th.join();
    };
    run(tally);
    int64_t at = 0;                                      // start of (bin, slice): bins in order, slices in order within a bin
    for (size_t k = 0; k < bins; ++k)
      for (int64_t t = 0; t < T; ++t) { const int64_t c = count[size_t(t)][k]; count[size_t(t)][k] = at; at += c; }
    run(scatter);
  } else {
    for (int64_t i = 0; i < n; ++i) out[size_t(i)] = ids ? ids[i] : int32_t(i);
    std::stable_sort(out.begin(), out.end(), [&](int32_t a, int32_t b) { return len_of(a) > len_of(b); });
  }
}

// OR over a byte range on a few host threads (residue-code validation of a shard: 3 GB in 0.1 s instead of 1.5)
uint8_t or_of_bytes(const uint8_t* data, int64_t n)
{
  const int64_t nthreads = std::max<int64_t>(1, std::min<int64_t>({int64_t(std::thread::hardware_concurrency()), 16, n >> 24}));
  std::vector<uint8_t> acc(size_t(nthreads), 0);
  auto scan = [&](int64_t t) {
    const uint8_t* p = data + n * t / nthreads;
    const uint8_t* e = data + n * (t + 1) / nthreads;
    uint64_t a8 = 0;
    uint8_t a = 0;
    for (; p < e && (reinterpret_cast<uintptr_t>(p) & 7); ++p) a |= *p;
    for (; p + 8 <= e; p += 8) { uint64_t v; std::memcpy(&v, p, 8); a8 |= v; }
    for (; p < e; ++p) a |= *p;
    for (int k = 0; k < 8; ++k) a |= uint8_t(a8 >> (8 * k));
    acc[size_t(t)] = a;
  };
  if (nthreads == 1) scan(0);
  else {
    std::vector<std::thread> pool;
    for (int64_t t = 0; t < nthreads; ++t) pool.emplace_back(scan, t);
    for (std::thread& t : pool) t.join();
  }
  uint8_t any = 0;
  for (uint8_t a : acc) any |= a;
  return any;
}

// Host tables and device buffers of a shard, from the sequence lengths alone (`offsets`: nseq + 1 prefix sums): everything
// an open does that needs no residue.  order: also the length order (the pipelined open computes it beside the transfer).
int ingest_tables(swa_db* db, const int64_t* offsets, int64_t nseq, bool order)
{
  if (nseq > 0x7ffffff0) return fail(SWA_EINVAL, "more than 2^31 sequences in one shard; shard the database");
  db->nseq = nseq;
  db->h_offsets.resize(size_t(nseq) + 1);
  const int64_t base = offsets[0];
  int64_t longest = 0;
  bool backwards = false;
  {
    std::mutex mu;
    parallel_for(nseq + 1, 1 << 18, [&](int64_t lo, int64_t hi) {
      int64_t lg = 0;
      bool bad = false;
      for (int64_t s = lo; s < hi; ++s) {
        db->h_offsets[size_t(s)] = offsets[s] - base;
        if (s < nseq) { const int64_t len = offsets[s + 1] - offsets[s]; bad |= len < 0; lg = std::max(lg, len); }
      }
      std::lock_guard<std::mutex> g(mu);
      longest = std::max(longest, lg);
      backwards |= bad;
    });
  }
  if (backwards) return fail(SWA_EINVAL, "sequence offsets must be non-decreasing");
  db->nsym = db->h_offsets[size_t(nseq)];
  db->active_sym = db->nsym;
  db->longest = longest;
  HIP_TRY(hipSetDevice(db->device));
  if (!db->stream) {
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, db->device));
    db->cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    HIP_TRY(hipStreamCreate(&db->stream));
    for (hipEvent_t& e : db->ev) HIP_TRY(hipEventCreate(&e));
  }
  if (!db->stream2) {
    HIP_TRY(hipStreamCreateWithFlags(&db->stream2, hipStreamNonBlocking));
    for (hipEvent_t& e : db->ev2) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  HIP_TRY(db->offsets.reserve(size_t(nseq) + 1));
  HIP_TRY(hipMemcpyAsync(db->offsets.p, db->h_offsets.data(), (size_t(nseq) + 1) * sizeof(int64_t), hipMemcpyHostToDevice, db->stream));
  HIP_TRY(db->scores.reserve(size_t(nseq)));
  HIP_TRY(db->ovf_list.reserve(size_t(nseq)));
  db->cand_cap = int(std::max<int64_t>(1, std::min<int64_t>(nseq, 1 << 20)));
  HIP_TRY(db->ctl.reserve(48 + size_t(db->cand_cap) * (sizeof(swa_cand) / sizeof(int32_t))));
  HIP_TRY(db->matrix.reserve(1024));
  if (order) order_by_length([&](int32_t id) { return db->len_of(id); }, nullptr, nseq, db->h_order);
  db->main.built = db->single.built = db->single4.built = db->view.built = false;
  db->view_of = nullptr;
  db->nwin = 0;
  return SWA_OK;
}

// residues == nullptr: db->residues already holds them on the device (translated shards)
int ingest(swa_db* db, const uint8_t* residues, const int64_t* offsets, int64_t nseq)
{
  int rc = ingest_tables(db, offsets, nseq, true);
  if (rc != SWA_OK) return rc;
  const int64_t base = offsets[0];
  // Residue codes index the LDS profile (code x row stride) and the 32 x 32 matrix: a code outside the alphabet - a
  // corrupt .psq, a caller's array - would read out of bounds and score silently wrong.  One OR over all bytes.
  if (residues && db->nsym) {
    const uint8_t bad_bits = db->symtype == SWA_SYMTYPE_NUCLEOTIDE ? 0xF0 : 0xE0;
    const uint8_t any = or_of_bytes(residues + base, db->nsym);
    if (any & bad_bits)
      return fail(SWA_EINVAL, db->symtype == SWA_SYMTYPE_NUCLEOTIDE ? "database residue code out of range (nucleotide codes are 4-bit masks, < 16)"
                                                                   : "database residue code out of range (must be < 32)");
  }
  std::vector<uint8_t> packed_host;
  if (residues && db->symtype == SWA_SYMTYPE_NUCLEOTIDE) {
    // nucleotide codes are 4-bit masks (validated above): the shard keeps two per byte - residue i in byte i >> 1, low
    // nibble first, whatever sequence it belongs to - and every kernel reads them through seq_residue (sw_common.cuh)
    db->packed = true;
    const int64_t nbytes = (db->nsym + 1) / 2;
    packed_host.assign(size_t(nbytes) + 16, 0);
    const uint8_t* src = residues + base;
    parallel_for(nbytes, 1 << 22, [&](int64_t lo, int64_t hi) {
      for (int64_t b = lo; b < hi; ++b) {
        const int64_t i = 2 * b;
        packed_host[size_t(b)] = uint8_t((src[i] & 15) | (i + 1 < db->nsym ? (src[i + 1] & 15) << 4 : 0));
      }
    });
    HIP_TRY(db->residues.reserve(size_t(nbytes) + 16));
    if (nbytes) HIP_TRY(hipMemcpyAsync(db->residues.p, packed_host.data(), size_t(nbytes), hipMemcpyHostToDevice, db->stream));
  } else if (residues) {
    HIP_TRY(db->residues.reserve(size_t(db->nsym) + 16));
    if (db->nsym) HIP_TRY(hipMemcpyAsync(db->residues.p, residues + base, size_t(db->nsym), hipMemcpyHostToDevice, db->stream));
  }
  HIP_TRY(hipStreamSynchronize(db->stream));           // packed_host goes out of scope
  // a nucleotide shard is searched with both strands in one pass over the one-sequence-per-row stream (single4): its
  // pair stream is only built if a single-strand search or a short query asks for it
  if (db->packed) return SWA_OK;
  return build_batches(db, db->h_order.data(), nseq, 2, db->main);
}

uint32_t f16_pair(float v)
{
  const uint32_t b = f16_bits(v);
  return b | b << 16;
}

// all sequences, one per DPP row (dual-query kernel, and the 32-bit kernel when f16 does not apply)
int ensure_single(swa_db* db)
{
  const int lrc = settle_loading(db, true, nullptr);     // h_order is the loader's to fill: an empty set must never be marked built
  if (lrc != SWA_OK) return lrc;
  if (db->single.built) return SWA_OK;
  return build_batches(db, db->h_order.data(), int64_t(db->h_order.size()), 1, db->single);
}
// the same at 4 bits per base (nucleotide shards, two-query kernel with 16-lane chains)
int ensure_single4(swa_db* db)
{
  const int lrc = settle_loading(db, true, nullptr);
  if (lrc != SWA_OK) return lrc;
  if (db->single4.built) return SWA_OK;
  return build_batches(db, db->h_order.data(), int64_t(db->h_order.size()), 1, db->single4, true);
}
// all sequences, two per DPP row
int ensure_main(swa_db* db)
{
  const int lrc = settle_loading(db, true, nullptr);
  if (lrc != SWA_OK) return lrc;
  if (db->main.built) return SWA_OK;
  return build_batches(db, db->h_order.data(), int64_t(db->h_order.size()), 2, db->main);
}
}  // namespace
static int open_resident(const char* basename, int symtype, int device, int64_t first_seqno, int64_t last_seqno, swa_db** out);
#include "sw_loading.inc"
namespace {
// ---- windows: long database sequences, score-exact ------------------------------------------------------------------
// A sequence is worked on by ONE chain of at most 16 lanes, a column per ~1.5 us whatever else the shard holds: a
// 35 000-residue protein takes 50 ms, a chromosome minutes (the reference streams any length 4 columns at a time,
// search7.cc:836-847).  Cut it into windows that overlap by the longest stretch of the sequence a positive-scoring
// alignment can cover,
//     O = qlen + qlen x hi / R + 1     (at most qlen aligned columns; every further column sits in a gap and costs at
//                                        least R of a total that cannot exceed qlen x hi),
// window k = residues [k W, k W + W + O): an alignment starting in [k W, (k + 1) W) lies inside window k whole, a
// window's best local score never exceeds the sequence's, so the MAXIMUM OVER THE WINDOWS IS THE SCORE - exactly.
// Windows are searched as sequences of their own (ids nseq + v, swa_seqs.wstart / wlen) by the same kernels, re-queued
// one by one, and folded into their parents (swa_fold_windows) before anything reads the scores.
// Which sequences: longer than Lmax = max(1.5 (W + O), half the columns one chain gets if the shard is spread evenly
// over 64 chains per CU) - below that a sequence cannot outlast the rest of the shard.  W + O = max(1.25 O, 4096).
// The set a search launches over becomes a VIEW: tables of its own = [windows + the few sequences sharing the first
// batches | the set's remaining batches, offsets rebased], two stream regions under one base pointer; nothing is copied
// but 12 bytes per batch.  Rebuilt when the query changes O; shards without long sequences never build one.
struct WindowPlan { bool on = false; int64_t O = 0, W = 0, Lmax = 0; };
// slots_per_cu = database sequences one CU works on at a time under the kernel that will run: 4 SIMDs x resident waves x
// chains per wave x sequences per chain.  Round 2 assumed 64 (16-lane chains of sequence pairs, two waves) for every
// kernel; the one-lane kernels of short queries hold 3 072 - with them a shard is spread 48 times thinner, and a
// 35 000-residue protein (5 ms on one lane, alone) outlasts a 10-residue search of ten million sequences (3.7 ms).
WindowPlan plan_windows(const swa_db* db, int64_t qlen, int slots_per_cu)
{
  WindowPlan w;
  if (db->opt.window == 0 || db->ge <= 0 || db->h_order.empty()) return w;
  w.O = db->hi > 0 ? qlen + qlen * db->hi / db->ge + 1 : qlen + 1;
  // columns one slot works through if the shard is spread evenly; half of that is what a sequence may have without outlasting
  // the rest, and windows are cut about that long (at least 512 columns, at most 4 096, never less than 1.25 overlaps)
  const int64_t avg = db->active_sym / std::max<int64_t>(1, int64_t(db->cus) * std::max(1, slots_per_cu));
  const int64_t wlen = std::max<int64_t>(w.O + w.O / 4, std::min<int64_t>(4096, std::max<int64_t>(512, avg / 2)));
  w.W = db->opt.window_step > 0 ? db->opt.window_step : wlen - w.O;
  // ... and nothing shorter than four mean lengths is cut: on a small shard avg is tiny, and windowing the ordinary tail of the
  // length distribution would recompute overlaps for nothing
  const int64_t mean4 = 4 * db->active_sym / std::max<int64_t>(1, int64_t(db->h_order.size()));
  w.Lmax = db->opt.window > 0 ? db->opt.window : std::max<int64_t>({3 * (w.W + w.O) / 2, avg / 2, mean4});
  if (w.W + w.O >= (int64_t(1) << 30)) return w;         // window lengths are 32 bit
  w.on = db->len_of(db->h_order[0]) > w.Lmax;            // h_order: longest first
  return w;
}
// resident waves per SIMD of a build with K rows per lane (launch bounds of the kernels: split_waves_for / cb_waves_for)
int waves_for_rows(int K, bool bound)
{
  return bound ? (K <= 5 ? 8 : K <= 10 ? 6 : K <= 20 ? 4 : K <= 29 ? 3 : 2) : (K <= 8 ? 8 : K <= 12 ? 6 : K <= 20 ? 4 : K <= 31 ? 3 : 2);
}
int slots_per_cu(int G, int K, bool bound, int seqs_per_chain)
{
  if (G <= 0 || K <= 0) return 64;
  return 4 * waves_for_rows(K, bound) * (64 / G) * seqs_per_chain;
}

// the set to launch over: `set` itself, or the view that replaces its long sequences by windows
int prepare_view(swa_db* db, const BatchSet& set, int per_row, int64_t qlen, const BatchSet** out, int nslots = 64)
{
  *out = &set;
  const WindowPlan wp = plan_windows(db, qlen, nslots);
  if (!wp.on) { db->nwin = 0; db->view_of = nullptr; return SWA_OK; }
  if (db->view_of == &set && db->view.built && db->view_O == wp.O && db->view_W == wp.W && db->view_Lmax == wp.Lmax) {
    *out = &db->view;
    return SWA_OK;
  }
  hipStream_t st = db->stream;
  BatchSet& v = db->view;
  v.built = false;
  db->view_of = nullptr;
  const int64_t n = int64_t(db->h_order.size());
  const int per_batch = 4 * per_row;
  // The set's sequences in batch order: the shard's length order cut into batches, or - a streamed-in shard's main set -
  // its own slot table.  Batches are by descending length either way, so the long sequences sit in a prefix of them.
  auto id_at = [&](int64_t b, int j) -> int32_t {
    if (!set.h_slots.empty()) return set.h_slots[size_t(b) * SWA_SLOTS + size_t((j / per_row) * 2 + j % per_row)];
    const int64_t i = b * per_batch + j;
    return i < n ? db->h_order[size_t(i)] : -1;
  };
  std::vector<int32_t> longs, rest;
  int64_t nskip = 0;
  for (; nskip < set.nbatches; ++nskip) {
    bool any = false;
    for (int j = 0; j < per_batch; ++j) { const int32_t id = id_at(nskip, j); any |= id >= 0 && db->len_of(id) > wp.Lmax; }
    // a merged table (streamed-in shard) is ordered by batch STEPS with ties broken by part: a batch without a long member may
    // come before one with - stop only where no later batch can hold one (steps = longest member rounded up to even)
    if (!any && (set.h_slots.empty() || set.h_steps[size_t(nskip)] <= wp.Lmax)) break;
    for (int j = 0; j < per_batch; ++j) {
      const int32_t id = id_at(nskip, j);
      if (id >= 0) (db->len_of(id) > wp.Lmax ? longs : rest).push_back(id);
    }
  }
  const int64_t nlong = int64_t(longs.size());
  // windows of the long sequences, then the sequences that shared their batches
  db->h_wstart.clear();
  db->h_wlen.clear();
  std::vector<int32_t> parents, wfirst, entries;
  for (int64_t i = 0; i < nlong; ++i) {
    const int32_t p = longs[size_t(i)];
    const int64_t o = db->h_offsets[size_t(p)], plen = db->h_offsets[size_t(p) + 1] - o;
    parents.push_back(p);
    wfirst.push_back(int32_t(db->h_wstart.size()));
    for (int64_t start = 0; start == 0 || start + wp.O < plen; start += wp.W) {
      db->h_wstart.push_back(o + start);
      db->h_wlen.push_back(int32_t(std::min<int64_t>(wp.W + wp.O, plen - start)));
    }
    if (db->h_wstart.size() > size_t(0x3fffffff)) return fail(SWA_EINVAL, "too many windows; raise window_step");
  }
  wfirst.push_back(int32_t(db->h_wstart.size()));
  db->nwin = int64_t(db->h_wstart.size());
  if (db->nseq + db->nwin > 0x7ffffff0) return fail(SWA_EINVAL, "too many windows for one shard");
  for (int64_t v2 = 0; v2 < db->nwin; ++v2) entries.push_back(int32_t(db->nseq + v2));
  entries.insert(entries.end(), rest.begin(), rest.end());
  std::vector<int32_t> ordered;
  order_by_length([&](int32_t id) { return db->len_of(id); }, entries.data(), int64_t(entries.size()), ordered);
  // everything indexed by id grows by the windows (contents between searches do not matter)
  const size_t total_ids = size_t(db->nseq + db->nwin);
  HIP_TRY(hipStreamSynchronize(st));
  HIP_TRY(db->scores.reserve(total_ids));
  HIP_TRY(db->ovf_list.reserve(total_ids));
  if (db->scores2.cap) HIP_TRY(db->scores2.reserve(total_ids));
  if (db->ovf_list2.cap) HIP_TRY(db->ovf_list2.reserve(total_ids));
  if (db->scores64.cap) HIP_TRY(db->scores64.reserve(total_ids));
  if (db->scores64b.cap) HIP_TRY(db->scores64b.reserve(total_ids));
  HIP_TRY(db->wstart.reserve(size_t(db->nwin)));
  HIP_TRY(db->wlen.reserve(size_t(db->nwin)));
  HIP_TRY(db->wparents.reserve(parents.size()));
  HIP_TRY(db->wfirst.reserve(wfirst.size()));
  HIP_TRY(hipMemcpyAsync(db->wstart.p, db->h_wstart.data(), db->h_wstart.size() * sizeof(int64_t), hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(db->wlen.p, db->h_wlen.data(), db->h_wlen.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(db->wparents.p, parents.data(), parents.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(db->wfirst.p, wfirst.data(), wfirst.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
  db->nparents = int(parents.size());
  // the long part: batches with chunk offsets counted from 0
  const int64_t ne = int64_t(ordered.size());
  const int64_t nl = (ne + per_batch - 1) / per_batch;
  const int64_t nm = int64_t(set.nbatches) - nskip;
  std::vector<int32_t> slots(size_t(nl) * SWA_SLOTS, -1);
  std::vector<swa_batch> batches(static_cast<size_t>(nl));
  uint64_t lchunks = 0;
  v.h_steps.clear();
  for (int64_t b = 0; b < nl; ++b) {
    int64_t longest = 0;
    for (int j = 0; j < per_batch; ++j) {
      const int64_t i = b * per_batch + j;
      if (i >= ne) break;
      const int32_t id = ordered[size_t(i)];
      slots[size_t(b) * SWA_SLOTS + (j / per_row) * 2 + j % per_row] = id;
      longest = std::max(longest, db->len_of(id));
    }
    const int64_t steps = std::max<int64_t>(16, (longest + 1) & ~int64_t(1));
    if (lchunks > 0xffffffffull || steps > 0x7ffffff0) return fail(SWA_EINVAL, "window stream exceeds 2^32 chunks");
    batches[size_t(b)].offset = uint32_t(lchunks);
    batches[size_t(b)].steps = int32_t(steps);
    v.h_steps.push_back(int32_t(steps));
    lchunks += uint64_t((steps + 15) / 16);
  }
  const size_t unit = set.nibbles ? 32 : 128;            // bytes per 16-column chunk
  HIP_TRY(v.stream.reserve(size_t(lchunks) * unit / 2 + 64));
  const uintptr_t pa = reinterpret_cast<uintptr_t>(set.stream.p), pb = reinterpret_cast<uintptr_t>(v.stream.p);
  const uintptr_t base = std::min(pa, pb) & ~uintptr_t(127);
  const uint64_t da = (pa - base) / unit, dl = (pb - base) / unit;
  if ((pa - base) % unit || (pb - base) % unit || da + uint64_t(set.chunks) > 0xffffffffull || dl + lchunks > 0xffffffffull) {
    // the two allocations lie more than 2^32 chunks apart (128 GiB for the 32-byte chunks of a nucleotide shard - hipMalloc
    // promises nothing about distances): search without windows, which is always correct, only slower for the long sequences
    db->nwin = 0;
    db->nparents = 0;
    db->view_of = nullptr;
    return SWA_OK;                                         // *out still names the set itself
  }
  for (swa_batch& b : batches) b.offset += uint32_t(dl);
  HIP_TRY(v.slots.reserve(size_t(nl + nm) * SWA_SLOTS));
  HIP_TRY(v.batches.reserve(size_t(nl + nm)));
  HIP_TRY(hipMemcpyAsync(v.slots.p, slots.data(), slots.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(v.batches.p, batches.data(), batches.size() * sizeof(swa_batch), hipMemcpyHostToDevice, st));
  if (nm > 0) {
    HIP_TRY(hipMemcpyAsync(v.slots.p + size_t(nl) * SWA_SLOTS, set.slots.p + size_t(nskip) * SWA_SLOTS,
                           size_t(nm) * SWA_SLOTS * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    HIP_TRY(swa_launch_rebase(set.batches.p + nskip, v.batches.p + nl, int(nm), uint32_t(da), st));
  }
  v.stream_base = reinterpret_cast<const uint16_t*>(base);
  v.nibbles = set.nibbles;
  {
    const swa_seqs sq = db->seqs();
    HIP_TRY(swa_launch_format(&sq, v.slots.p, v.batches.p, int(nl), reinterpret_cast<void*>(base), set.nibbles ? 1 : 0, st));
  }
  HIP_TRY(hipStreamSynchronize(st));                     // host vectors go out of scope
  int64_t skipped_chunks = 0;
  for (int64_t b = 0; b < nskip; ++b) skipped_chunks += (set.h_steps[size_t(b)] + 15) / 16;
  v.h_steps.insert(v.h_steps.end(), set.h_steps.begin() + nskip, set.h_steps.end());
  v.nbatches = int(nl + nm);
  v.nlong = int(nl);
  v.off_long = int64_t(dl);
  v.off_main = int64_t(da) + skipped_chunks;
  v.chunks = int64_t(lchunks) + set.chunks - skipped_chunks;
  v.built = true;
  db->view_of = &set;
  db->view_O = wp.O; db->view_W = wp.W; db->view_Lmax = wp.Lmax;
  *out = &v;
  return SWA_OK;
}

// windows -> parents, on whatever score arrays the search filled (after every re-queue, before anything reads them)
int fold_windows(swa_db* db, bool two, hipStream_t st)
{
  if (!db->nwin || !db->nparents) return SWA_OK;
  HIP_TRY(swa_launch_fold(db->scores.p, db->scores64.p, db->wparents.p, db->wfirst.p, db->nparents, int(db->nseq), st));
  if (two) HIP_TRY(swa_launch_fold(db->scores2.p, db->scores64b.p, db->wparents.p, db->wfirst.p, db->nparents, int(db->nseq), st));
  return SWA_OK;
}

// The 4-bit stream pads with code 0, which must not score: true of every matrix swa_matrix_nucleotide makes
// (matrices.cc:531-538 leaves row 0 at -1); a caller's own matrix with a positive entry there takes the 16-bit stream
bool nibble_stream_ok(const swa_db* db)
{
  if (!db->packed) return false;
  for (int q = 0; q < 32; ++q) if (db->h_matrix[q] > 0) return false;
  return true;
}

struct MpRun {
  int mode = 0;                 // 0 f16 pair, 1 f16 dual, 2 int32, 3 int64
  const BatchSet* set = nullptr;
  const uint8_t* q1 = nullptr;  // device pointers
  const uint8_t* q2 = nullptr;
  int32_t* scores = nullptr;
  int32_t* scores2 = nullptr;
  int32_t* ovf_count = nullptr;
  int32_t* ovf_list = nullptr;
  int32_t* ovf_count2 = nullptr;
  int32_t* ovf_list2 = nullptr;
  long long* scores64 = nullptr;  // 64-bit results (mode 3); null = db->scores64
  int64_t qlen_a = 0, qlen_b = 0; // mode 1 with two different queries: their own row counts (0 = qlen)
};

int mp_rows_for(int mode, int64_t qlen)
{
  // measured on MI355X (tools/gpu_nt_knobs.py): 16 rows per lane at 4 waves/SIMD beats 24 or 32 rows at
  // 3 or 2 waves even though it needs more passes.  For the f16 policies the passes are then made as even as
  // the query allows: npass = ceil(qlen / 256), K = ceil(qlen / (16 npass)) in 9..16, so that at most
  // 16 npass - 1 padding rows are computed (769 rows: 4 passes of 13 rows per lane instead of 16)
  if (mode == 0 || mode == 1) {
    if (mode == 1 && qlen <= 128) return 8;
    const int64_t npass = (qlen + 255) / 256;
    const int64_t k = (qlen + 16 * npass - 1) / (16 * npass);
    return int(std::max<int64_t>(9, std::min<int64_t>(16, k)));
  }
  if (mode == 2) return qlen <= 128 ? 8 : 16;
  return 8;
}

// limit below which the f16 forms are exact for K rows per lane
int64_t f16_limit(const swa_db* db, int K) { return 2048 - db->hi - int64_t(K + 1) * db->ge; }

int launch_mp_run(swa_db* db, const MpRun& r, int64_t qlen, hipStream_t st)
{
  int K = mp_rows_for(r.mode, qlen);
  if (db->opt.mp_k > 0 && r.mode <= 1) K = int(db->opt.mp_k);
  swa_mp_params p{};
  p.qseq = r.q1;
  p.qseq2 = r.q2;
  p.matrix = db->matrix.p;
  p.qlen = int32_t(qlen);
  p.qlen_a = int32_t(r.qlen_a);
  p.qlen_b = int32_t(r.qlen_b);
  p.rows_per_lane = K;
  p.npass = int32_t((qlen + 16 * K - 1) / (16 * K));
  p.stream = r.set->sp();
  p.batches = r.set->batches.p;
  p.slots = r.set->slots.p;
  p.nbatches = r.set->nbatches;
  p.counter = db->ctl.p + 0;
  p.scores = r.scores;
  p.scores2 = r.scores2;
  p.scores64 = r.scores64 ? r.scores64 : db->scores64.p;
  p.limit = r.mode <= 1 ? f16_limit(db, K) : (1ll << 31) - db->hi - int64_t(K + 1) * db->ge;
  p.ovf_count = r.ovf_count;
  p.ovf_list = r.ovf_list;
  p.ovf_count2 = r.ovf_count2;
  p.ovf_list2 = r.ovf_list2;
  p.gapopenextend = db->goe;
  p.gapextend = db->ge;
  p.gapextend_f = float(db->ge);
  p.negQR = f16_pair(-float(db->goe - db->ge));
  p.negR = f16_pair(-float(db->ge));
  p.negKR = f16_pair(-float(int64_t(K) * db->ge));
  for (int i = 0; i <= K + 1; ++i) p.rowc[i] = f16_pair(float(int64_t(i) * db->ge));

  const int rpu = r.mode == 0 ? 8 : 4;
  const size_t lds = size_t(32) * ((K + rpu - 1) / rpu) * 256;
  // resident waves per SIMD the kernel's VGPR count allows (-Rpass-analysis=kernel-resource-usage)
  int wps = swa_mp_waves(r.mode, K);
  if (db->opt.mp_w > 0) { p.tune_w = int(db->opt.mp_w); if (r.mode == 1 && K == 32) wps = p.tune_w; }
  const int waves_cu = 4 * wps;
  const int by_lds = std::max(1, int(160 * 1024 / lds));
  int nw = 4;
  while (nw < 8 && (waves_cu + nw - 1) / nw > by_lds) nw += 2;     // fewer, larger blocks when LDS is the limit
  const int threads = nw * 64;
  const int per_cu = std::max(1, std::min(by_lds, waves_cu / nw));
  // One launch over batches [lo, hi) of the set, whose steps are non-increasing there.  With several passes every resident
  // wave owns a hand-over buffer of (columns of its batch + 48) x 4 rows x (H, F), sized by the range's longest batch; the
  // total is bounded, so batches too long for the common allotment run first, on as few waves as their size allows.
  auto launch_range = [&](int lo, int hi) -> int {
    swa_mp_params q = p;
    q.batches = r.set->batches.p + lo;
    q.slots = r.set->slots.p + size_t(lo) * SWA_SLOTS;
    q.nbatches = hi - lo;
    if (q.nbatches <= 0) return SWA_OK;
    const int supers = (q.nbatches + nw - 1) / nw;
    const int blocks = std::max(1, std::min(supers, db->cus * per_cu));
    if (q.npass > 1) {
      const size_t vbytes = r.mode == 3 ? 8 : 4;
      const size_t per_col = 8 * vbytes;
      size_t free_b = 0, total_b = 0;
      HIP_TRY(hipMemGetInfo(&free_b, &total_b));
      size_t budget = std::min<size_t>(size_t(4) << 30, (free_b + db->boundary.bytes()) / 4);
      if (db->opt.boundary_mb >= 0) budget = size_t(db->opt.boundary_mb) << 20;     // tests
      const std::vector<int32_t>& hs = r.set->h_steps;
      const int64_t need = int64_t(hs.empty() ? db->longest : hs[size_t(lo)]) + 48;
      const int64_t cols_all = int64_t(budget / (size_t(blocks) * nw * per_col));
      if (need <= cols_all) {
        q.boundary_cols = int32_t(need);
        HIP_TRY(db->boundary.reserve(size_t(blocks) * nw * size_t(need) * per_col));
        q.boundary = db->boundary.p;
      } else {
        // batches [lo, lo + i0) are too long for the common allotment
        const int64_t fit = std::max<int64_t>(cols_all, 64);
        const int i0 = int(std::partition_point(hs.begin() + lo, hs.begin() + hi, [&](int32_t st) { return int64_t(st) + 48 > fit; }) - (hs.begin() + lo));
        const int supers_a = (i0 + nw - 1) / nw;
        int blocks_a = int(std::max<int64_t>(1, std::min<int64_t>(supers_a, int64_t(budget / (size_t(nw) * size_t(need) * per_col)))));
        const size_t bytes_a = size_t(blocks_a) * nw * size_t(need) * per_col;
        const size_t bytes_b = size_t(blocks) * nw * size_t(fit) * per_col;
        if (bytes_a > free_b + db->boundary.bytes())
          return fail(SWA_ENOMEM, "a database sequence is too long for the multi-pass kernel's hand-over buffer; use a shorter query");
        HIP_TRY(db->boundary.reserve(std::max(bytes_a, bytes_b)));
        q.boundary = db->boundary.p;
        swa_mp_params qa = q;
        qa.nbatches = i0;
        qa.boundary_cols = int32_t(need);
        HIP_TRY(hipMemsetAsync(db->ctl.p + 0, 0, sizeof(int32_t), st));
        HIP_TRY(swa_launch_mp(r.mode, K, &qa, blocks_a, threads, st));
        q.batches += i0;
        q.slots += size_t(i0) * SWA_SLOTS;
        q.nbatches -= i0;
        q.boundary_cols = int32_t(fit);
        if (q.nbatches <= 0) return SWA_OK;
      }
    }
    HIP_TRY(hipMemsetAsync(db->ctl.p + 0, 0, sizeof(int32_t), st));
    HIP_TRY(swa_launch_mp(r.mode, K, &q, blocks, threads, st));
    return SWA_OK;
  };
  // A view (windows of long sequences, prepare_view) is TWO length-sorted regions - [windows and what shared their batches |
  // the set's remaining batches] - and the second may hold longer batches than the first (windows are at most W + O long,
  // unwindowed sequences up to Lmax): sized by h_steps[0] alone, the hand-over of a multi-pass launch would be overrun.
  // Each region gets a launch of its own, sized by its own longest batch.
  const int nb_all = r.set->nbatches;
  const int split = p.npass > 1 && r.set->stream_base && r.set->nlong > 0 && r.set->nlong < nb_all ? r.set->nlong : nb_all;
  int rc = launch_range(0, split);
  if (rc == SWA_OK && split < nb_all) rc = launch_range(split, nb_all);
  return rc;
}

// Runs of batches whose pass hand-over (8 bytes per stream element) fits the buffer budget; reserves db->boundary.
struct PassRuns {
  std::vector<int> cut{0};                             // run i = batches [cut[i], cut[i+1])
  std::vector<int64_t> first_chunk{0};                 // stream chunk run i starts at
};
int plan_pass_runs(swa_db* db, const BatchSet& bs, PassRuns& runs)
{
  size_t free_b = 0, total_b = 0;
  HIP_TRY(hipMemGetInfo(&free_b, &total_b));
  const size_t avail = free_b + db->boundary.bytes();
  size_t budget = std::min<size_t>(size_t(64) << 30, avail / 4);
  if (db->opt.boundary_mb >= 0) budget = size_t(db->opt.boundary_mb) << 20;     // tests
  const size_t per_chunk = 64 * 8;
  const int nb = bs.nbatches;
  size_t largest = 0, bytes = 0;
  // chunk offsets count from the set's base pointer; a view has two regions (windows | the set's own batches)
  int64_t chunk = bs.stream_base ? (bs.nlong ? bs.off_long : bs.off_main) : 0;
  runs.first_chunk[0] = chunk;
  for (int b = 0; b < nb; ++b) {
    const size_t need = size_t((bs.h_steps[size_t(b)] + 15) / 16) * per_chunk;
    if (bs.stream_base && b == bs.nlong && b > 0) {        // a run never straddles the two regions
      runs.cut.push_back(b);
      runs.first_chunk.push_back(bs.off_main);
      chunk = bs.off_main;
      bytes = 0;
    }
    if (bytes && bytes + need > budget) {
      runs.cut.push_back(b);
      runs.first_chunk.push_back(chunk);
      bytes = 0;
    }
    bytes += need;
    chunk += (bs.h_steps[size_t(b)] + 15) / 16;
    largest = std::max(largest, bytes);
  }
  runs.cut.push_back(nb);
  if (largest > avail)
    return fail(SWA_ENOMEM, "a database sequence is too long for the multi-pass hand-over buffer; use a shorter query");
  HIP_TRY(db->boundary.reserve(largest));
  return SWA_OK;
}

// Queries longer than one pass of the tuned kernel (928 rows): passes of 16 x K rows, K <= 56, made as even as the
// query allows, each pass ONE launch of the MP build of swa_narrow_split_kernel over a run of batches.  The last
// row of a pass is handed to the next launch through db->boundary, 8 bytes per element of the residue stream
// (13 GB for a 10 M-sequence protein database: HBM is the one thing this box has to spare, and at 26 GB of extra
// traffic per pass boundary it costs 3 ms of a 130 ms pass).  The buffer is capped (64 GB or a quarter of the free memory); batches are taken in runs
// that fit it, all passes of a run before the next run.
void split_pass_shape(int64_t qlen, int* npass, int* K, int kmax = 56)   // 57+ rows of a pass build spill
{
  const int64_t np = (qlen + 16 * kmax - 1) / (16 * kmax);
  *npass = int(np);
  *K = int(std::max<int64_t>(30, (qlen + 16 * np - 1) / (16 * np)));
}

int split_pass_rows(int64_t qlen)
{
  int npass = 0, K = 0;
  split_pass_shape(qlen, &npass, &K);
  return K;
}

// bound: the passes are bound builds (top-K searches, see run_search)
int launch_split_passes(swa_db* db, const BatchSet& bs, int64_t qlen, hipStream_t st, bool bound, int64_t bound_min)
{
  int npass = 0, K = 0;
  split_pass_shape(qlen, &npass, &K);
  const int Nb = bound ? swa_bound_period() : 0;
  swa_narrow_params p{};
  p.query = reinterpret_cast<const swa_query*>(db->qblock.p);
  p.stream = bs.sp();
  p.counter = db->ctl.p + 0;
  p.scores = db->scores.p;
  p.ovf_count = db->ctl.p + 1;
  p.ovf_list = db->ovf_list.p;
  p.negQ = f16_pair(-float(db->goe));
  p.negR = f16_pair(-float(db->ge));
  p.shifted = 1;
  p.limit = int32_t(bound ? std::min<int64_t>(f16_limit(db, K + Nb), bound_min) : f16_limit(db, K));
  p.gapextend_f = float(db->ge);
  p.negQR = f16_pair(-float(db->goe - db->ge));
  p.negKR = f16_pair(-float(int64_t(K) * db->ge));
  for (int r = 0; r <= K + Nb + 1; ++r) p.rowc[r] = f16_pair(float(int64_t(r) * db->ge));

  PassRuns runs;
  const int prc = plan_pass_runs(db, bs, runs);
  if (prc != SWA_OK) return prc;
  const std::vector<int>& cut = runs.cut;
  const std::vector<int64_t>& first_chunk = runs.first_chunk;
  p.boundary = db->boundary.p;
  for (size_t i = 0; i + 1 < cut.size(); ++i) {
    p.batches = bs.batches.p + cut[i];
    p.slots = bs.slots.p + size_t(cut[i]) * SWA_SLOTS;
    p.nbatches = cut[i + 1] - cut[i];
    p.boundary_base = first_chunk[i];
    for (int pass = 0; pass < npass; ++pass) {
      p.row0 = pass * 16 * K;
      p.pass = pass;
      p.last = pass + 1 == npass;
      HIP_TRY(hipMemsetAsync(db->ctl.p + 0, 0, sizeof(int32_t), st));
      HIP_TRY(bound ? swa_launch_narrow_bound_pass(K, &p, persistent_blocks(db, p.nbatches), st)
                    : swa_launch_narrow_pass(K, &p, persistent_blocks(db, p.nbatches), st));
    }
  }
  return SWA_OK;
}

// The same for two queries (swa_dual_kernel's MP build over the one-sequence-per-row stream): passes of at most 56 rows
// per lane for nucleotide alphabets (1 KB of LDS per residue code and 4 rows), 32 for the others.
void dual_pass_shape(int64_t qlen, int nres, int* npass, int* K)
{
  const int kmax = nres == 16 ? 56 : 32;
  const int64_t np = (qlen + 16 * kmax - 1) / (16 * kmax);
  *npass = int(np);
  *K = int(std::max<int64_t>(nres == 16 ? 32 : 17, (qlen + 16 * np - 1) / (16 * np)));
}
int dual_pass_rows(int64_t qlen, int nres)
{
  int npass = 0, K = 0;
  dual_pass_shape(qlen, nres, &npass, &K);
  return K;
}

int launch_dual_passes(swa_db* db, const BatchSet& bs, int64_t qlen, int nres, hipStream_t st, int64_t qlen_a = 0, int64_t qlen_b = 0)
{
  int npass = 0, K = 0;
  dual_pass_shape(qlen, nres, &npass, &K);
  swa_mp_params p{};
  p.nibbles = bs.nibbles ? 1 : 0;
  p.qlen_a = int32_t(qlen_a);
  p.qlen_b = int32_t(qlen_b);
  p.qseq = db->qseq_p;
  p.qseq2 = db->qseq2_p;
  p.matrix = db->matrix.p;
  p.qlen = int32_t(qlen);
  p.rows_per_lane = K;
  p.npass = npass;
  p.stream = bs.sp();
  p.counter = db->ctl.p + 0;
  p.scores = db->scores.p;
  p.scores2 = db->scores2.p;
  p.limit = f16_limit(db, K);
  p.ovf_count = db->ctl.p + 1;
  p.ovf_list = db->ovf_list.p;
  p.ovf_count2 = db->ctl.p + 3;
  p.ovf_list2 = db->ovf_list2.p;
  p.gapextend_f = float(db->ge);
  p.negQR = f16_pair(-float(db->goe - db->ge));
  p.negR = f16_pair(-float(db->ge));
  p.negKR = f16_pair(-float(int64_t(K) * db->ge));
  for (int i = 0; i <= K + 1; ++i) p.rowc[i] = f16_pair(float(int64_t(i) * db->ge));
  PassRuns runs;
  const int prc = plan_pass_runs(db, bs, runs);
  if (prc != SWA_OK) return prc;
  p.boundary = db->boundary.p;
  for (size_t i = 0; i + 1 < runs.cut.size(); ++i) {
    p.batches = bs.batches.p + runs.cut[i];
    p.slots = bs.slots.p + size_t(runs.cut[i]) * SWA_SLOTS;
    p.nbatches = runs.cut[i + 1] - runs.cut[i];
    p.boundary_base = runs.first_chunk[i];
    for (int pass = 0; pass < npass; ++pass) {
      p.row0 = pass * 16 * K;
      p.pass = pass;
      p.last = pass + 1 == npass;
      HIP_TRY(hipMemsetAsync(db->ctl.p + 0, 0, sizeof(int32_t), st));
      HIP_TRY(swa_launch_dual_pass(K, nres, &p, db->cus, st));
    }
  }
  return SWA_OK;
}

// ---- control block -------------------------------------------------------------------------------------------
// db->ctl is ONE device allocation: 16 ints of counters followed by the candidate records of the hit filter, so that
// one asynchronous copy into the page-locked block db->pin brings back everything a search wants on the host and
// the search synchronises with the stream ONCE:
//   [0] work-queue head of the first-pass kernel     [1] re-queue count, query 1     [3] re-queue count, query 2
//   [4] [5] work-queue heads of the device-driven re-queue kernels                   [8] candidate count
//   [10..13] two 64-bit tallies (totalhits, obvious)                                 [16..] swa_cand records
constexpr int CTL_INTS = 48;            // three 64-byte lines of counters (the third: tallies of a pair's second query)
constexpr int CTL_CAND = 8, CTL_TALLY = 10;
constexpr int CTL_LOADFLAGS = 6;        // OR of every byte the loader has stripped so far (a part-wise first pass copies it here)

constexpr int CAND_EAGER = 4096;        // candidate records copied back together with the counters
constexpr int REQUEUE_CAP = 1 << 16;    // sequences the device-driven re-queue takes; longer lists go through the host

inline const int32_t* ctl_host(const swa_db* db) { return reinterpret_cast<const int32_t*>(db->pin); }
inline swa_cand* cand_dev(swa_db* db) { return reinterpret_cast<swa_cand*>(db->ctl.p + CTL_INTS); }

int ensure_pin(swa_db* db, size_t bytes)
{
  if (bytes <= db->pin_bytes) return SWA_OK;
  if (db->pin) { (void)hipHostFree(db->pin); db->pin = nullptr; db->pin_bytes = 0; }
  HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&db->pin), bytes, hipHostMallocDefault));
  db->pin_bytes = bytes;
  return SWA_OK;
}
// page-locked block: [ctl copy + eager candidates | query upload area]
constexpr size_t PIN_CTL_BYTES = CTL_INTS * sizeof(int32_t) + size_t(CAND_EAGER) * sizeof(swa_cand);

// counters (+ the first `ncand` candidate records) to the host, then wait for the stream: the one synchronisation
int sync_ctl(swa_db* db, int ncand, hipStream_t st)
{
  HIP_TRY(hipMemcpyAsync(db->pin, db->ctl.p, CTL_INTS * sizeof(int32_t) + size_t(ncand) * sizeof(swa_cand),
                         hipMemcpyDeviceToHost, st));
  if (db->opt.watchdog_s <= 0) {
    HIP_TRY(hipStreamSynchronize(st));
    return SWA_OK;
  }
  // option "watchdog_s": poll instead of blocking; a stream that does not drain in time is reported with the control block
  // as the device holds it (read over a stream of its own) - queue heads, re-queue counts -
  // and the call fails.  The handle is not usable afterwards (kernels may still be spinning): close the process.
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    const hipError_t q = hipStreamQuery(st);
    if (q == hipSuccess) return SWA_OK;
    if (q != hipErrorNotReady) HIP_TRY(q);
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > double(db->opt.watchdog_s)) break;
    std::this_thread::sleep_for(std::chrono::microseconds(200));
  }
  std::string msg = "search did not come back within " + std::to_string(db->opt.watchdog_s) + " s; control block";
  hipStream_t side = nullptr;
  int32_t snap[CTL_INTS] = {};
  if (hipStreamCreateWithFlags(&side, hipStreamNonBlocking) != hipSuccess) side = nullptr;
  for (int shot = 0; shot < 2 && side; ++shot) {          // twice, a second apart: stuck, or crawling?
    if (shot) std::this_thread::sleep_for(std::chrono::seconds(1));
    if (hipMemcpyAsync(snap, db->ctl.p, sizeof snap, hipMemcpyDeviceToHost, side) == hipSuccess && hipStreamSynchronize(side) == hipSuccess) {
      msg += shot ? " || one second later" : "";
      for (int i = 0; i < CTL_INTS; ++i) msg += (i % 16 == 0 ? " | " : " ") + std::to_string(snap[i]);
    } else {
      msg += " unreadable";
    }
  }
  if (side) (void)hipStreamDestroy(side);
  return fail(SWA_ENODEV, msg);
}

// query (+ its descriptor for the first-pass kernel) in ONE copy out of page-locked memory:
// device layout of db->qblock = [swa_query | residues of query 1 | residues of query 2]
// The second query of a two-query search when it is a DIFFERENT query (swa_search_pair_topk) rather than another strand
// or frame of the first: its own length (the shorter query is padded to `qlen` rows that score -1 against everything)
// and its own score window.  Zero lengths = both queries have qlen rows and share the window.
struct Pair {
  int64_t qlen_a = 0, qlen_b = 0;
  bool own_window = false;
  int64_t minscore_b = 0, maxscore_b = 0, keep_b = 0;
  int64_t total_b = 0, obvious_b = 0;          // out
};
constexpr int CTL_TALLY_B = 36;                // tallies of the second score array of a pair (third counter line)

int upload_queries(swa_db* db, const uint8_t* q1, const uint8_t* q2, int64_t qlen, hipStream_t st, int64_t qlen_a = 0,
                   int64_t qlen_b = 0)
{
  const size_t qpad = (size_t(qlen) + 15) & ~size_t(15);
  const size_t bytes = 32 + qpad * (q2 ? 2 : 1);
  int rc = ensure_pin(db, PIN_CTL_BYTES + 32 + 2 * std::max<size_t>(qpad, 4096));
  if (rc != SWA_OK) return rc;
  HIP_TRY(db->qblock.reserve(32 + 2 * std::max<size_t>(qpad, 4096)));
  unsigned char* h = db->pin + PIN_CTL_BYTES;
  db->qseq_p = db->qblock.p + 32;
  db->qseq2_p = db->qblock.p + 32 + qpad;
  swa_query hq{db->qseq_p, db->matrix.p, int32_t(qlen)};
  std::memset(h, 0, 32);
  std::memcpy(h, &hq, sizeof hq);
  const size_t na = size_t(qlen_a ? qlen_a : qlen), nb = size_t(qlen_b ? qlen_b : qlen);
  std::memset(h + 32, SWA_PAD, qpad * (q2 ? 2 : 1));     // rows past a query's end: never scored (profile builders check the row)
  std::memcpy(h + 32, q1, na);
  if (q2) std::memcpy(h + 32 + qpad, q2, nb);
  HIP_TRY(hipMemcpyAsync(db->qblock.p, h, bytes, hipMemcpyHostToDevice, st));
  return SWA_OK;
}

int read_requeue(swa_db* db, int ctl_index, const int32_t* list, std::vector<int32_t>& out, hipStream_t st)
{
  int32_t n = 0;
  HIP_TRY(hipMemcpyAsync(&n, db->ctl.p + ctl_index, sizeof n, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  out.clear();
  if (n > 0) {
    out.resize(size_t(n));
    HIP_TRY(hipMemcpy(out.data(), list, size_t(n) * sizeof(int32_t), hipMemcpyDeviceToHost));
    std::sort(out.begin(), out.end());                 // deterministic whatever the wave timing was
  }
  return SWA_OK;
}

// int32 is exact for a wave per sequence when qlen x highest score stays below 2^30 (then nothing can reach the
// 64-bit hop either)
bool wave_requeue_ok(const swa_db* db, int64_t qlen)
{
  return db->opt.wave_requeue != 0 && qlen < (int64_t(1) << 24) &&
         std::max<int64_t>(qlen, 1) * std::max<int64_t>(db->hi, 1) < (int64_t(1) << 30) &&
         db->goe < (int64_t(1) << 30) && db->ge < (int64_t(1) << 30);
}
// ... and the list can be worked off without the host ever seeing it when one pass of the wave kernel holds the query
bool device_requeue_ok(const swa_db* db, int64_t qlen)
{
  return wave_requeue_ok(db, qlen) && !db->opt.requeue_host && qlen <= 64 * swa_endpoints_rows_for(int(qlen));
}

// the device-driven re-queue behind a first pass: four waves per sequence where that shortens the chain of a step, else one
hipError_t launch_requeue(const swa_db* db, const swa_seqs& sq, const int32_t* list, const int32_t* count, int32_t* work, const uint8_t* qseq,
                          int64_t qlen, int* scores, hipStream_t st, swa_counters_t& c)
{
  const int kb = swa_requeue_block_rows_for(int(std::min<int64_t>(qlen, 1 << 20)));
  const bool block = kb > 0 && db->opt.requeue_block == 1;
  c.requeue_form = block ? 2 : 1;
  if (block)
    return swa_launch_requeue_block(&sq, list, count, REQUEUE_CAP, work, qseq, int(qlen), db->matrix.p, int(db->goe), int(db->ge), scores,
                                    db->cus * 8, st);
  return swa_launch_requeue_wave(&sq, list, count, REQUEUE_CAP, work, qseq, int(qlen), db->matrix.p, int(db->goe), int(db->ge), scores,
                                 db->cus * 8, st);
}

// 32-bit then 64-bit kernels over a re-queue list the host holds, one query; results land in `scores`
// (64-bit values in `s64` with the sentinel in `scores`).
int run_wide(swa_db* db, std::vector<int32_t>& requeue, const uint8_t* qdev, int64_t qlen, int32_t* scores,
             DevBuf<long long>& s64, int64_t* n32, int64_t* n64, hipStream_t st)
{
  // A short list is latency-bound in the batch kernels (one 16-lane chain per sequence, the longest sequence sets the
  // time): a wave per sequence - the end-point kernel of the alignment phase, 64 lanes on one sequence - finishes the
  // 1 500 sequences the bound build sends back for the bench query in 1.3 ms instead of 1.5, the 390 of a 5 000-row
  // query in a fraction of the batch kernel's 20 passes.
  {
    // the host-driven paths below read the shard's length order and build sets from it: a shard whose first pass went part
    // by part is complete by now (its parts' events are behind us) - adopt the loader's tables before touching them
    const int lrc = settle_loading(db, true, nullptr);
    if (lrc != SWA_OK) return lrc;
  }
  const swa_seqs sq = db->seqs();
  bool by_wave = !requeue.empty() && requeue.size() <= size_t(REQUEUE_CAP) && int64_t(requeue.size()) < db->nseq &&
                 wave_requeue_ok(db, qlen);
  const bool passes = by_wave && qlen > 64 * swa_endpoints_rows_for(int(qlen));   // over 2 048 rows: hand-over per column
  std::vector<int64_t> boff;
  int64_t columns = 0;
  if (passes) {
    boff.resize(requeue.size());
    for (size_t i = 0; i < requeue.size(); ++i) {
      boff[i] = columns;
      columns += db->len_of(requeue[i]);
    }
    by_wave = columns <= (int64_t(1) << 28);           // 2 x 4 bytes of hand-over per column: at most 2 GB
  }
  if (by_wave) {
    HIP_TRY(db->rq_ids.reserve(requeue.size()));
    HIP_TRY(hipMemcpyAsync(db->rq_ids.p, requeue.data(), requeue.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    if (passes) {
      HIP_TRY(db->rq_bh.reserve(size_t(columns) + 1));
      HIP_TRY(db->rq_bf.reserve(size_t(columns) + 1));
      HIP_TRY(db->rq_boff.reserve(requeue.size()));
      HIP_TRY(hipMemcpyAsync(db->rq_boff.p, boff.data(), boff.size() * sizeof(int64_t), hipMemcpyHostToDevice, st));
    }
    HIP_TRY(swa_launch_endpoints_wave(&sq, db->rq_ids.p, nullptr, int(requeue.size()), qdev, int(qlen),
                                      db->matrix.p, int(db->goe), int(db->ge), passes ? db->rq_bh.p : nullptr,
                                      passes ? db->rq_bf.p : nullptr, passes ? db->rq_boff.p : nullptr, nullptr, scores, st));
    HIP_TRY(hipStreamSynchronize(st));                 // the id list lives in the caller's vector
    *n32 += int64_t(requeue.size());
    requeue.clear();
    return SWA_OK;
  }
  for (int bits = 32; !requeue.empty() && bits <= 64; bits += 32) {
    const BatchSet* set = &db->scratch;
    if (int64_t(requeue.size()) == db->nseq) {
      const int rc = ensure_single(db);
      if (rc != SWA_OK) return rc;
      set = &db->single;
    } else {
      std::vector<int32_t> ordered;
      order_by_length([&](int32_t id) { return db->len_of(id); }, requeue.data(), int64_t(requeue.size()), ordered);
      const int rc = build_batches(db, ordered.data(), int64_t(ordered.size()), 1, db->scratch);
      if (rc != SWA_OK) return rc;
    }
    if (bits == 64) HIP_TRY(s64.reserve(size_t(db->nseq + db->nwin)));
    HIP_TRY(hipMemsetAsync(db->ctl.p + 1, 0, sizeof(int32_t), st));
    MpRun r;
    r.mode = bits == 32 ? 2 : 3;
    r.set = set;
    r.q1 = qdev;
    r.scores = scores;
    r.scores64 = s64.p;
    r.ovf_count = db->ctl.p + 1;
    r.ovf_list = db->ovf_list.p;
    const int rc = launch_mp_run(db, r, qlen, st);
    if (rc != SWA_OK) return rc;
    *(bits == 32 ? n32 : n64) += int64_t(requeue.size());
    requeue.clear();
    if (bits == 32) {
      const int rc2 = read_requeue(db, 1, db->ovf_list.p, requeue, st);
      if (rc2 != SWA_OK) return rc2;
    }
  }
  return SWA_OK;
}

int check_query(const swa_db* db, const uint8_t* q, int64_t qlen)
{
  if (!db) return fail(SWA_EINVAL, "null database handle");
  if (!db->scoring_set) return fail(SWA_ESTATE, "swa_set_scoring must be called before searching");
  if (qlen < 0 || (qlen > 0 && !q)) return fail(SWA_EINVAL, "bad query");
  if (qlen > (1 << 20)) return fail(SWA_EINVAL, "query longer than 2^20 residues");
  for (int64_t i = 0; i < qlen; ++i)
    if (q[i] >= 32) return fail(SWA_EINVAL, "query symbol code out of range (must be < 32)");
  return SWA_OK;
}

bool f16_applicable(const swa_db* db)
{
  // f16 pairs are exact while every value stays within +-2048: needs modest scores and penalties
  return db->hi >= 0 && db->hi < 512 && db->lo > -1024 && db->goe >= db->ge && db->goe <= 1024 && db->ge >= 0 &&
         db->ge <= 16;
}
// What a search has enqueued and not yet seen on the host
struct Pending {
  swa_counters_t c{};
  bool used_bound = false;
  bool dev1 = false, dev2 = false;   // the re-queue list of query 1 / 2 was worked off by the device-driven kernel
  bool two = false;
  bool empty = false;
  bool windows = false;              // the first pass ran over a view: window scores are folded into their parents
};

int finish_empty(swa_db* db, Pending& pd, bool two, hipStream_t st)
{
  if (db->nseq) {
    HIP_TRY(hipMemsetAsync(db->scores.p, 0, size_t(db->nseq) * sizeof(int32_t), st));
    if (two) {
      HIP_TRY(db->scores2.reserve(size_t(db->nseq)));
      HIP_TRY(hipMemsetAsync(db->scores2.p, 0, size_t(db->nseq) * sizeof(int32_t), st));
      HIP_TRY(swa_launch_mark_excluded(db->scores2.p, db->excluded.p, int(db->n_excluded), st));
    }
    HIP_TRY(swa_launch_mark_excluded(db->scores.p, db->excluded.p, int(db->n_excluded), st));
  }
  HIP_TRY(hipMemsetAsync(db->ctl.p, 0, CTL_INTS * sizeof(int32_t), st));
  pd.empty = true;
  pd.two = two;
  return SWA_OK;
}

bool bound_wanted(const swa_db* db, int64_t qlen, int64_t bound_min)
{
  const int Nb = swa_bound_period();
  if (bound_min <= 0 || db->opt.bound == 0) return false;
  if (db->opt.bound == 1) return true;
  return !db->bound_is_off(qlen, bound_min) && bound_min >= 4 * int64_t(Nb) * db->ge &&
         bound_min > int64_t(Nb + 2) * db->ge + db->goe;
}

// the escalation loop: packed f16 -> 32 bit -> 64 bit.  Scores end up in db->scores / scores64.  Everything is
// ENQUEUED on the handle's stream; the caller synchronises (sync_ctl) and calls settle_search.
// bound_min > 0: the caller only wants the sequences scoring at least bound_min (top-K searches); the first pass may
// then be the bound build of the kernel (sw_cb_kernels.hip), which leaves placeholders below bound_min for the rest
int run_search(swa_db* db, const uint8_t* query, int64_t qlen, int64_t bound_min, Pending& pd)
{
  int rc = check_query(db, query, qlen);
  if (rc != SWA_OK) return rc;
  HIP_TRY(hipSetDevice(db->device));
  hipStream_t st = db->stream;
  pd = Pending{};
  swa_counters_t& c = pd.c;
  c.cells = db->active_sym * qlen;
  if (bound_min > 0) db->cur_qhash = swa_db::hash_query(query, qlen, nullptr, 0);
  rc = ensure_pin(db, PIN_CTL_BYTES + 32 + 8192);
  if (rc != SWA_OK) return rc;
  // a shard that is still loading (sw_loading.inc): the first pass goes part by part if it is a single-pass build
  bool loading = false;
  rc = settle_loading(db, false, &loading);
  if (rc != SWA_OK) return rc;
  if (qlen == 0 || (db->h_order.empty() && !loading)) return finish_empty(db, pd, false, st);
  const bool f16 = f16_applicable(db);
  const int K = swa_narrow_rows_for(int(std::min<int64_t>(qlen, 4096)));     // tuned single-pass kernels: qlen <= 1024
  const bool force_mp = db->opt.force_mp == 1;
  const bool single_pass = qlen <= 16 * 58 && K > 0 && !force_mp;
  if (loading && (db->packed || !(f16 && single_pass && db->opt.narrow_variant != 1))) {     // any other first pass (and a nucleotide shard's pair stream) wants the whole shard
    rc = settle_loading(db, true, nullptr);
    if (rc != SWA_OK) return rc;
    loading = false;
  }
  HIP_TRY(hipEventRecord(db->ev[0], st));
  rc = upload_queries(db, query, nullptr, qlen, st);
  if (rc != SWA_OK) return rc;
  const swa_query* dquery = reinterpret_cast<const swa_query*>(db->qblock.p);
  HIP_TRY(hipMemsetAsync(db->ctl.p, 0, CTL_INTS * sizeof(int32_t), st));
  if (!loading) {
    rc = ensure_main(db);                                // nucleotide shards build their pair stream on first use
    if (rc != SWA_OK) return rc;
  }
  std::vector<int32_t> requeue;
  bool used_bound = false;
  // Lanes per sequence pair G and rows per lane K = ceil(qlen / G) of the single-pass build: the argmax over the measured
  // table of every build that exists (kernel_choice.cpp; option "lanes" pins the chain length: A/B runs and tests).  One
  // lane per pair for short queries, chains of 2 / 4 / 8 / 16 lanes beyond; the bound build of the same shape for top-K
  // searches whose threshold clears its slack (6 instead of 7.5 instructions per cell pair, two state registers per row
  // instead of three: its lanes reach 60..62 rows where the exact build stops at 48).
  const int Nb = swa_bound_period();
  const bool want_bound = bound_wanted(db, qlen, bound_min);
  swa::ChoiceEnv env;
  env.qlen = qlen;
  env.want_bound = want_bound;
  env.hi = db->hi; env.goe = db->goe; env.ge = db->ge;
  env.longest = db->longest;
  env.mean_len = loading ? double(db->active_sym) / double(db->nseq) : db->h_order.empty() ? 325.0 : double(db->active_sym) / double(db->h_order.size());
  env.lanes = int(db->opt.lanes);
  env.long_lanes = db->opt.long_lanes != 0;
  env.bound_period = Nb;
  const swa::KernelPick pick = single_pass ? swa::pick_first_pass(env) : swa::KernelPick{};
  const int G = pick.G, Kg = pick.K;
  if (loading && !(Kg > 0 && f16_limit(db, Kg) >= 1024)) {
    rc = settle_loading(db, true, nullptr);
    if (rc == SWA_OK) rc = ensure_main(db);
    if (rc != SWA_OK) return rc;
    loading = false;
  }
  Loading* const LD = loading ? db->loading.get() : nullptr;
  const BatchSet* bsp = &db->main;                       // ... or the view that cuts long sequences into windows
  if (!loading) {                                        // (the parts of a loading shard are searched as they are)
    rc = prepare_view(db, db->main, 2, qlen, &bsp, slots_per_cu(G, Kg, pick.bound, 2));
    if (rc != SWA_OK) return rc;
  }
  const BatchSet& bs = *bsp;
  const swa_seqs sq = db->seqs();
  HIP_TRY(hipEventRecord(db->ev[1], st));
  if (f16 && single_pass && Kg > 0 && f16_limit(db, Kg) >= 1024 && db->opt.narrow_variant != 1) {
    const int K = Kg;
    swa_narrow_params p{};
    p.query = dquery;
    p.stream = bs.sp();
    p.batches = bs.batches.p;
    p.slots = bs.slots.p;
    p.nbatches = bs.nbatches;
    p.counter = db->ctl.p + 0;
    p.scores = db->scores.p;
    p.ovf_count = db->ctl.p + 1;
    p.ovf_list = db->ovf_list.p;
    p.negQ = f16_pair(-float(db->goe));
    p.negR = f16_pair(-float(db->ge));
    p.shifted = 1;                                     // row-shifted form, 7.5 ops per cell pair
    p.limit = int32_t(f16_limit(db, K));
    p.gapextend_f = float(db->ge);
    p.negQR = f16_pair(-float(db->goe - db->ge));
    p.negKR = f16_pair(-float(int64_t(K) * db->ge));
    for (int r = 0; r <= K + 1; ++r) p.rowc[r] = f16_pair(float(int64_t(r) * db->ge));
    c.narrow_rows = K;
    c.narrow_shifted = G == 8 ? 2 : G == 4 ? 3 : G == 2 ? 7 : G == 1 ? 11 : 1;
    const int per_wave = 16 / G;                                    // a wave takes 16 / G batches at a time ...
    // ... and the bound builds on chains of lanes take `concat` such sets one behind the other without draining in between
    // (sw_cb_kernel.inc), except at the tail of the queue: the last four sets per resident wave - the shortest sequences, which
    // decide when the last wave finishes - are handed out one at a time.  Returns the number of items of the queue.
    auto plan_items = [&](swa_narrow_params& q) -> int {
      const int sets = (q.nbatches + per_wave - 1) / per_wave;
      q.concat = 1;
      q.concat_items = 0;
      if (pick.bound && G > 1 && db->opt.concat != 1 && db->opt.concat != 0) {
        const int M = db->opt.concat > 0 ? int(std::min<int64_t>(db->opt.concat, 64)) : 16;
        const int64_t resident = int64_t(db->cus) * 4 * waves_for_rows(K, true);
        const int tail = int(std::min<int64_t>(sets, db->opt.concat_tail >= 0 ? db->opt.concat_tail : 4 * resident));
        q.concat_items = (sets - tail) / M;
        if (q.concat_items > 0) q.concat = M;
      }
      return q.concat_items + (sets - q.concat_items * q.concat);
    };
    const int items = plan_items(p);
    p.twin = db->opt.twin != 0 ? 1 : 0;
    int blocks = persistent_blocks(db, items);
    p.pipe = int32_t(db->opt.pipe);
    if (db->opt.blocks_per_cu > 0) blocks = std::max(1, std::min((items + 3) / 4, db->cus * int(db->opt.blocks_per_cu)));
    // Bound build (6 instead of 7.5 instructions per cell pair): its result is at most (period - 1) R above the score,
    // everything at or above bound_min is recomputed by the 32-bit kernel.  Used when the threshold is far enough above
    // that slack for the recomputed share to be negligible (option "bound" = 0 never, 1 whenever a build exists); if more
    // than 2 % of the sequences come back it is switched off for this (query length, threshold) and the exact kernel runs
    used_bound = pick.bound;
    // The re-queue list is worked off by ONE small kernel enqueued behind this one in the same stream (below).  Rounds 2-3
    // ran it BESIDE the first pass, as a second kernel on a second stream polling the list - which rests on two kernels being
    // resident together (HIP promises no such thing; on MI355X a 512-thread producer froze beside spinning followers) and was
    // taken out in round 4; DESIGN 4.10 has the numbers and what was tried instead.
    if (used_bound) {
      p.limit = int32_t(std::min<int64_t>(f16_limit(db, K + Nb), bound_min));
      for (int r = 0; r <= K + Nb + 1; ++r) p.rowc[r] = f16_pair(float(int64_t(r) * db->ge));
      c.narrow_shifted = 8;
    }
    // the build the table chose, over whatever set of batches q names
    auto launch_first = [&](const swa_narrow_params& q, int nblocks, hipStream_t s) -> hipError_t {
      if (used_bound)
        return G == 1 ? (K <= 24 ? swa_launch_one_bound_c(K, &q, nblocks, s) : K <= 48 ? swa_launch_one_bound_d(K, &q, nblocks, s) : swa_launch_one_bound_e(K, &q, nblocks, s))
                      : K > 48 && G == 2 ? swa_launch_narrow_bound_long2(K, &q, nblocks, s) : K > 48 && G == 4 ? swa_launch_narrow_bound_long4(K, &q, nblocks, s)
                      : K > 48 && G == 8 ? swa_launch_narrow_bound_long8(K, &q, nblocks, s)
                      : G == 2 ? swa_launch_narrow_bound_g2(K, &q, nblocks, s) : G == 4 ? swa_launch_narrow_bound_g4(K, &q, nblocks, s) : G == 8 ? swa_launch_narrow_bound_g8(K, &q, nblocks, s)
                      : swa_launch_narrow_bound_g16(K, &q, nblocks, s);
      if (G == 1) return K <= 24 ? swa_launch_narrow_one_a(K, &q, nblocks, s) : swa_launch_narrow_one_b(K, &q, nblocks, s);
      return swa_launch_narrow_split(G, K, &q, nblocks, s);
    };
    if (loading) {
      // Part by part, as the loader publishes them.  Each launch has a queue head of its own; the launches alternate
      // between the handle's two streams, so the blocks of part i + 1 move in as those of part i run out (no kernel waits
      // for another: the overlap is the hardware's to give, nothing depends on it).
      const int P = int(LD->parts.size());
      HIP_TRY(hipMemsetAsync(LD->heads.p, 0, size_t(P) * sizeof(int32_t), st));
      HIP_TRY(hipEventRecord(db->ev2[0], st));                         // query, matrix, counters: in place
      HIP_TRY(hipStreamWaitEvent(db->stream2, db->ev2[0], 0));
      for (int i = 0; i < P; ++i) {
        rc = wait_part(LD, i);
        if (rc != SWA_OK) return rc;
        const LoadPart& part = LD->parts[size_t(i)];
        hipStream_t ps = (i & 1) ? db->stream2 : st;
        HIP_TRY(hipStreamWaitEvent(ps, part.ready, 0));
        swa_narrow_params q = p;
        q.stream = LD->arena.p;                                        // the part's batches address its region of the one stream
        q.batches = LD->pbatches.p + part.batch_base;
        q.slots = LD->pslots.p + size_t(part.batch_base) * SWA_SLOTS;
        q.nbatches = int32_t(part.plan.batches.size());
        q.counter = LD->heads.p + i;
        if (q.nbatches == 0) continue;
        const int pitems = plan_items(q);
        int pblocks = persistent_blocks(db, pitems);
        if (db->opt.blocks_per_cu > 0) pblocks = std::max(1, std::min((pitems + 3) / 4, db->cus * int(db->opt.blocks_per_cu)));
        HIP_TRY(launch_first(q, pblocks, ps));
      }
      if (P > 1) {
        HIP_TRY(hipEventRecord(db->ev2[1], db->stream2));
        HIP_TRY(hipStreamWaitEvent(st, db->ev2[1], 0));
      }
      c.loading_parts = P;
      // every unterminate kernel is ordered before the last part's event, so the OR of all residue bytes is final here: it
      // comes back with the counters and settle_search refuses scores computed from codes >= 32 (they index outside the LDS
      // profile and the 32 x 32 matrix; the resident reader rejects such a volume before any kernel runs)
      HIP_TRY(hipMemcpyAsync(db->ctl.p + CTL_LOADFLAGS, LD->flags.p, sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    } else {
      HIP_TRY(launch_first(p, blocks, st));
    }
    c.narrow = db->nseq;
  } else if (f16 && !force_mp && qlen <= 1024 && K > 0 && db->hi < 1024 && (db->opt.narrow_variant == 1 || f16_limit(db, K) < 1024)) {
    swa_narrow_params p{};                             // plain form (8.5 ops): K*R would eat the f16 range
    p.query = dquery;
    p.stream = bs.sp();
    p.batches = bs.batches.p;
    p.slots = bs.slots.p;
    p.nbatches = bs.nbatches;
    p.counter = db->ctl.p + 0;
    p.scores = db->scores.p;
    p.limit = int32_t(2048 - db->hi);
    p.ovf_count = db->ctl.p + 1;
    p.ovf_list = db->ovf_list.p;
    p.negQ = f16_pair(-float(db->goe));
    p.negR = f16_pair(-float(db->ge));
    c.narrow_rows = K;
    HIP_TRY(swa_launch_narrow(K, &p, persistent_blocks(db, p.nbatches), st));
    c.narrow = db->nseq;
  } else if (f16 && !force_mp && qlen > 16 * 58 && f16_limit(db, split_pass_rows(qlen)) >= 1024) {
    int np = 0, Kp = 0;                                // long query: passes of the tuned kernel, or of its bound build
    split_pass_shape(qlen, &np, &Kp);
    used_bound = want_bound && f16_limit(db, Kp + Nb) >= 1024;
    rc = launch_split_passes(db, bs, qlen, st, used_bound, bound_min);
    if (rc != SWA_OK) return rc;
    c.narrow_rows = Kp;
    c.narrow_shifted = used_bound ? 9 : 5;
    c.narrow = db->nseq;
  } else if (f16 && f16_limit(db, mp_rows_for(0, qlen)) >= 1024) {
    MpRun r;                                           // multi-pass pair kernel (short passes: large gap-extension penalties)
    r.mode = 0;
    r.set = &bs;
    r.q1 = db->qseq_p;
    r.scores = db->scores.p;
    r.ovf_count = db->ctl.p + 1;
    r.ovf_list = db->ovf_list.p;
    rc = launch_mp_run(db, r, qlen, st);
    if (rc != SWA_OK) return rc;
    c.narrow_rows = mp_rows_for(0, qlen);
    c.narrow_shifted = 1;
    c.narrow = db->nseq;
  }
  HIP_TRY(hipEventRecord(db->ev[2], st));
  pd.used_bound = used_bound;
  if (c.narrow && device_requeue_ok(db, qlen)) {
    // the list stays on the device: a persistent grid of waves takes entries off it until the count the first pass left
    HIP_TRY(launch_requeue(db, sq, db->ovf_list.p, db->ctl.p + 1, db->ctl.p + 4, db->qseq_p, qlen, db->scores.p, st, c));
    pd.dev1 = true;
  } else {
    if (c.narrow) {
      rc = read_requeue(db, 1, db->ovf_list.p, requeue, st);
      if (rc != SWA_OK) return rc;
      if (used_bound && int64_t(requeue.size()) * 50 > db->nseq && db->opt.bound != 1) {
        db->note_bound_off(qlen, bound_min);   // threshold too close to the bulk of the scores
        return run_search(db, query, qlen, 0, pd);
      }
    } else {
      requeue.assign(db->h_order.begin(), db->h_order.end());
    }
    rc = run_wide(db, requeue, db->qseq_p, qlen, db->scores.p, db->scores64, &c.wide, &c.full, st);
    if (rc != SWA_OK) return rc;
    c.requeue_form = 3;
  }
  pd.windows = c.narrow && bsp != &db->main;
  if (pd.windows) { rc = fold_windows(db, false, st); if (rc != SWA_OK) return rc; }
  HIP_TRY(swa_launch_mark_excluded(db->scores.p, db->excluded.p, int(db->n_excluded), st));
  HIP_TRY(hipEventRecord(db->ev[3], st));
  return SWA_OK;
}

// Two queries of equal length against every sequence in one pass (nucleotide plus/minus strand).
// Scores of query 1 end up in db->scores (64-bit values in scores64), of query 2 in db->scores2 (scores64b).
int run_search2(swa_db* db, const uint8_t* q1, const uint8_t* q2, int64_t qlen, int64_t bound_min, Pending& pd,
                int64_t qlen_a = 0, int64_t qlen_b = 0)
{
  const int64_t qa = qlen_a ? qlen_a : qlen, qb = qlen_b ? qlen_b : qlen;      // rows of query 1 / 2; qlen = the longer
  int rc = check_query(db, q1, qa);
  if (rc == SWA_OK) rc = check_query(db, q2, qb);
  // a shard that is still loading (sw_loading.inc): the single-pass two-query builds go part by part when the parts are in the
  // layout they stream - the 4-bit one-sequence-per-row parts of a nucleotide shard under 16-lane chains (the reference's
  // both-strand search, swipe.cc:1403), the pair-stream parts of a protein shard under chains of 2 / 4 / 8 lanes
  bool loading = false;
  if (rc == SWA_OK) rc = settle_loading(db, false, &loading);
  if (rc != SWA_OK) return rc;
  HIP_TRY(hipSetDevice(db->device));
  hipStream_t st = db->stream;
  pd = Pending{};
  pd.two = true;
  swa_counters_t& c = pd.c;
  c.cells = db->active_sym * (qa + qb);
  if (bound_min > 0) db->cur_qhash = swa_db::hash_query(q1, qa, q2, qb);
  rc = ensure_pin(db, PIN_CTL_BYTES + 32 + 8192);
  if (rc != SWA_OK) return rc;
  if (qlen == 0 || (db->h_order.empty() && !loading)) return finish_empty(db, pd, true, st);
  HIP_TRY(hipEventRecord(db->ev[0], st));
  rc = upload_queries(db, q1, q2, qlen, st, qlen_a, qlen_b);
  if (rc != SWA_OK) return rc;
  HIP_TRY(hipMemsetAsync(db->ctl.p, 0, CTL_INTS * sizeof(int32_t), st));
  std::vector<int32_t> rq1, rq2;
  const bool nib = nibble_stream_ok(db);
  const BatchSet* bsp = nullptr;
  bool windows = false;
  // everything indexed by sequence id also holds the windows of a view (ids nseq + v)
  auto reserve2 = [&](bool view) -> int {
    const size_t ids = size_t(db->nseq + (view ? db->nwin : 0));
    HIP_TRY(db->scores2.reserve(ids));
    HIP_TRY(db->ovf_list2.reserve(ids));
    return SWA_OK;
  };                 // nucleotide shard: 16-lane chains stream 4 bits per base
  bool listed = false;                                   // the first pass left re-queue lists on the device
  bool used_bound = false;
  HIP_TRY(hipEventRecord(db->ev[1], st));
  // single pass with the whole query in registers when it fits (nucleotide alphabets: 1008 rows, others 512);
  // option "dual_mp" = 1 forces the multi-pass kernel (A/B, tests)
  const int nres = db->symtype == SWA_SYMTYPE_NUCLEOTIDE ? 16 : 32;
  const bool dual_mp = db->opt.dual_mp == 1;
  // Lanes per sequence Gd and rows per lane Kd of the single-pass two-query build: as in run_search the argmax over the
  // measured table of every build that exists (kernel_choice.cpp pick_dual; profiles/r03_kernel_rates_dual.txt) - ONE lane
  // per sequence up to 48 nucleotide / 32 other rows (sw_one_dual.hip), chains of 2 / 4 / 8 / 16 lanes beyond (nucleotide:
  // up to 60..63 rows per lane; others 32), and for pairs of protein queries / frames in a top-K search the bound build
  // (sw_cb_dual_*.hip: 17..62 rows on 4 and 8 lanes, 17..32 on 16).  No single-pass build: passes of the 16-lane kernel.
  const int Nb = swa_bound_period();
  swa::ChoiceEnv env;
  env.qlen = qlen;
  env.want_bound = bound_wanted(db, qlen, bound_min);
  env.hi = db->hi; env.goe = db->goe; env.ge = db->ge;
  env.longest = db->longest;
  env.mean_len = loading ? double(db->active_sym) / double(std::max<int64_t>(1, db->nseq))
                         : db->h_order.empty() ? 325.0 : double(db->active_sym) / double(db->h_order.size());
  env.lanes = int(db->opt.lanes);
  env.long_lanes = db->opt.long_lanes != 0;
  env.bound_period = Nb;
  swa::KernelPick pick = dual_mp ? swa::KernelPick{} : swa::pick_dual(env, nres, int(db->opt.dual_kmax));
  if (loading && !db->loading->nt && pick.G == 16 && env.lanes == 0) {
    // A protein shard that is still loading holds its parts in the pair format, which chains of 2 / 4 / 8 lanes stream and 16-lane
    // chains do not.  Where a shorter chain has a build for this query (the bound builds reach 62 rows per lane: a pair of 375-row
    // queries is 8 x 47), take it and follow the loader rather than wait for the whole shard with the marginally faster build
    // (round 6; VERDICT r5: "builds that stream another layout wait for the whole shard").
    for (int g : {8, 4, 2}) {
      env.lanes = g;
      const swa::KernelPick alt = swa::pick_dual(env, nres, int(db->opt.dual_kmax));
      if (alt.G == g && alt.K > 0 && f16_limit(db, alt.K) >= 1024) { pick = alt; break; }
    }
    env.lanes = 0;
  }
  const int Gd = pick.G ? pick.G : 16, Kd = pick.K;
  if (loading) {
    const bool part_ok = f16_applicable(db) && Kd > 0 && f16_limit(db, Kd) >= 1024 &&
                         (db->loading->nt ? (Gd == 16 && nib) : Gd < 16);
    if (!part_ok) {                                        // any other build wants a set the parts are not: wait for the shard
      rc = settle_loading(db, true, nullptr);
      if (rc != SWA_OK) return rc;
      loading = false;
    }
  }
  Loading* const LD = loading ? db->loading.get() : nullptr;
  if (f16_applicable(db) && Kd > 0 && f16_limit(db, Kd) >= 1024) {
    static const BatchSet no_set{};
    const BatchSet* setp = &no_set;
    if (!loading) {
      rc = Gd < 16 ? ensure_main(db) : nib ? ensure_single4(db) : ensure_single(db);
      if (rc != SWA_OK) return rc;
      const BatchSet& whole = Gd < 16 ? db->main : nib ? db->single4 : db->single;
      rc = prepare_view(db, whole, Gd < 16 ? 2 : 1, qlen, &bsp, slots_per_cu(Gd, Kd, pick.bound, 1));
      if (rc == SWA_OK) rc = reserve2(bsp != &whole);
      if (rc != SWA_OK) return rc;
      setp = bsp;
      windows = bsp != &whole;
    } else {
      rc = reserve2(false);                                // (the parts of a loading shard are searched as they are: no windows)
      if (rc != SWA_OK) return rc;
    }
    const BatchSet& set = *setp;
    swa_mp_params p{};
    p.nibbles = loading ? (LD->nt ? 1 : 0) : (set.nibbles ? 1 : 0);
    p.qlen_a = int32_t(qlen_a);
    p.qlen_b = int32_t(qlen_b);
    p.qseq = db->qseq_p;
    p.qseq2 = db->qseq2_p;
    p.matrix = db->matrix.p;
    p.qlen = int32_t(qlen);
    p.rows_per_lane = Kd;
    p.npass = 1;
    p.stream = set.sp();
    p.batches = set.batches.p;
    p.slots = set.slots.p;
    p.nbatches = set.nbatches;
    p.counter = db->ctl.p + 0;
    p.scores = db->scores.p;
    p.scores2 = db->scores2.p;
    p.limit = f16_limit(db, Kd);
    p.ovf_count = db->ctl.p + 1;
    p.ovf_list = db->ovf_list.p;
    p.ovf_count2 = db->ctl.p + 3;
    p.ovf_list2 = db->ovf_list2.p;
    p.gapextend_f = float(db->ge);
    p.negQR = f16_pair(-float(db->goe - db->ge));
    p.negR = f16_pair(-float(db->ge));
    p.negKR = f16_pair(-float(int64_t(Kd) * db->ge));
    for (int i = 0; i <= Kd + 1; ++i) p.rowc[i] = f16_pair(float(int64_t(i) * db->ge));
    // bound build (sw_cb_dual.hip) under the same rule as in run_search
    used_bound = pick.bound;
    if (used_bound) {
      p.limit = std::min<int64_t>(f16_limit(db, Kd + Nb), bound_min);
      for (int i = 0; i <= Kd + Nb + 1; ++i) p.rowc[i] = f16_pair(float(int64_t(i) * db->ge));
      // sequences back to back, as the one-query bound build (option concat; the launcher lays the queue out)
      p.concat = db->opt.concat == 0 || db->opt.concat == 1 ? 1 : int32_t(db->opt.concat > 0 ? std::min<int64_t>(db->opt.concat, 64) : 16);
      p.concat_tail = int32_t(std::min<int64_t>(db->opt.concat_tail, 0x7fffffff));
    }
    auto launch_first2 = [&](const swa_mp_params& q, hipStream_t s) -> hipError_t {
      if (used_bound) return swa_launch_dual_bound(Gd, Kd, &q, db->cus, s);
      if (Gd == 1) return swa_launch_dual_one(Kd, nres, &q, db->cus, s);
      return swa_launch_dual(Kd, nres, Gd, &q, db->cus, s);
    };
    if (loading) {
      // part by part as the loader publishes them, exactly as run_search does: a queue head per launch, the launches
      // alternating between the handle's two streams, every part writing the two score arrays and the two re-queue lists
      const int P = int(LD->parts.size());
      HIP_TRY(hipMemsetAsync(LD->heads.p, 0, size_t(P) * sizeof(int32_t), st));
      HIP_TRY(hipEventRecord(db->ev2[0], st));
      HIP_TRY(hipStreamWaitEvent(db->stream2, db->ev2[0], 0));
      for (int i = 0; i < P; ++i) {
        rc = wait_part(LD, i);
        if (rc != SWA_OK) return rc;
        const LoadPart& part = LD->parts[size_t(i)];
        hipStream_t ps = (i & 1) ? db->stream2 : st;
        HIP_TRY(hipStreamWaitEvent(ps, part.ready, 0));
        swa_mp_params q = p;
        q.stream = LD->arena.p;
        q.batches = LD->pbatches.p + part.batch_base;
        q.slots = LD->pslots.p + size_t(part.batch_base) * SWA_SLOTS;
        q.nbatches = int32_t(part.plan.batches.size());
        q.counter = LD->heads.p + i;
        if (q.nbatches == 0) continue;
        HIP_TRY(launch_first2(q, ps));
      }
      if (P > 1) {
        HIP_TRY(hipEventRecord(db->ev2[1], db->stream2));
        HIP_TRY(hipStreamWaitEvent(st, db->ev2[1], 0));
      }
      c.loading_parts = P;
      if (!LD->nt) HIP_TRY(hipMemcpyAsync(db->ctl.p + CTL_LOADFLAGS, LD->flags.p, sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    } else {
      HIP_TRY(launch_first2(p, st));
    }
    c.narrow_rows = Kd;
    c.narrow_shifted = used_bound ? 10 : Gd == 1 ? 12 : 4;   // single-pass dual kernel / its bound build / one lane per sequence
    c.narrow = db->nseq;
    listed = true;
  } else if (f16_applicable(db) && !dual_mp && Gd == 16 && f16_limit(db, dual_pass_rows(qlen, nres)) >= 1024) {
    rc = nib ? ensure_single4(db) : ensure_single(db);
    if (rc != SWA_OK) return rc;
    const BatchSet& whole = nib ? db->single4 : db->single;
    rc = prepare_view(db, whole, 1, qlen, &bsp);
    if (rc == SWA_OK) rc = reserve2(bsp != &whole);
    windows = bsp != &whole;
    if (rc == SWA_OK) rc = launch_dual_passes(db, *bsp, qlen, nres, st, qlen_a, qlen_b);   // long queries: one launch per pass of the same kernel
    if (rc != SWA_OK) return rc;
    c.narrow_rows = dual_pass_rows(qlen, nres);
    c.narrow_shifted = 6;
    c.narrow = db->nseq;
    listed = true;
  } else if (f16_applicable(db) && f16_limit(db, mp_rows_for(1, qlen)) >= 1024) {
    rc = ensure_single(db);
    if (rc == SWA_OK) rc = prepare_view(db, db->single, 1, qlen, &bsp);
    if (rc == SWA_OK) rc = reserve2(bsp != &db->single);
    if (rc != SWA_OK) return rc;
    windows = bsp != &db->single;
    MpRun r;
    r.mode = 1;
    r.qlen_a = qlen_a;
    r.qlen_b = qlen_b;
    r.set = bsp;
    r.q1 = db->qseq_p;
    r.q2 = db->qseq2_p;
    r.scores = db->scores.p;
    r.scores2 = db->scores2.p;
    r.ovf_count = db->ctl.p + 1;
    r.ovf_list = db->ovf_list.p;
    r.ovf_count2 = db->ctl.p + 3;
    r.ovf_list2 = db->ovf_list2.p;
    rc = launch_mp_run(db, r, qlen, st);
    if (rc != SWA_OK) return rc;
    c.narrow_rows = mp_rows_for(1, qlen);
    c.narrow_shifted = 1;
    c.narrow = db->nseq;
    listed = true;
  }
  HIP_TRY(hipEventRecord(db->ev[2], st));
  pd.used_bound = used_bound;
  if (!listed) { rc = reserve2(false); if (rc != SWA_OK) return rc; }
  const swa_seqs sq = db->seqs();
  if (listed && device_requeue_ok(db, qlen)) {
    HIP_TRY(launch_requeue(db, sq, db->ovf_list.p, db->ctl.p + 1, db->ctl.p + 4, db->qseq_p, qa, db->scores.p, st, c));
    HIP_TRY(launch_requeue(db, sq, db->ovf_list2.p, db->ctl.p + 3, db->ctl.p + 5, db->qseq2_p, qb, db->scores2.p, st, c));
    pd.dev1 = pd.dev2 = true;
  } else {
    if (listed) {
      rc = read_requeue(db, 1, db->ovf_list.p, rq1, st);
      if (rc == SWA_OK) rc = read_requeue(db, 3, db->ovf_list2.p, rq2, st);
      if (rc != SWA_OK) return rc;
      if (used_bound && int64_t(rq1.size() + rq2.size()) * 50 > 2 * db->nseq && db->opt.bound != 1) {
        db->note_bound_off(qlen, bound_min);
        return run_search2(db, q1, q2, qlen, 0, pd, qlen_a, qlen_b);
      }
    } else {
      rq1.assign(db->h_order.begin(), db->h_order.end());
      rq2 = rq1;
    }
    int64_t full2 = 0;
    rc = run_wide(db, rq1, db->qseq_p, qa, db->scores.p, db->scores64, &c.wide, &c.full, st);
    if (rc == SWA_OK) rc = run_wide(db, rq2, db->qseq2_p, qb, db->scores2.p, db->scores64b, &c.wide, &full2, st);
    if (rc != SWA_OK) return rc;
    c.full += full2;
    c.requeue_form = 3;
  }
  pd.windows = listed && windows;
  if (pd.windows) { rc = fold_windows(db, true, st); if (rc != SWA_OK) return rc; }
  HIP_TRY(swa_launch_mark_excluded(db->scores.p, db->excluded.p, int(db->n_excluded), st));
  HIP_TRY(swa_launch_mark_excluded(db->scores2.p, db->excluded.p, int(db->n_excluded), st));
  HIP_TRY(hipEventRecord(db->ev[3], st));
  return SWA_OK;
}

// After sync_ctl: what the device-driven re-queue could not know.  *again: the search must be repeated (the bound
// build sent back too much - it is now off for this query length and threshold); *changed: scores were rewritten
// after the caller's filter ran, so the filter must run again.
int settle_search(swa_db* db, Pending& pd, const uint8_t* q1, const uint8_t* q2, int64_t qlen, int64_t bound_min,
                  bool* again, bool* changed, int64_t qlen_a = 0, int64_t qlen_b = 0)
{
  *again = *changed = false;
  hipStream_t st = db->stream;
  if (pd.empty) return SWA_OK;
  if (pd.c.loading_parts > 0 && (uint32_t(ctl_host(db)[CTL_LOADFLAGS]) & ~0x1Fu))
    return fail(SWA_EINVAL, "database load failed: database residue code out of range (must be < 32)");
  const int64_t n1 = pd.dev1 ? ctl_host(db)[1] : 0, n2 = pd.dev2 ? ctl_host(db)[3] : 0;
  if (pd.dev1 || pd.dev2) {
    const int64_t lists = pd.dev2 ? 2 : 1;
    if (pd.used_bound && (n1 + n2) * 50 > lists * db->nseq && db->opt.bound != 1) {
      db->note_bound_off(qlen, bound_min);
      *again = true;
      return SWA_OK;
    }
    // lists beyond the device-driven kernel's reach (it stopped at REQUEUE_CAP entries): the host takes them whole
    for (int which = 0; which < 2; ++which) {
      const int64_t n = which ? n2 : n1;
      if (n <= REQUEUE_CAP) { pd.c.wide += n; continue; }
      std::vector<int32_t> list(static_cast<size_t>(n));
      HIP_TRY(hipMemcpy(list.data(), which ? db->ovf_list2.p : db->ovf_list.p, list.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
      std::sort(list.begin(), list.end());
      int64_t full = 0;
      const int64_t ql = which ? (qlen_b ? qlen_b : qlen) : (qlen_a ? qlen_a : qlen);
      const int rc = run_wide(db, list, which ? db->qseq2_p : db->qseq_p, ql, which ? db->scores2.p : db->scores.p,
                              which ? db->scores64b : db->scores64, &pd.c.wide, &full, st);
      if (rc != SWA_OK) return rc;
      pd.c.full += full;
      HIP_TRY(swa_launch_mark_excluded(which ? db->scores2.p : db->scores.p, db->excluded.p, int(db->n_excluded), st));
      *changed = true;
    }
    if (*changed) {
      if (pd.windows) { const int rc = fold_windows(db, pd.two, st); if (rc != SWA_OK) return rc; }
      HIP_TRY(hipEventRecord(db->ev[3], st));
      HIP_TRY(hipStreamSynchronize(st));
    }
  }
  float ms = 0;
  HIP_TRY(hipEventElapsedTime(&ms, db->ev[1], db->ev[2]));
  pd.c.kernel_ms = ms;
  HIP_TRY(hipEventElapsedTime(&ms, db->ev[0], db->ev[3]));
  pd.c.total_ms = ms;
  (void)q1; (void)q2;
  return SWA_OK;
}

// all scores of one / two queries on the device, host informed: the body of swa_search / swa_search2
int search_all(swa_db* db, const uint8_t* q1, const uint8_t* q2, int64_t qlen, swa_counters_t* counters)
{
  for (;;) {
    Pending pd;
    int rc = q2 ? run_search2(db, q1, q2, qlen, 0, pd) : run_search(db, q1, qlen, 0, pd);
    if (rc != SWA_OK) return rc;
    rc = sync_ctl(db, 0, db->stream);
    if (rc != SWA_OK) return rc;
    bool again = false, changed = false;
    rc = settle_search(db, pd, q1, q2, qlen, 0, &again, &changed);
    if (rc != SWA_OK) return rc;
    if (again) continue;
    if (counters) *counters = pd.c;
    return SWA_OK;
  }
}

}  // namespace

extern "C" const char* swa_last_error(void) { return swa::g_last_error.c_str(); }

extern "C" int swa_device_count(void)
{
  try {                                              // a COUNT, never a status: anything unexpected is "no device"
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n < 0 ? 0 : n;
  } catch (...) { return 0; }
}

// all guard regions of all live device allocations of this process (SWA_REDZONES=1): *buffers = allocations looked at,
// *touched = guard bytes that no longer hold the fill pattern; report (may be NULL) names the first few
extern "C" int swa_redzones_check(int64_t* buffers, int64_t* touched, char* report, int64_t report_cap)
try {
  if (buffers) *buffers = 0;
  if (touched) *touched = 0;
  if (report && report_cap > 0) report[0] = 0;
  if (!Redzones::on()) return fail(SWA_ESTATE, "red zones are off: set SWA_REDZONES=1 before the library is first used");
  HIP_TRY(hipDeviceSynchronize());
  Redzones& r = Redzones::get();
  std::lock_guard<std::mutex> g(r.mu);
  std::vector<unsigned char> host(2 * REDZONE);
  std::string text;
  int64_t bad = 0, listed = 0;
  for (const Redzones::Entry& e : r.live) {
    HIP_TRY(hipMemcpy(host.data(), e.raw, REDZONE, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(host.data() + REDZONE, e.raw + REDZONE + e.bytes, REDZONE, hipMemcpyDeviceToHost));
    int64_t before = 0, after = 0, first_after = -1, last_before = -1;
    for (size_t i = 0; i < REDZONE; ++i) {
      if (host[i] != 0xA5) { ++before; last_before = int64_t(i); }
      if (host[REDZONE + i] != 0xA5) { ++after; if (first_after < 0) first_after = int64_t(i); }
    }
    bad += before + after;
    if ((before || after) && listed < 8) {
      ++listed;
      text += "allocation of " + std::to_string(e.bytes) + " bytes (elements of " + std::to_string(e.elem) + "): " + std::to_string(before) +
              " guard bytes written in front (nearest " + std::to_string(last_before < 0 ? 0 : int64_t(REDZONE) - last_before) + " bytes before the start), " +
              std::to_string(after) + " behind (first at +" + std::to_string(first_after < 0 ? 0 : first_after) + " past the end); ";
    }
  }
  if (buffers) *buffers = int64_t(r.live.size());
  if (touched) *touched = bad;
  if (report && report_cap > 0) { std::strncpy(report, text.c_str(), size_t(report_cap) - 1); report[report_cap - 1] = 0; }
  return SWA_OK;
} SWA_CATCH

extern "C" int swa_db_from_memory(const uint8_t* residues, const int64_t* offsets, int64_t nseq, int symtype,
                                  int device, int64_t first_seqno, int64_t total_seqcount,
                                  int64_t total_symcount, swa_db** out)
try {
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
  options_from_environment(db->opt);
  db->first_seqno = first_seqno;
  const int rc = ingest(db, residues, offsets, nseq);
  if (rc != SWA_OK) { delete db; return rc; }
  db->total_seq = total_seqcount > 0 ? total_seqcount : db->nseq;
  db->total_sym = total_symcount > 0 ? total_symcount : db->nsym;
  *out = db;
  return SWA_OK;
} SWA_CATCH

// the old reader: everything into host memory (blastdb.cpp read_blast_db), then one upload.  What swa_db_open falls back
// to for the databases the pipelined loader leaves alone (nucleotide volumes, masked aliases, irregular indexes)
static int open_resident(const char* basename, int symtype, int device, int64_t first_seqno, int64_t last_seqno, swa_db** out)
{
  if (!out) return fail(SWA_EINVAL, "null output handle");
  *out = nullptr;
  swa::HostDb h;
  int rc = swa::read_blast_db(basename, symtype, first_seqno, last_seqno, h);
  if (rc != SWA_OK) return rc;
  // a masked alias searches only its members and computes statistics on its own NSEQ / LENGTH (hits.cc:333-342)
  rc = swa_db_from_memory(h.residues.data(), h.offsets.data(), int64_t(h.offsets.size()) - 1, symtype, device,
                          h.first_seqno, h.masked ? h.masked_seqcount : h.total_seqcount,
                          h.masked ? h.masked_symcount : h.total_symcount, out);
  if (rc == SWA_OK && h.masked) {
    rc = swa_db_set_inclusion(*out, h.included.data(), int64_t(h.included.size()));
    if (rc != SWA_OK) { swa_db_close(*out); *out = nullptr; }
  }
  return rc;
}

namespace {
// a nucleotide shard as its six translations.  The nucleotide residues come from the host (one base per byte, at offsets[0])
// or are on the device already (dev_nt: the 4-bit form a pipelined open leaves there, offsets counted from 0)
int translated_shard(const uint8_t* nt_residues, const uint8_t* dev_nt, const int64_t* offsets, int64_t nseq, int db_gencode, int device,
                     int64_t first_seqno, int64_t total_seqcount, int64_t total_symcount, swa_db** out)
{
  if (!out) return fail(SWA_EINVAL, "null output handle");
  *out = nullptr;
  if (nseq < 0 || !offsets || (!nt_residues && !dev_nt && nseq > 0 && offsets[nseq] > offsets[0]))
    return fail(SWA_EINVAL, "bad database arrays");
  if (nseq > 0x7ffffff0 / 6) return fail(SWA_EINVAL, "too many sequences for one translated shard; shard the database");
  uint8_t table[4096];
  int rc = swa_translate_table(db_gencode, table);
  if (rc != SWA_OK) return fail(SWA_EINVAL, "Illegal database genetic code specified.");
  if (device < 0 || device >= swa_device_count())
    return fail(SWA_ENODEV, "no such HIP device (swipe_amd has no CPU fallback)");
  swa_db* db = new (std::nothrow) swa_db;
  if (!db) return fail(SWA_ENOMEM, "out of host memory");
  struct Guard { swa_db* d; ~Guard() { delete d; } } guard{db};
  db->device = device;
  db->symtype = SWA_SYMTYPE_PROTEIN;
  db->frames = 6;
  options_from_environment(db->opt);
  db->first_seqno = first_seqno;
  // virtual offsets: frame f of either strand holds (len - f) / 3 residues (database.cc:1188)
  const int64_t base = offsets[0];
  std::vector<int64_t> ntoff(size_t(nseq) + 1), voff(size_t(6 * nseq) + 1);
  db->h_ntlen.resize(size_t(nseq));
  voff[0] = 0;
  for (int64_t s = 0; s < nseq; ++s) {
    const int64_t len = offsets[s + 1] - offsets[s];
    if (len < 0) return fail(SWA_EINVAL, "sequence offsets must be non-decreasing");
    ntoff[size_t(s)] = offsets[s] - base;
    db->h_ntlen[size_t(s)] = len;
    db->nt_longest = std::max(db->nt_longest, len);
    for (int t = 0; t < 6; ++t) {
      const int64_t plen = len - t % 3 > 0 ? (len - t % 3) / 3 : 0;
      voff[size_t(6 * s + t) + 1] = voff[size_t(6 * s + t)] + plen;
    }
  }
  ntoff[size_t(nseq)] = offsets[nseq] - base;
  db->nt_sym = ntoff[size_t(nseq)];
  const int64_t total = voff[size_t(6 * nseq)];
  HIP_TRY(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  db->cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  HIP_TRY(hipStreamCreate(&db->stream));
  for (hipEvent_t& e : db->ev) HIP_TRY(hipEventCreate(&e));
  {
    // the nucleotide form lives on the device only for the duration of the translation pre-pass
    DevBuf<uint8_t> d_nt, d_table;
    DevBuf<int64_t> d_ntoff, d_voff;
    if (!dev_nt) HIP_TRY(d_nt.reserve(size_t(db->nt_sym) + 16));
    HIP_TRY(d_table.reserve(4096));
    HIP_TRY(d_ntoff.reserve(ntoff.size()));
    HIP_TRY(d_voff.reserve(voff.size()));
    HIP_TRY(db->residues.reserve(size_t(total) + 16));
    if (db->nt_sym && !dev_nt) HIP_TRY(hipMemcpyAsync(d_nt.p, nt_residues + base, size_t(db->nt_sym), hipMemcpyHostToDevice, db->stream));
    HIP_TRY(hipMemcpyAsync(d_table.p, table, 4096, hipMemcpyHostToDevice, db->stream));
    HIP_TRY(hipMemcpyAsync(d_ntoff.p, ntoff.data(), ntoff.size() * sizeof(int64_t), hipMemcpyHostToDevice, db->stream));
    HIP_TRY(hipMemcpyAsync(d_voff.p, voff.data(), voff.size() * sizeof(int64_t), hipMemcpyHostToDevice, db->stream));
    HIP_TRY(swa_launch_translate(dev_nt ? dev_nt : d_nt.p, dev_nt ? 1 : 0, d_ntoff.p, d_voff.p, 6 * nseq, d_table.p, db->residues.p, total, db->stream));
    HIP_TRY(hipStreamSynchronize(db->stream));
  }
  rc = ingest(db, nullptr, voff.data(), 6 * nseq);
  if (rc != SWA_OK) return rc;
  db->total_seq = total_seqcount > 0 ? total_seqcount : nseq;
  db->total_sym = total_symcount > 0 ? total_symcount : db->nt_sym;
  guard.d = nullptr;
  *out = db;
  return SWA_OK;
}
}  // namespace

extern "C" int swa_db_from_memory_translated(const uint8_t* nt_residues, const int64_t* offsets, int64_t nseq,
                                             int db_gencode, int device, int64_t first_seqno,
                                             int64_t total_seqcount, int64_t total_symcount, swa_db** out)
try {
  return translated_shard(nt_residues, nullptr, offsets, nseq, db_gencode, device, first_seqno, total_seqcount, total_symcount, out);
} SWA_CATCH

extern "C" int swa_db_open_translated(const char* basename, int db_gencode, int device, int64_t first_seqno,
                                      int64_t last_seqno, swa_db** out)
try {
  if (!out) return fail(SWA_EINVAL, "null output handle");
  *out = nullptr;
  {
    // the nucleotide volumes through the pipelined open (round 5), translated out of the 4-bit residues it leaves on the
    // device; a masked alias too (round 6): its OID mask is taken off the loader before the nucleotide handle adopts it and
    // becomes the translated shard's inclusion set (six frames per member), its NSEQ / LENGTH the totals of the statistics
    uint8_t table[4096];
    if (swa_translate_table(db_gencode, table) != SWA_OK) return fail(SWA_EINVAL, "Illegal database genetic code specified.");
    swa_db* nt = nullptr;
    int prc = open_pipelined(basename, SWA_SYMTYPE_NUCLEOTIDE, device, first_seqno, last_seqno, &nt);
    if (prc != SWA_OK) return prc;
    if (nt) {
      std::vector<uint8_t> mask;
      if (nt->loading && nt->loading->masked) mask = nt->loading->included;
      prc = settle_loading(nt, true, nullptr, true);
      if (prc == SWA_OK)
        prc = translated_shard(nullptr, nt->residues.p, nt->h_offsets.data(), nt->nseq, db_gencode, device, nt->first_seqno, nt->total_seq,
                               nt->total_sym, out);
      swa_db_close(nt);
      if (prc == SWA_OK && !mask.empty()) {
        prc = swa_db_set_inclusion(*out, mask.data(), int64_t(mask.size()));
        if (prc != SWA_OK) { swa_db_close(*out); *out = nullptr; }
      }
      return prc;
    }
  }
  swa::HostDb h;
  int rc = swa::read_blast_db(basename, SWA_SYMTYPE_NUCLEOTIDE, first_seqno, last_seqno, h);
  if (rc != SWA_OK) return rc;
  rc = swa_db_from_memory_translated(h.residues.data(), h.offsets.data(), int64_t(h.offsets.size()) - 1, db_gencode,
                                     device, h.first_seqno, h.masked ? h.masked_seqcount : h.total_seqcount,
                                     h.masked ? h.masked_symcount : h.total_symcount, out);
  if (rc == SWA_OK && h.masked) {
    rc = swa_db_set_inclusion(*out, h.included.data(), int64_t(h.included.size()));
    if (rc != SWA_OK) { swa_db_close(*out); *out = nullptr; }
  }
  return rc;
} SWA_CATCH

extern "C" int swa_blastdb_read(const char* basename, int symtype, int64_t first_seqno, int64_t last_seqno,
                                uint8_t** residues, int64_t** offsets, int64_t* nseq, int64_t* total_seqcount,
                                int64_t* total_symcount, int64_t* longest)
try {
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
} SWA_CATCH

extern "C" void swa_free(void* p) { std::free(p); }

extern "C" int swa_blastdb_defline(const char* basename, int symtype, int64_t seqno, char* buf, int64_t buflen, int64_t* seqlen)
try {
  if (!basename || !buf || buflen < 1) return fail(SWA_EINVAL, "bad argument");
  std::vector<std::string> d;
  std::vector<int64_t> len;
  const int rc = swa::read_blast_deflines(basename, symtype, std::vector<int64_t>{seqno}, d, len);
  if (rc != SWA_OK) return rc;
  std::snprintf(buf, size_t(buflen), "%s", d[0].substr(0, d[0].find('\n')).c_str());
  if (seqlen) *seqlen = len[0];
  return SWA_OK;
} SWA_CATCH

extern "C" int swa_blastdb_deflines(const char* basename, int symtype, int64_t seqno, char* buf, int64_t buflen, int64_t* needed)
try {
  if (!basename || buflen < 0 || (buflen > 0 && !buf) || !needed) return fail(SWA_EINVAL, "bad argument");
  std::vector<std::string> d;
  std::vector<int64_t> len;
  const int rc = swa::read_blast_deflines(basename, symtype, std::vector<int64_t>{seqno}, d, len);
  if (rc != SWA_OK) return rc;
  *needed = int64_t(d[0].size()) + 1;
  if (*needed > buflen) return fail(SWA_ERANGE, "defline buffer too small");
  std::memcpy(buf, d[0].c_str(), d[0].size() + 1);
  return SWA_OK;
} SWA_CATCH

extern "C" int swa_db_info(const swa_db* db, swa_db_info_t* info)
try {
  if (!db || !info) return fail(SWA_EINVAL, "null argument");
  info->seqcount = db->nseq / db->frames;
  info->symcount = db->frames == 1 ? db->nsym : db->nt_sym;
  info->longest = db->frames == 1 ? db->longest : db->nt_longest;
  info->frames = db->frames;
  info->first_seqno = db->first_seqno;
  info->total_seqcount = db->total_seq;
  info->total_symcount = db->total_sym;
  info->hbm_bytes = int64_t(db->streamed ? streamed_hbm(db) : db->hbm_bytes() + loading_hbm(db));
  return SWA_OK;
} SWA_CATCH

extern "C" int swa_db_set_inclusion(swa_db* db, const uint8_t* include, int64_t n)
try {
  if (!db) return fail(SWA_EINVAL, "null database handle");
  if (db->streamed) return streamed_set_inclusion(db, include, n);     // a budgeted shard: per part (sw_streamed.inc)
  { const int src_ = loaded(db); if (src_ != SWA_OK) return src_; }
  return apply_inclusion(db, include, n);
} SWA_CATCH

namespace {
int apply_inclusion(swa_db* db, const uint8_t* include, int64_t n, bool release_now)
{
  const int64_t real = db->nseq / db->frames;
  if (include && n != real) return fail(SWA_EINVAL, "inclusion array must have one entry per sequence of the shard");
  HIP_TRY(hipSetDevice(db->device));
  if (release_now) release_loader_leftovers(db);      // not from a search that adopts a masked shard on its way (sw_loading.inc)
  std::vector<int32_t> in, ex;
  db->active_sym = 0;
  for (int64_t v = 0; v < db->nseq; ++v) {
    if (!include || include[v / db->frames]) {
      in.push_back(int32_t(v));
      db->active_sym += db->h_offsets[size_t(v) + 1] - db->h_offsets[size_t(v)];
    } else {
      ex.push_back(int32_t(v));
    }
  }
  order_by_length([&](int32_t id) { return db->len_of(id); }, in.data(), int64_t(in.size()), db->h_order);
  db->n_excluded = int64_t(ex.size());
  HIP_TRY(db->excluded.reserve(ex.size()));
  if (!ex.empty()) HIP_TRY(hipMemcpy(db->excluded.p, ex.data(), ex.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  db->main.built = db->single.built = db->single4.built = db->view.built = false;
  db->view_of = nullptr;
  db->nwin = 0;
  return db->packed ? SWA_OK : ensure_main(db);
}
}  // namespace

extern "C" void swa_db_close(swa_db* db)
{
  if (!db) return;
  (void)hipSetDevice(db->device);
  delete db;
}

extern "C" int swa_set_scoring(swa_db* db, const int64_t* matrix, int64_t gapopenextend, int64_t gapextend)
try {
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
  if (db->streamed) {
    db->scoring_set = true;
    return streamed_set_scoring(db, matrix, gapopenextend, gapextend);
  }
  HIP_TRY(hipSetDevice(db->device));
  release_loader_leftovers(db);                      // (a no-op unless a streamed-in shard's tables were adopted by a search)
  HIP_TRY(hipMemcpyAsync(db->matrix.p, db->h_matrix, sizeof db->h_matrix, hipMemcpyHostToDevice, db->stream));
  HIP_TRY(hipStreamSynchronize(db->stream));
  db->scoring_set = true;
  db->bound_off.clear();
  return SWA_OK;
} SWA_CATCH

extern "C" int swa_search(swa_db* db, const uint8_t* query, int64_t qlen, int64_t* scores, swa_counters_t* counters);

namespace {
struct Cand { int64_t seqno, score; int32_t which, dtag; };

// hits_enter acceptance test (hits.cc:174-184) over one / two score arrays: ENQUEUES the counter reset and the filter
// kernel(s); candidates of both arrays land in the same record list, tagged 0 / 1
int enqueue_filter(swa_db* db, bool two, int64_t minscore, int64_t maxscore, hipStream_t st, const Pair* pair = nullptr)
{
  HIP_TRY(hipMemsetAsync(db->ctl.p + CTL_CAND, 0, (16 - CTL_CAND) * sizeof(int32_t), st));
  unsigned long long* tallies = reinterpret_cast<unsigned long long*>(db->ctl.p + CTL_TALLY);
  HIP_TRY(swa_launch_filter(db->scores.p, db->scores64.p, int(db->nseq), 0, minscore, maxscore, db->ctl.p + CTL_CAND,
                            db->cand_cap, cand_dev(db), tallies, st));
  if (two && pair && pair->own_window) {                 // a different query: its own thresholds and its own counts
    unsigned long long* tallies_b = reinterpret_cast<unsigned long long*>(db->ctl.p + CTL_TALLY_B);
    HIP_TRY(hipMemsetAsync(tallies_b, 0, 2 * sizeof(unsigned long long), st));
    HIP_TRY(swa_launch_filter(db->scores2.p, db->scores64b.p, int(db->nseq), 1, pair->minscore_b, pair->maxscore_b,
                              db->ctl.p + CTL_CAND, db->cand_cap, cand_dev(db), tallies_b, st));
  } else if (two) {
    HIP_TRY(swa_launch_filter(db->scores2.p, db->scores64b.p, int(db->nseq), 1, minscore, maxscore, db->ctl.p + CTL_CAND,
                              db->cand_cap, cand_dev(db), tallies, st));
  }
  return SWA_OK;
}

// more candidates than the compaction buffer (a very permissive threshold): take every score to the host, find
// the score of the keep-th best accepted entry from a histogram and keep only what can still make the list
int candidates_by_histogram(swa_db* db, const int32_t* scores, const long long* scores64, int32_t which, int32_t tag,
                            int64_t keep, int64_t minscore, int64_t maxscore, std::vector<Cand>& cand)
{
  std::vector<int32_t> s32(size_t(db->nseq));
  HIP_TRY(hipMemcpy(s32.data(), scores, s32.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
  std::vector<long long> s64;
  auto value = [&](int64_t i) -> int64_t {
    const int64_t v = s32[size_t(i)];
    return v == SWA_SCORE_IN_64 ? int64_t(s64[size_t(i)]) : v;
  };
  for (int64_t i = 0; i < db->nseq && s64.empty(); ++i)
    if (s32[size_t(i)] == SWA_SCORE_IN_64) {
      s64.resize(size_t(db->nseq));
      HIP_TRY(hipMemcpy(s64.data(), scores64, s64.size() * sizeof(long long), hipMemcpyDeviceToHost));
    }
  constexpr int64_t BINS = 1 << 16;
  std::vector<int64_t> hist(size_t(BINS), 0);
  for (int64_t i = 0; i < db->nseq; ++i) {
    const int64_t v = value(i);
    if (v >= minscore && v <= maxscore) ++hist[size_t(std::min<int64_t>(std::max<int64_t>(v, 0), BINS - 1))];
  }
  int64_t floor_score = minscore, seen = 0;
  for (int64_t b = BINS - 1; b >= 0; --b) {
    seen += hist[size_t(b)];
    if (seen >= keep) { floor_score = std::max(minscore, b == BINS - 1 ? minscore : b); break; }
  }
  (void)which;
  for (int64_t i = 0; i < db->nseq; ++i) {
    const int64_t v = value(i);
    if (v >= floor_score && v <= maxscore) cand.push_back({db->first_seqno + i / db->frames, v, tag, int32_t(i % db->frames)});
  }
  return SWA_OK;
}

// after sync_ctl(db, CAND_EAGER): the filter's results out of the page-locked block (tag0 / tag1 = the `which` value
// the caller wants on candidates of the first / second score array)
int gather_candidates(swa_db* db, bool two, int32_t tag0, int32_t tag1, int64_t keep, int64_t minscore, int64_t maxscore,
                      std::vector<Cand>& cand, int64_t* totalhits, int64_t* obvious, Pair* pair = nullptr)
{
  const int32_t* h = ctl_host(db);
  const int64_t ncand = h[CTL_CAND];
  unsigned long long tl[2];
  std::memcpy(tl, h + CTL_TALLY, sizeof tl);
  *totalhits += int64_t(tl[0]);
  *obvious += int64_t(tl[1]);
  if (pair && pair->own_window) {
    std::memcpy(tl, h + CTL_TALLY_B, sizeof tl);
    pair->total_b += int64_t(tl[0]);                     // += : a streamed database calls this once per part
    pair->obvious_b += int64_t(tl[1]);
  }
  if (ncand <= db->cand_cap) {
    const swa_cand* rec = reinterpret_cast<const swa_cand*>(db->pin + CTL_INTS * sizeof(int32_t));
    std::vector<swa_cand> rest;
    if (ncand > CAND_EAGER) {                              // beyond what came back with the counters
      rest.resize(size_t(ncand));
      HIP_TRY(hipMemcpy(rest.data(), cand_dev(db), rest.size() * sizeof(swa_cand), hipMemcpyDeviceToHost));
      rec = rest.data();
    }
    cand.reserve(cand.size() + size_t(ncand));
    for (int64_t i = 0; i < ncand; ++i)
      cand.push_back({db->first_seqno + rec[i].idx / db->frames, rec[i].score, rec[i].which ? tag1 : tag0,
                      int32_t(rec[i].idx % db->frames)});
    return SWA_OK;
  }
  int rc = candidates_by_histogram(db, db->scores.p, db->scores64.p, 0, tag0, keep, minscore, maxscore, cand);
  if (rc == SWA_OK && two && pair && pair->own_window)
    rc = candidates_by_histogram(db, db->scores2.p, db->scores64b.p, 1, tag1, pair->keep_b, pair->minscore_b, pair->maxscore_b, cand);
  else if (rc == SWA_OK && two)
    rc = candidates_by_histogram(db, db->scores2.p, db->scores64b.p, 1, tag1, keep, minscore, maxscore, cand);
  return rc;
}

// hits.cc:188-190 (score desc, seqno desc); entries of query 1 were entered before those of
// query 2 (swipe.cc:1403 loops the strands in that order), so they win remaining ties
bool cand_before(const Cand& a, const Cand& b)
{
  if (a.score != b.score) return a.score > b.score;
  if (a.seqno != b.seqno) return a.seqno > b.seqno;
  if (a.which != b.which) return a.which < b.which;
  return a.dtag < b.dtag;      // frames of one sequence are entered in start_list order (swipe.cc:1379-1384)
}

// One top-K search of one / two queries: first pass, re-queues and the hit filter are enqueued back to back, ONE
// copy brings counters and candidates into page-locked memory, ONE synchronisation.  Candidates are appended to
// `cand` with `which` = tag0 / tag1.
int search_candidates(swa_db* db, const uint8_t* q1, const uint8_t* q2, int64_t qlen, int64_t keep, int64_t minscore,
                      int64_t maxscore, int32_t tag0, int32_t tag1, std::vector<Cand>& cand, int64_t* totalhits,
                      int64_t* obvious, swa_counters_t* counters, Pair* pair = nullptr)
{
  hipStream_t st = db ? db->stream : nullptr;
  const int64_t qa = pair ? pair->qlen_a : 0, qb = pair ? pair->qlen_b : 0;
  // one first pass serves both queries: it must keep whatever EITHER threshold wants
  const int64_t bound_min = pair && pair->own_window ? std::min(minscore, pair->minscore_b) : minscore;
  for (;;) {
    Pending pd;
    int rc = q2 ? run_search2(db, q1, q2, qlen, bound_min, pd, qa, qb) : run_search(db, q1, qlen, minscore, pd);
    if (rc != SWA_OK) return rc;
    if (db->nseq) {
      rc = enqueue_filter(db, q2 != nullptr, minscore, maxscore, st, pair);
      if (rc != SWA_OK) return rc;
    }
    rc = sync_ctl(db, db->nseq ? std::min(CAND_EAGER, db->cand_cap) : 0, st);
    if (rc != SWA_OK) return rc;
    bool again = false, changed = false;
    rc = settle_search(db, pd, q1, q2, qlen, bound_min, &again, &changed, qa, qb);
    if (rc != SWA_OK) return rc;
    if (again) continue;
    if (changed && db->nseq) {
      rc = enqueue_filter(db, q2 != nullptr, minscore, maxscore, st, pair);
      if (rc == SWA_OK) rc = sync_ctl(db, std::min(CAND_EAGER, db->cand_cap), st);
      if (rc != SWA_OK) return rc;
    }
    if (db->nseq) {
      rc = gather_candidates(db, q2 != nullptr, tag0, tag1, keep, minscore, maxscore, cand, totalhits, obvious, pair);
      if (rc != SWA_OK) return rc;
    }
    if (counters) *counters = pd.c;
    return SWA_OK;
  }
}

int download_scores(swa_db* db, const int32_t* dev, const DevBuf<long long>& dev64, int64_t* out)
{
  // the int32 scores land in a host buffer the handle keeps and are widened into the caller's int64 array by a few
  // threads on disjoint ranges
  const int64_t n = db->nseq;
  if (db->h_stage.size() < size_t(n)) db->h_stage.resize(size_t(n));
  int32_t* staged = db->h_stage.data();
  HIP_TRY(hipMemcpy(staged, dev, size_t(n) * sizeof(int32_t), hipMemcpyDeviceToHost));
  const int64_t nthreads = std::max<int64_t>(1, std::min<int64_t>({int64_t(std::thread::hardware_concurrency()), 16, n >> 18}));
  std::vector<uint8_t> wide(size_t(nthreads), 0);
  auto widen = [&](int64_t t) {
    const int64_t lo = n * t / nthreads, hi = n * (t + 1) / nthreads;
    bool any = false;
    for (int64_t i = lo; i < hi; ++i) {
      const int32_t v = staged[i];
      any |= v == SWA_SCORE_IN_64;
      out[i] = v;
    }
    wide[size_t(t)] = any;
  };
  if (nthreads == 1) {
    widen(0);
  } else {
    std::vector<std::thread> pool;
    for (int64_t t = 0; t < nthreads; ++t) pool.emplace_back(widen, t);
    for (std::thread& t : pool) t.join();
  }
  bool any = false;
  for (uint8_t w : wide) any |= w != 0;
  if (any) {                                               // scores beyond 32 bits live in the 64-bit array
    std::vector<long long> s64((size_t(n)));
    HIP_TRY(hipMemcpy(s64.data(), dev64.p, s64.size() * sizeof(long long), hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < n; ++i)
      if (out[i] == SWA_SCORE_IN_64) out[i] = s64[size_t(i)];
  }
  return SWA_OK;
}
}  // namespace

#include "sw_streamed.inc"

extern "C" int swa_search(swa_db* db, const uint8_t* query, int64_t qlen, int64_t* scores, swa_counters_t* counters)
try {
  if (db && db->streamed) {
    const int rc0 = check_query(db, query, qlen);
    return rc0 != SWA_OK ? rc0 : streamed_all_scores(db, query, qlen, scores, counters);
  }
  const int rc = search_all(db, query, nullptr, qlen, counters);
  if (rc != SWA_OK || !scores || db->nseq == 0) return rc;
  return download_scores(db, db->scores.p, db->scores64, scores);
} SWA_CATCH

extern "C" int swa_search_topk(swa_db* db, const uint8_t* query, int64_t qlen, int64_t keep, int64_t minscore,
                               int64_t maxscore, swa_hit_t* hits, int64_t* nhits, int64_t* totalhits,
                               int64_t* obvious, swa_counters_t* counters)
try {
  if (keep < 0 || (keep > 0 && !hits) || !nhits) return fail(SWA_EINVAL, "bad hit buffer");
  if (db && db->frames != 1) return fail(SWA_ESTATE, "translated shard: use swa_search_frames_topk");
  *nhits = 0;
  int64_t tot = 0, obv = 0;
  std::vector<Cand> cand;
  int rc = SWA_OK;
  if (db && db->streamed) {
    rc = check_query(db, query, qlen);
    if (rc == SWA_OK) rc = streamed_candidates(db, query, qlen, keep, minscore, maxscore, cand, &tot, &obv, counters);
  } else {
    rc = search_candidates(db, query, nullptr, qlen, keep, minscore, maxscore, 0, 0, cand, &tot, &obv, counters);
  }
  if (rc != SWA_OK) return rc;
  if (totalhits) *totalhits = tot;
  if (obvious) *obvious = obv;
  const size_t k = std::min<size_t>(size_t(keep), cand.size());
  std::partial_sort(cand.begin(), cand.begin() + k, cand.end(), cand_before);
  for (size_t i = 0; i < k; ++i) hits[i] = {cand[i].seqno, cand[i].score};
  *nhits = int64_t(k);
  return SWA_OK;
} SWA_CATCH

extern "C" int swa_search2(swa_db* db, const uint8_t* query1, const uint8_t* query2, int64_t qlen,
                           int64_t* scores1, int64_t* scores2, swa_counters_t* counters)
try {
  if (!db) return fail(SWA_EINVAL, "null database handle");
  if (!query2 && qlen > 0) return fail(SWA_EINVAL, "bad query");
  if (db && db->streamed) {
    int rc0 = check_query(db, query1, qlen);
    if (rc0 == SWA_OK) rc0 = check_query(db, query2 ? query2 : query1, qlen);
    return rc0 != SWA_OK ? rc0 : streamed_all_scores(db, query1, qlen, scores1, counters, query2 ? query2 : query1, scores2);
  }
  int rc = search_all(db, query1, query2 ? query2 : query1, qlen, counters);
  if (rc != SWA_OK || db->nseq == 0) return rc;
  if (scores1) rc = download_scores(db, db->scores.p, db->scores64, scores1);
  if (rc == SWA_OK && scores2) rc = download_scores(db, db->scores2.p, db->scores64b, scores2);
  return rc;
} SWA_CATCH

extern "C" int swa_search2_topk(swa_db* db, const uint8_t* query1, const uint8_t* query2, int64_t qlen, int64_t keep,
                                int64_t minscore, int64_t maxscore, swa_hit_t* hits, int32_t* which, int64_t* nhits,
                                int64_t* totalhits, int64_t* obvious, swa_counters_t* counters)
try {
  if (keep < 0 || (keep > 0 && (!hits || !which)) || !nhits) return fail(SWA_EINVAL, "bad hit buffer");
  if (db && db->frames != 1) return fail(SWA_ESTATE, "translated shard: use swa_search_frames_topk");
  if (!query2 && qlen > 0) return fail(SWA_EINVAL, "bad query");
  *nhits = 0;
  int64_t tot = 0, obv = 0;
  std::vector<Cand> cand;
  int rc = SWA_OK;
  if (db && db->streamed) {
    rc = check_query(db, query1, qlen);
    if (rc == SWA_OK) rc = check_query(db, query2, qlen);
    if (rc == SWA_OK) rc = streamed_candidates(db, query1, qlen, keep, minscore, maxscore, cand, &tot, &obv, counters, query2, 1, nullptr);
  } else {
    rc = search_candidates(db, query1, query2 ? query2 : query1, qlen, keep, minscore, maxscore, 0, 1, cand, &tot, &obv, counters);
  }
  if (rc != SWA_OK) return rc;
  if (totalhits) *totalhits = tot;
  if (obvious) *obvious = obv;
  const size_t k = std::min<size_t>(size_t(keep), cand.size());
  std::partial_sort(cand.begin(), cand.begin() + k, cand.end(), cand_before);
  for (size_t i = 0; i < k; ++i) { hits[i] = {cand[i].seqno, cand[i].score}; which[i] = cand[i].which; }
  *nhits = int64_t(k);
  return SWA_OK;
} SWA_CATCH

// Two different queries of a multi-query file in one pass (swipe.cc:2561-2575: the reference's unit of work is a query
// FILE): each query keeps its own hit list, thresholds and counts; results equal two swa_search_topk calls.
extern "C" int swa_search_pair_topk(swa_db* db, const uint8_t* query1, int64_t qlen1, const uint8_t* query2, int64_t qlen2,
                                    int64_t keep1, int64_t minscore1, int64_t maxscore1, int64_t keep2, int64_t minscore2,
                                    int64_t maxscore2, swa_hit_t* hits1, int64_t* nhits1, int64_t* totalhits1, int64_t* obvious1,
                                    swa_hit_t* hits2, int64_t* nhits2, int64_t* totalhits2, int64_t* obvious2,
                                    swa_counters_t* counters)
try {
  if (keep1 < 0 || keep2 < 0 || (keep1 > 0 && !hits1) || (keep2 > 0 && !hits2) || !nhits1 || !nhits2)
    return fail(SWA_EINVAL, "bad hit buffer");
  if (db && db->frames != 1) return fail(SWA_ESTATE, "translated shard: use swa_search_frames_topk");
  if (qlen1 <= 0 || qlen2 <= 0) return fail(SWA_EINVAL, "swa_search_pair_topk needs two non-empty queries");
  *nhits1 = *nhits2 = 0;
  Pair pair;
  pair.qlen_a = qlen1;
  pair.qlen_b = qlen2;
  pair.own_window = true;
  pair.minscore_b = minscore2;
  pair.maxscore_b = maxscore2;
  pair.keep_b = keep2;
  int64_t tot = 0, obv = 0;
  std::vector<Cand> cand;
  int rc = SWA_OK;
  if (db && db->streamed) {
    rc = check_query(db, query1, qlen1);
    if (rc == SWA_OK) rc = check_query(db, query2, qlen2);
    if (rc == SWA_OK) rc = streamed_candidates(db, query1, std::max(qlen1, qlen2), keep1, minscore1, maxscore1, cand, &tot, &obv, counters,
                                               query2, 1, &pair);
  } else {
    rc = search_candidates(db, query1, query2, std::max(qlen1, qlen2), keep1, minscore1, maxscore1, 0, 1, cand, &tot, &obv, counters, &pair);
  }
  if (rc != SWA_OK) return rc;
  if (totalhits1) *totalhits1 = tot;
  if (obvious1) *obvious1 = obv;
  if (totalhits2) *totalhits2 = pair.total_b;
  if (obvious2) *obvious2 = pair.obvious_b;
  std::vector<Cand> c1, c2;
  for (const Cand& c : cand) (c.which ? c2 : c1).push_back(c);
  const size_t k1 = std::min<size_t>(size_t(keep1), c1.size()), k2 = std::min<size_t>(size_t(keep2), c2.size());
  std::partial_sort(c1.begin(), c1.begin() + k1, c1.end(), cand_before);
  std::partial_sort(c2.begin(), c2.begin() + k2, c2.end(), cand_before);
  for (size_t i = 0; i < k1; ++i) hits1[i] = {c1[i].seqno, c1[i].score};
  for (size_t i = 0; i < k2; ++i) hits2[i] = {c2[i].seqno, c2[i].score};
  *nhits1 = int64_t(k1);
  *nhits2 = int64_t(k2);
  return SWA_OK;
} SWA_CATCH

extern "C" int swa_search_frames_topk(swa_db* db, int nq, const uint8_t* const* queries, const int64_t* qlens,
                                      const int32_t* qtags, int64_t keep, int64_t minscore, int64_t maxscore,
                                      swa_fhit_t* hits, int64_t* nhits, int64_t* totalhits, int64_t* obvious,
                                      swa_counters_t* counters)
try {
  if (!db) return fail(SWA_EINVAL, "null database handle");
  if (nq < 1 || nq > 6 || !queries || !qlens) return fail(SWA_EINVAL, "between 1 and 6 query frames expected");
  if (keep < 0 || (keep > 0 && !hits) || !nhits) return fail(SWA_EINVAL, "bad hit buffer");
  *nhits = 0;
  int64_t tot = 0, obv = 0;
  std::vector<Cand> cand;
  swa_counters_t sum{};
  // one pass over the shard per query frame, in the order search_chunk loops them (swipe.cc:1403-1404);
  // frames of equal length share a pass in the two halves of the packed lanes
  for (int i = 0; i < nq;) {
    swa_counters_t c{};
    const bool pair = i + 1 < nq && qlens[i] == qlens[i + 1] && qlens[i] > 0;
    int rc = SWA_OK;
    if (db->streamed) {                                   // a shard over its HBM budget: the same pass, part by part
      rc = check_query(db, queries[i], qlens[i]);
      if (rc == SWA_OK && pair) rc = check_query(db, queries[i + 1], qlens[i]);
      if (rc == SWA_OK) rc = streamed_candidates(db, queries[i], qlens[i], keep, minscore, maxscore, cand, &tot, &obv, &c,
                                                 pair ? queries[i + 1] : nullptr, i + 1, nullptr, i);
    } else {
      rc = search_candidates(db, queries[i], pair ? queries[i + 1] : nullptr, qlens[i], keep, minscore, maxscore, i,
                             i + 1, cand, &tot, &obv, &c);
    }
    if (rc != SWA_OK) return rc;
    sum.narrow += c.narrow; sum.wide += c.wide; sum.full += c.full; sum.cells += c.cells;
    sum.kernel_ms += c.kernel_ms; sum.total_ms += c.total_ms;
    sum.narrow_rows = c.narrow_rows; sum.narrow_shifted = c.narrow_shifted;
    i += pair ? 2 : 1;
  }
  if (counters) *counters = sum;
  if (totalhits) *totalhits = tot;
  if (obvious) *obvious = obv;
  const size_t k = std::min<size_t>(size_t(keep), cand.size());
  std::partial_sort(cand.begin(), cand.begin() + k, cand.end(), cand_before);
  for (size_t i = 0; i < k; ++i) {
    const int tag = qtags ? qtags[cand[i].which] : 0;
    hits[i] = {cand[i].seqno, cand[i].score, tag / 3, tag % 3, cand[i].dtag / 3, cand[i].dtag % 3};
  }
  *nhits = int64_t(k);
  return SWA_OK;
} SWA_CATCH

extern "C" int swa_set_option(swa_db* db, const char* key, const char* value)
try {
  if (!db || !key) return fail(SWA_EINVAL, "null argument");
  for (const OptionKey& k : kOptionKeys)
    if (!std::strcmp(k.key, key)) {
      int64_t v = 0;
      if (value && !parse_option_value(key, value, &v)) return fail(SWA_EINVAL, std::string("bad value for option ") + key);
      if (!value) v = Options{}.*(k.field);                                          // NULL: back to the default
      db->opt.*(k.field) = v;
      if (db->streamed) for (swa_db* slot : db->streamed->slot) slot->opt.*(k.field) = v;
      return SWA_OK;
    }
  return fail(SWA_EINVAL, std::string("unknown option ") + key);
} SWA_CATCH

namespace {
// index of (seqno, dstrand, dframe) among the sequences the shard holds; *minus = reverse-complement on the fly
int locate(const swa_db* db, int64_t seqno, int dstrand, int dframe, int64_t* local, bool* minus)
{
  const int64_t real = seqno - db->first_seqno;
  if (real < 0 || real >= db->nseq / db->frames) return fail(SWA_EINVAL, "sequence number outside this shard");
  if (dstrand < 0 || dstrand > 1 || dframe < 0 || dframe > 2) return fail(SWA_EINVAL, "database strand / frame out of range");
  *minus = false;
  if (db->frames == 6) { *local = 6 * real + 3 * dstrand + dframe; return SWA_OK; }
  if (dframe) return fail(SWA_EINVAL, "database frames need a translated shard");
  if (dstrand) {
    if (db->symtype != SWA_SYMTYPE_NUCLEOTIDE) return fail(SWA_EINVAL, "database strand 1 needs a nucleotide database");
    *minus = true;
  }
  *local = real;
  return SWA_OK;
}

// search16s over the listed (sequence, strand, frame) triples: out = scores | bestpos | bestq, n entries each
int endpoints_on_device(swa_db* db, const uint8_t* query, int64_t qlen, const int64_t* seqnos, const int32_t* dstrands,
                        const int32_t* dframes, int64_t n, std::vector<long long>& out)
{
  if (n > (1 << 20)) return fail(SWA_EINVAL, "too many sequences for the alignment phase");
  std::vector<int32_t> ids((size_t(n)));
  std::vector<uint8_t> minus((size_t(n)), 0);
  for (int64_t i = 0; i < n; ++i) {
    int64_t local = 0;
    bool rc_flag = false;
    const int rc = locate(db, seqnos[i], dstrands ? dstrands[i] : 0, dframes ? dframes[i] : 0, &local, &rc_flag);
    if (rc != SWA_OK) return rc;
    ids[size_t(i)] = int32_t(local);
    minus[size_t(i)] = rc_flag ? 1 : 0;
  }
  HIP_TRY(hipSetDevice(db->device));
  hipStream_t st = db->stream;
  const swa_seqs sq = db->seqs();
  const size_t threads = size_t((n + 63) / 64) * 64;
  DevBuf<int32_t> d_ids;
  DevBuf<uint8_t> d_minus;
  DevBuf<long long> d_h, d_e, d_out;
  HIP_TRY(d_ids.reserve(size_t(n)));
  HIP_TRY(d_minus.reserve(size_t(n)));
  HIP_TRY(d_out.reserve(3 * size_t(n)));
  HIP_TRY(db->qseq.reserve(size_t(qlen > 0 ? qlen : 1)));
  if (qlen) HIP_TRY(hipMemcpyAsync(db->qseq.p, query, size_t(qlen), hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(d_ids.p, ids.data(), size_t(n) * sizeof(int32_t), hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(d_minus.p, minus.data(), size_t(n), hipMemcpyHostToDevice, st));
  // one wave per sequence in int32 whenever no score can leave 32 bits; the one-thread 64-bit form otherwise
  const int64_t span = std::max<int64_t>(qlen, 1) * std::max<int64_t>(db->hi, 1);
  if (!db->opt.endpoints_thread && span < (int64_t(1) << 30) && db->goe < (int64_t(1) << 30) &&
      db->ge < (int64_t(1) << 30)) {
    const int rows = swa_endpoints_rows_for(int(qlen));
    DevBuf<int> d_bh, d_bf;
    DevBuf<int64_t> d_boff;
    const bool passes = qlen > 64 * rows;
    if (passes) {                                 // bottom-row hand-over between passes: one int pair per column
      std::vector<int64_t> boff((size_t(n)));
      int64_t total = 0;
      for (int64_t i = 0; i < n; ++i) {
        boff[size_t(i)] = total;
        total += db->h_offsets[size_t(ids[size_t(i)]) + 1] - db->h_offsets[size_t(ids[size_t(i)])];
      }
      HIP_TRY(d_bh.reserve(size_t(total) + 1));
      HIP_TRY(d_bf.reserve(size_t(total) + 1));
      HIP_TRY(d_boff.reserve(size_t(n)));
      HIP_TRY(hipMemcpyAsync(d_boff.p, boff.data(), size_t(n) * sizeof(int64_t), hipMemcpyHostToDevice, st));
      HIP_TRY(hipStreamSynchronize(st));          // boff lives on this stack frame
    }
    HIP_TRY(swa_launch_endpoints_wave(&sq, d_ids.p, d_minus.p, int(n), db->qseq.p, int(qlen),
                                      db->matrix.p, int(db->goe), int(db->ge), passes ? d_bh.p : nullptr,
                                      passes ? d_bf.p : nullptr, passes ? d_boff.p : nullptr, d_out.p, nullptr, st));
    out.resize(3 * size_t(n));
    HIP_TRY(hipMemcpyAsync(out.data(), d_out.p, out.size() * sizeof(long long), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return SWA_OK;
  }
  HIP_TRY(d_h.reserve(threads * size_t(qlen > 0 ? qlen : 1)));
  HIP_TRY(d_e.reserve(threads * size_t(qlen > 0 ? qlen : 1)));
  HIP_TRY(swa_launch_endpoints(&sq, d_ids.p, d_minus.p, int(n), db->qseq.p, int(qlen),
                               db->matrix.p, db->goe, db->ge, d_h.p, d_e.p, d_out.p, st));
  out.resize(3 * size_t(n));
  HIP_TRY(hipMemcpyAsync(out.data(), d_out.p, out.size() * sizeof(long long), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return SWA_OK;
}

// db_getsequence (database.cc:1237-1401) out of the resident shard
int fetch_sequence(swa_db* db, int64_t seqno, int dstrand, int dframe, std::vector<uint8_t>& seq)
{
  int64_t local = 0;
  bool minus = false;
  const int rc = locate(db, seqno, dstrand, dframe, &local, &minus);
  if (rc != SWA_OK) return rc;
  const int64_t o = db->h_offsets[size_t(local)], len = db->h_offsets[size_t(local) + 1] - o;
  seq.resize(size_t(len));
  if (len) {
    HIP_TRY(hipSetDevice(db->device));
    if (db->packed) {                                    // two residues per byte: the covering bytes, unpacked here
      const int64_t b0 = o >> 1, b1 = (o + len + 1) >> 1;
      std::vector<uint8_t> raw(size_t(b1 - b0));
      HIP_TRY(hipMemcpy(raw.data(), db->residues.p + b0, raw.size(), hipMemcpyDeviceToHost));
      for (int64_t k = 0; k < len; ++k) seq[size_t(k)] = (raw[size_t(((o + k) >> 1) - b0)] >> (((o + k) & 1) * 4)) & 15;
    } else {
      HIP_TRY(hipMemcpy(seq.data(), db->residues.p + o, size_t(len), hipMemcpyDeviceToHost));
    }
  }
  if (minus) {
    static const uint8_t compl4[16] = {0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15};   // database.cc ntcompl
    std::reverse(seq.begin(), seq.end());
    for (uint8_t& c : seq) c = compl4[c & 15];
  }
  return SWA_OK;
}
// the same for a list: one gather kernel and one copy instead of a synchronous copy per sequence
int fetch_sequences(swa_db* db, const int64_t* seqnos, const int32_t* dstrands, const int32_t* dframes, int64_t n,
                    std::vector<std::vector<uint8_t>>& seqs)
{
  seqs.assign(size_t(n), {});
  if (n == 0) return SWA_OK;
  std::vector<int32_t> ids((size_t(n)));
  std::vector<uint8_t> minus((size_t(n)), 0);
  std::vector<int64_t> out_off(size_t(n) + 1, 0);
  for (int64_t i = 0; i < n; ++i) {
    int64_t local = 0;
    bool m = false;
    const int rc = locate(db, seqnos[i], dstrands ? (dstrands[i] ? 1 : 0) : 0, dframes ? dframes[i] : 0, &local, &m);
    if (rc != SWA_OK) return rc;
    ids[size_t(i)] = int32_t(local);
    minus[size_t(i)] = m;
    out_off[size_t(i) + 1] = out_off[size_t(i)] + db->h_offsets[size_t(local) + 1] - db->h_offsets[size_t(local)];
  }
  const int64_t total = out_off[size_t(n)];
  HIP_TRY(hipSetDevice(db->device));
  hipStream_t st = db->stream;
  const swa_seqs sq = db->seqs();
  DevBuf<int32_t> d_ids;
  DevBuf<int64_t> d_off;
  DevBuf<uint8_t> d_out;
  HIP_TRY(d_ids.reserve(size_t(n)));
  HIP_TRY(d_off.reserve(size_t(n)));
  HIP_TRY(d_out.reserve(size_t(total) + 1));
  HIP_TRY(hipMemcpyAsync(d_ids.p, ids.data(), size_t(n) * sizeof(int32_t), hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(d_off.p, out_off.data(), size_t(n) * sizeof(int64_t), hipMemcpyHostToDevice, st));
  HIP_TRY(swa_launch_gather(&sq, d_ids.p, d_off.p, int(n), d_out.p, st));
  std::vector<uint8_t> all(size_t(total) + 1);
  HIP_TRY(hipMemcpyAsync(all.data(), d_out.p, size_t(total), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  static const uint8_t compl4[16] = {0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15};     // database.cc ntcompl
  for (int64_t i = 0; i < n; ++i) {
    std::vector<uint8_t>& seq = seqs[size_t(i)];
    seq.assign(all.begin() + out_off[size_t(i)], all.begin() + out_off[size_t(i) + 1]);
    if (minus[size_t(i)]) {
      std::reverse(seq.begin(), seq.end());
      for (uint8_t& c : seq) c = compl4[c & 15];
    }
  }
  return SWA_OK;
}
}  // namespace

extern "C" int swa_search_endpoints_strand(swa_db* db, const uint8_t* query, int64_t qlen, const int64_t* seqnos,
                                           const int32_t* dstrands, const int32_t* dframes, int64_t n, int64_t* scores,
                                           int64_t* bestpos, int64_t* bestq)
try {
  { const int src_ = loaded(db); if (src_ != SWA_OK) return src_; }
  int rc = check_query(db, query, qlen);
  if (rc != SWA_OK) return rc;
  if (n < 0 || (n > 0 && (!seqnos || !scores || !bestpos || !bestq))) return fail(SWA_EINVAL, "bad argument");
  if (n == 0) return SWA_OK;
  if (db->streamed)                                       // bind the owning part, run, release
    return streamed_by_owner(db, seqnos, n, [&](swa_db* slot, const std::vector<int64_t>& idx) -> int {
      const size_t m = idx.size();
      std::vector<int64_t> sq(m), sc(m), bp(m), bq(m);
      std::vector<int32_t> ds(m, 0), df(m, 0);
      for (size_t k = 0; k < m; ++k) { sq[k] = seqnos[idx[k]]; if (dstrands) ds[k] = dstrands[idx[k]]; if (dframes) df[k] = dframes[idx[k]]; }
      const int r = swa_search_endpoints_strand(slot, query, qlen, sq.data(), ds.data(), df.data(), int64_t(m), sc.data(), bp.data(), bq.data());
      for (size_t k = 0; k < m && r == SWA_OK; ++k) { scores[idx[k]] = sc[k]; bestpos[idx[k]] = bp[k]; bestq[idx[k]] = bq[k]; }
      return r;
    });
  std::vector<long long> out;
  rc = endpoints_on_device(db, query, qlen, seqnos, dstrands, dframes, n, out);
  if (rc != SWA_OK) return rc;
  for (int64_t i = 0; i < n; ++i) {
    scores[i] = out[size_t(i)];
    bestpos[i] = out[size_t(n + i)];
    bestq[i] = out[size_t(2 * n + i)];
  }
  return SWA_OK;
} SWA_CATCH

extern "C" int swa_search_endpoints(swa_db* db, const uint8_t* query, int64_t qlen, const int64_t* seqnos, int64_t n,
                                    int64_t* scores, int64_t* bestpos, int64_t* bestq)
try {
  return swa_search_endpoints_strand(db, query, qlen, seqnos, nullptr, nullptr, n, scores, bestpos, bestq);
} SWA_CATCH

extern "C" int swa_db_sequence(swa_db* db, int64_t seqno, int dstrand, int dframe, uint8_t* buf, int64_t cap,
                               int64_t* len, int64_t* ntlen)
try {
  { const int src_ = loaded(db); if (src_ != SWA_OK) return src_; }
  if (!db || !len || cap < 0 || (cap > 0 && !buf)) return fail(SWA_EINVAL, "bad argument");
  if (db->streamed)
    return streamed_by_owner(db, &seqno, 1, [&](swa_db* slot, const std::vector<int64_t>&) -> int {
      return swa_db_sequence(slot, seqno, dstrand, dframe, buf, cap, len, ntlen);
    });
  std::vector<uint8_t> seq;
  const int rc = fetch_sequence(db, seqno, dstrand, dframe, seq);
  if (rc != SWA_OK) return rc;
  if (ntlen) *ntlen = db->frames == 6 ? db->h_ntlen[size_t(seqno - db->first_seqno)] : 0;
  *len = int64_t(seq.size());
  if (int64_t(seq.size()) > cap) return fail(SWA_ERANGE, "sequence buffer too small");
  if (!seq.empty()) std::memcpy(buf, seq.data(), seq.size());
  return SWA_OK;
} SWA_CATCH

namespace {
// hits_align's call of align() (hits.cc:585-616) for one sequence already on the host.  hint_score != 0:
// trust (q_end, d_end); otherwise find score and end by the forward sweep.
int align_on_host(const uint8_t* query, int64_t qlen, const uint8_t* dseq, int64_t dlen, const int32_t* matrix,
                  int64_t gapopen, int64_t gapextend, int64_t hint_score, int64_t q_end, int64_t d_end,
                  swa_alignment_t& a, std::vector<swa::EditOp>& ops)
{
  int64_t score = hint_score;
  a.hinted = hint_score ? 1 : 0;
  if (!hint_score) {
    q_end = d_end = 0;
    score = swa::forward_end(query, qlen, dseq, dlen, matrix, gapopen, gapextend, &q_end, &d_end);
  }
  int64_t q_start = 0, d_start = 0;
  if (qlen <= 0 || dlen <= 0 || q_end >= qlen || d_end >= dlen || q_end < 0 || d_end < 0 ||
      !swa::backward_start(query, dseq, matrix, gapopen, gapextend, score, q_end, d_end, &q_start, &d_start))
    return fail(SWA_EINVAL, "Internal error in align function.");     // align.cc:156 (e.g. a hit of score 0)
  ops.clear();
  swa::edit_script(query, dseq, matrix, gapopen, gapextend, q_start, d_start, q_end, d_end, ops);
  swa::count_columns(query, dseq, matrix, q_start, d_start, ops, &a.identities, &a.positives, &a.indels, &a.aligned,
                     &a.gaps);
  a.score = score;
  a.dlen = dlen;
  a.q_start = q_start; a.q_end = q_end; a.d_start = d_start; a.d_end = d_end;
  return SWA_OK;
}
}  // namespace

extern "C" int swa_traceback(const uint8_t* query, int64_t qlen, const uint8_t* dseq, int64_t dlen, const int64_t* M,
                             int64_t gapopen, int64_t gapextend, int64_t hint_score, int64_t hint_q_end,
                             int64_t hint_d_end, swa_alignment_t* out, char* text, int64_t text_cap, int64_t* text_used)
try {
  if (!query || !dseq || !M || !out || !text_used || text_cap < 0 || (text_cap > 0 && !text) || qlen < 0 || dlen < 0)
    return fail(SWA_EINVAL, "bad argument");
  for (int64_t i = 0; i < qlen; ++i) if (query[i] >= 32) return fail(SWA_EINVAL, "query symbol code >= 32");
  for (int64_t j = 0; j < dlen; ++j) if (dseq[j] >= 32) return fail(SWA_EINVAL, "sequence symbol code >= 32");
  std::vector<int32_t> m32(1024);
  for (int i = 0; i < 1024; ++i) m32[size_t(i)] = int32_t(M[i]);
  std::memset(out, 0, sizeof *out);
  std::vector<swa::EditOp> ops;
  const int rc = align_on_host(query, qlen, dseq, dlen, m32.data(), gapopen, gapextend, hint_score, hint_q_end,
                               hint_d_end, *out, ops);
  if (rc != SWA_OK) return rc;
  std::string s;
  for (const swa::EditOp& op : ops) { s += op.kind; s += std::to_string(op.count); }
  out->cigar_offset = 0;
  out->cigar_len = int64_t(s.size());
  *text_used = int64_t(s.size()) + 1;
  if (*text_used > text_cap) return fail(SWA_ERANGE, "text buffer too small for the edit script");
  std::memcpy(text, s.c_str(), s.size() + 1);
  return SWA_OK;
} SWA_CATCH

// align_chunk + hits_align (swipe.cc:339-414, hits.cc:546-618): end points on the GPU, start point and edit
// script on the host
extern "C" int swa_align_hits(swa_db* db, const uint8_t* query, int64_t qlen, const int64_t* seqnos,
                              const int32_t* dstrands, const int32_t* dframes, int64_t n, swa_alignment_t* out,
                              char* text, int64_t text_cap, int64_t* text_used)
try {
  { const int src_ = loaded(db); if (src_ != SWA_OK) return src_; }
  int rc = check_query(db, query, qlen);
  if (rc != SWA_OK) return rc;
  if (n < 0 || text_cap < 0 || !text_used || (n > 0 && (!seqnos || !out)) || (text_cap > 0 && !text))
    return fail(SWA_EINVAL, "bad argument");
  *text_used = 0;
  if (n == 0) return SWA_OK;
  if (db->streamed) {                                     // every hit aligned by the part that owns it; scripts in list order
    std::vector<std::string> scripts{size_t(n)};
    rc = streamed_by_owner(db, seqnos, n, [&](swa_db* slot, const std::vector<int64_t>& idx) -> int {
      const size_t m = idx.size();
      std::vector<int64_t> sq(m);
      std::vector<int32_t> ds(m, 0), df(m, 0);
      for (size_t k = 0; k < m; ++k) { sq[k] = seqnos[idx[k]]; if (dstrands) ds[k] = dstrands[idx[k]]; if (dframes) df[k] = dframes[idx[k]]; }
      std::vector<swa_alignment_t> al(m);
      std::vector<char> buf(1 << 16);
      int64_t used = 0;
      int r = swa_align_hits(slot, query, qlen, sq.data(), ds.data(), df.data(), int64_t(m), al.data(), buf.data(), int64_t(buf.size()), &used);
      if (r == SWA_ERANGE) {
        buf.resize(size_t(used));
        r = swa_align_hits(slot, query, qlen, sq.data(), ds.data(), df.data(), int64_t(m), al.data(), buf.data(), int64_t(buf.size()), &used);
      }
      if (r != SWA_OK) return r;
      for (size_t k = 0; k < m; ++k) {
        out[idx[k]] = al[k];
        scripts[size_t(idx[k])].assign(buf.data() + al[k].cigar_offset, size_t(al[k].cigar_len));
      }
      return SWA_OK;
    });
    if (rc != SWA_OK) return rc;
    std::string all;
    for (int64_t i = 0; i < n; ++i) {
      out[i].cigar_offset = int64_t(all.size());
      out[i].cigar_len = int64_t(scripts[size_t(i)].size());
      all += scripts[size_t(i)];
      all += '\0';
    }
    *text_used = int64_t(all.size());
    if (int64_t(all.size()) > text_cap) return fail(SWA_ERANGE, "text buffer too small for the edit scripts");
    std::memcpy(text, all.data(), all.size());
    return SWA_OK;
  }
  std::vector<long long> ends;
  rc = endpoints_on_device(db, query, qlen, seqnos, dstrands, dframes, n, ends);
  if (rc != SWA_OK) return rc;
  const int64_t gapopen = db->goe - db->ge, gapextend = db->ge;
  const int64_t limit16 = 65536 - db->hi;                           // SCORELIMIT_16, matrices.cc:578
  // sequences come back from the device on this thread; the tracebacks are independent and run on a few host
  // threads (the reference spreads align_chunk over its worker threads the same way, swipe.cc:615-647)
  std::vector<std::vector<uint8_t>> dseqs;
  rc = fetch_sequences(db, seqnos, dstrands, dframes, n, dseqs);
  if (rc != SWA_OK) return rc;
  std::vector<std::string> scripts{size_t(n)};
  std::vector<int> status(size_t(n), SWA_OK);
  auto work = [&](int64_t first, int64_t step) {
    std::vector<swa::EditOp> ops;
    for (int64_t i = first; i < n; i += step) {
      const std::vector<uint8_t>& dseq = dseqs[size_t(i)];
      swa_alignment_t& a = out[i];
      std::memset(&a, 0, sizeof a);
      const int64_t score = ends[size_t(i)], d_end = ends[size_t(n + i)], q_end = ends[size_t(2 * n + i)];
      // the hint is honoured only if the 16-bit lanes did not saturate and neither coordinate is 0
      // (swipe.cc:404, hits.cc:587); otherwise align() starts from scratch
      const bool hinted = score < limit16 && q_end > 0 && d_end != 0;
      status[size_t(i)] = align_on_host(query, qlen, dseq.data(), int64_t(dseq.size()), db->h_matrix, gapopen, gapextend,
                                        hinted ? score : 0, q_end, d_end, a, ops);
      if (status[size_t(i)] != SWA_OK) continue;
      a.seqno = seqnos[i];
      a.dstrand = dstrands ? (dstrands[i] ? 1 : 0) : 0;
      a.dframe = dframes ? dframes[i] : 0;
      a.dlennt = db->frames == 6 ? db->h_ntlen[size_t(seqnos[i] - db->first_seqno)] : 0;
      std::string& sc = scripts[size_t(i)];
      for (const swa::EditOp& op : ops) { sc += op.kind; sc += std::to_string(op.count); }
    }
  };
  // two alignments per thread at least; the hits are ordered by score, so a thread takes every nthreads-th one
  const int64_t nthreads = std::max<int64_t>(1, std::min<int64_t>({n / 2, int64_t(std::thread::hardware_concurrency()), 64}));
  if (nthreads == 1) {
    work(0, 1);
  } else {
    std::vector<std::thread> pool;
    for (int64_t t = 0; t < nthreads; ++t) pool.emplace_back(work, t, nthreads);
    for (std::thread& t : pool) t.join();
  }
  std::string all;
  for (int64_t i = 0; i < n; ++i) {
    if (status[size_t(i)] != SWA_OK) return fail(status[size_t(i)], "Internal error in align function.");   // align.cc:156
    out[i].cigar_offset = int64_t(all.size());
    out[i].cigar_len = int64_t(scripts[size_t(i)].size());
    all += scripts[size_t(i)];
    all += '\0';
  }
  *text_used = int64_t(all.size());
  if (int64_t(all.size()) > text_cap) return fail(SWA_ERANGE, "text buffer too small for the edit scripts");
  std::memcpy(text, all.data(), all.size());
  return SWA_OK;
} SWA_CATCH
