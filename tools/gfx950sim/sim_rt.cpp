// gfx950sim: a stand-in for the 34 entry points of libamdhip64 that libswipe_amd.so uses, backed by the interpreter of
// sim_isa.cpp. LD_PRELOAD it (tools/gfx950sim/run.sh) and the product library runs unchanged, kernels included, on a machine
// without a GPU: "device" memory is host memory, every stream is synchronous, every kernel launch is interpreted at once,
// every device access is checked against the table of live allocations. TEST INFRASTRUCTURE ONLY - see sim_core.h.
//
// Environment: HIPSIM_CACHE (default /tmp/gfx950sim-cache) disassembly cache; HIPSIM_CUS (default 8) multiProcessorCount
// reported (persistent grids scale with it); HIPSIM_MEM_GB (default 16); HIPSIM_STATS=<file> per-kernel wave-instruction
// counts appended at exit and at every hipDeviceSynchronize; HIPSIM_ABORT=1 abort() on the first device fault;
// HIPSIM_POISON=<u32> initial register / LDS / allocation pattern; HIPSIM_SWITCH=<n> waves of a workgroup take turns every
// <= n instructions in a seeded random order (race hunting); HIPSIM_TRACE=1 one line per launch; HIPSIM_TRACE_INSN=<n> the first
// n instructions of wave 0 with EXEC / VCC / operands; HIPSIM_LDS_UNDEF=1 a read of an LDS byte no lane of the workgroup has
// written is a fault; HIPSIM_BACKTRACE=1 host stack on SIGSEGV.  A launch that asks for more LDS per workgroup or VGPRs per
// SIMD than gfx950 has is refused, as the hardware would.
// HIPSIM_ASYNC=<seed>: streams become queues. Launches, memsets and async copies are deferred until the host synchronises with
// something, and are then executed in a seeded order among the streams that only the edges HIP promises constrain: program order
// inside a stream, hipEventRecord -> hipStreamWaitEvent, the legacy null stream against blocking streams, hipFree as a device
// synchronisation. hipMemcpyAsync reads its host source (and writes its host destination) when it EXECUTES. seed % 3 picks the
// policy: 0 uniformly random stream, 1 lazy (only what the wait needs - every other stream is as late as it may be),
// 2 everything else first. A missing event edge, a host buffer reused under an in-flight copy or a result read before its
// synchronisation shows as wrong data under at least one policy. Still invisible: the memory model, co-residency, timing.
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <dlfcn.h>
#include <execinfo.h>
#include <signal.h>
#include <sys/stat.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fstream>
#include <map>
#include <algorithm>
#include <atomic>
#include <mutex>
#include <random>
#include <set>
#include <sstream>
#include "sim_core.h"

namespace {

using Lock = std::lock_guard<std::recursive_mutex>;

struct Alloc { uint64_t size; void* raw; bool host; uint8_t* shadow; };
struct FatBin;
struct Registered { FatBin* fb; std::string name; };
// Everything with a constructor lives in ONE object made on first use and never destroyed: a program that links
// libswipe_amd.so directly (swipe_amd_cli) runs the library's fat-binary registration before this preloaded library's own
// static initialisers, and may call in from atexit handlers after its destructors.
struct State {
    std::recursive_mutex mu;
    std::map<uint64_t, Alloc> allocs;        // start -> allocation; the checked range is [start, start + size)
    std::map<const void*, Registered> funcs;
    std::string fault;                       // sticky
};
State& S() { static State* s = new State; return *s; }
#define g_mu (S().mu)
#define g_allocs (S().allocs)
#define g_funcs (S().funcs)
#define g_fault (S().fault)
uint64_t g_allocated = 0;
std::atomic<uint64_t> g_gen{1};                // bumped by every allocation / free: invalidates the per-thread range caches
thread_local uint64_t t_lo = 1, t_hi = 0, t_gen = 0;    // last range that answered mem_ok
thread_local uint8_t* t_shadow = nullptr;               // ... and its shadow bytes (HIPSIM_MEM_UNDEF)

struct FatBin {
    const uint8_t* bundle = nullptr;
    size_t size = 0;
    std::string dir;
    bool loaded = false;
    std::map<std::string, sim::Kernel> kernels;
};
hipError_t g_last = hipSuccess;

struct CallCfg { dim3 grid, block; size_t shmem; hipStream_t stream; };

// streams and events; with HIPSIM_ASYNC a stream is a queue of operations (see the head of this file)
struct SimEvent { std::chrono::steady_clock::time_point t; bool recorded; uint64_t ticket; };
struct Op {
    enum Kind : uint8_t { LAUNCH, COPY, SET, REC, WAIT } kind = WAIT;
    sim::Kernel* k = nullptr; uint8_t* ka = nullptr; dim3 grid, block; size_t shmem = 0;                      // LAUNCH
    void* dst = nullptr; const void* src = nullptr; size_t n = 0; hipMemcpyKind ck = hipMemcpyDefault; int val = 0;   // COPY, SET
    SimEvent* ev = nullptr; uint64_t ticket = 0;                                                              // REC, WAIT
};
struct SimStream { bool blocking; std::deque<Op> q; };
struct Async {
    bool on = getenv("HIPSIM_ASYNC") != nullptr;
    uint64_t seed = on ? strtoull(getenv("HIPSIM_ASYNC"), nullptr, 0) : 0;
    int policy = (int)(seed % 3);
    std::mt19937_64 rng{seed * 0x9E3779B97F4A7C15ull + 1};
    std::set<SimStream*> streams;
    std::map<uint64_t, SimStream*> pending;   // ticket of a record that has not executed -> the stream it is queued on
    uint64_t next_ticket = 0;
};
Async& A() { static Async* a = new Async; return *a; }
thread_local std::vector<CallCfg> t_cfg;

uint32_t poison() { static uint32_t p = getenv("HIPSIM_POISON") ? (uint32_t)strtoul(getenv("HIPSIM_POISON"), nullptr, 0) : 0xBAD0BAD1u; return p; }
int env_int(const char* n, int d) { const char* e = getenv(n); return e ? atoi(e) : d; }

std::string self_dir() {
    Dl_info di;
    if (dladdr((void*)&self_dir, &di) && di.dli_fname) {
        std::string p = di.dli_fname;
        size_t s = p.rfind('/');
        return s == std::string::npos ? "." : p.substr(0, s);
    }
    return ".";
}

bool file_exists(const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0; }

size_t bundle_size(const uint8_t* b) {
    uint64_t n; memcpy(&n, b + 24, 8);
    size_t p = 32, end = 0;
    for (uint64_t i = 0; i < n; i++) {
        uint64_t off, size, tl;
        memcpy(&off, b + p, 8); memcpy(&size, b + p + 8, 8); memcpy(&tl, b + p + 16, 8);
        p += 24 + tl;
        if (off + size > end) end = off + size;
    }
    return end > p ? end : p;
}

bool load_fatbin(FatBin& fb, std::string& err) {
    if (fb.loaded) return true;
    if (memcmp(fb.bundle, "__CLANG_OFFLOAD_BUNDLE__", 24) != 0) { err = "fat binary is not an uncompressed clang offload bundle"; return false; }
    fb.size = bundle_size(fb.bundle);
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < fb.size; i++) { h ^= fb.bundle[i]; h *= 1099511628211ull; }
    const char* cache = getenv("HIPSIM_CACHE");
    std::string root = cache ? cache : "/tmp/gfx950sim-cache";
    mkdir(root.c_str(), 0777);
    char name[64]; snprintf(name, sizeof name, "/%016llx", (unsigned long long)h);
    fb.dir = root + name;
    if (!file_exists(fb.dir + "/ok")) {
        std::string tmp = fb.dir + ".tmp." + std::to_string(getpid());
        mkdir(tmp.c_str(), 0777);
        std::string bf = tmp + "/bundle.bin";
        { std::ofstream o(bf, std::ios::binary); o.write((const char*)fb.bundle, (std::streamsize)fb.size); }
        const char* prep = getenv("HIPSIM_PREP");
        // LD_PRELOAD must not follow into the tools
        std::string cmd = "env -u LD_PRELOAD python3 " + (prep ? std::string(prep) : self_dir() + "/prep.py") + " " + bf + " " + tmp + " 1>&2";
        if (system(cmd.c_str()) != 0) { err = "gfx950sim: '" + cmd + "' failed"; return false; }
        unlink(bf.c_str());
        if (rename(tmp.c_str(), fb.dir.c_str()) != 0 && !file_exists(fb.dir + "/ok")) { err = "gfx950sim: cannot move " + tmp; return false; }
        if (file_exists(tmp)) { std::string rm = "rm -rf " + tmp; if (system(rm.c_str())) {} }
    }
    std::ifstream m(fb.dir + "/co.meta");
    if (!m) { err = "gfx950sim: no co.meta in " + fb.dir; return false; }
    std::string line;
    sim::Kernel* cur = nullptr;
    while (std::getline(m, line)) {
        std::istringstream is(line);
        std::string tag; is >> tag;
        if (tag == "K") {
            sim::Kernel k;
            is >> k.name >> k.code_addr >> k.lds >> k.scratch >> k.kernarg >> k.rsrc1 >> k.rsrc2 >> k.rsrc3 >> k.props >> k.preload;
            k.sfile = fb.dir + "/co.s";
            cur = &(fb.kernels[k.name] = k);
        } else if (tag == "A" && cur) {
            sim::KArg a; is >> a.offset >> a.size >> a.kind;
            cur->args.push_back(a);
        }
    }
    fb.loaded = true;
    return true;
}

void dump_stats() {
    const char* path = getenv("HIPSIM_STATS");
    if (!path) return;
    Lock lk(g_mu);
    FILE* f = fopen(path, "a");
    if (!f) return;
    static const char* cls[] = {"salu", "valu", "vop3p", "lds", "vmem", "smem", "branch", "other"};
    for (auto& fr : g_funcs) {
        auto it = fr.second.fb->kernels.find(fr.second.name);
        if (it == fr.second.fb->kernels.end() || !it->second.launches) continue;
        sim::Kernel& k = it->second;
        fprintf(f, "{\"pid\": %d, \"kernel\": \"%s\", \"launches\": %llu, \"workgroups\": %llu", (int)getpid(), k.name.c_str(), (unsigned long long)k.launches, (unsigned long long)k.wgs);
        for (int i = 0; i < sim::C_N; i++) fprintf(f, ", \"%s\": %llu", cls[i], (unsigned long long)k.count[i]);
        fprintf(f, "}\n");
        k.launches = k.wgs = 0;
        for (int i = 0; i < sim::C_N; i++) k.count[i] = 0;
    }
    fclose(f);
}
struct AtExit { ~AtExit() { dump_stats(); } } g_atexit;

// HIPSIM_BACKTRACE=1: a SIGSEGV / SIGABRT of the host program prints its stack (the image has no debugger)
void on_fatal(int sig) {
    void* frames[64];
    int n = backtrace(frames, 64);
    fprintf(stderr, "gfx950sim: signal %d, backtrace:\n", sig);
    backtrace_symbols_fd(frames, n, 2);
    _exit(128 + sig);
}
struct Install { Install() { if (env_int("HIPSIM_BACKTRACE", 0)) { signal(SIGSEGV, on_fatal); signal(SIGABRT, on_fatal); signal(SIGBUS, on_fatal); } } } g_install;

hipError_t fail(hipError_t e) { g_last = e; return e; }
hipError_t sync_status() { return g_fault.empty() ? hipSuccess : fail(hipErrorLaunchFailure); }

bool check_dev(const void* p, size_t n, const char* what) {
    if (n == 0 || sim::mem_ok((uint64_t)(uintptr_t)p, n)) return true;
    fprintf(stderr, "gfx950sim: %s: %zu bytes at %p are not inside a live allocation: %s\n", what, n, p, sim::mem_describe((uint64_t)(uintptr_t)p).c_str());
    if (env_int("HIPSIM_ABORT", 0)) abort();
    return false;
}

}  // namespace

namespace sim {
// No lock: the table only changes under g_mu, and the thread that launched a kernel holds g_mu for as long as its workers run.
bool mem_ok(uint64_t a, uint64_t n) {
    const uint64_t gen = g_gen.load(std::memory_order_relaxed);
    if (t_gen == gen && a >= t_lo && a + n <= t_hi) return true;
    auto it = g_allocs.upper_bound(a);
    if (it == g_allocs.begin()) return false;
    --it;
    if (a + n > it->first + it->second.size) return false;
    t_lo = it->first; t_hi = it->first + it->second.size; t_gen = gen; t_shadow = it->second.shadow;
    return true;
}
uint8_t* mem_shadow(uint64_t a) { return t_shadow ? t_shadow + (a - t_lo) : nullptr; }
std::string mem_describe(uint64_t a) {
    char buf[256];
    auto it = g_allocs.upper_bound(a);
    if (it != g_allocs.begin()) {
        auto p = std::prev(it);
        snprintf(buf, sizeof buf, "%lld bytes past the end of the %llu-byte allocation at 0x%llx", (long long)(a - (p->first + p->second.size)), (unsigned long long)p->second.size,
                 (unsigned long long)p->first);
        std::string s = buf;
        if (it != g_allocs.end()) { snprintf(buf, sizeof buf, "; %llu bytes before the allocation at 0x%llx", (unsigned long long)(it->first - a), (unsigned long long)it->first); s += buf; }
        return s;
    }
    if (it != g_allocs.end()) { snprintf(buf, sizeof buf, "%llu bytes before the first allocation", (unsigned long long)(it->first - a)); return buf; }
    return "no allocation is live";
}
}  // namespace sim

static void* dev_alloc(size_t size, bool host) {
    const size_t guard = 256;
    size_t n = size ? size : 1;
    uint8_t* raw = (uint8_t*)aligned_alloc(256, (n + 2 * guard + 255) & ~(size_t)255);
    if (!raw) return nullptr;
    uint32_t p = poison();
    if (n <= (64u << 20)) for (size_t i = 0; i + 4 <= n + 2 * guard; i += 4) memcpy(raw + i, &p, 4);     // big buffers stay untouched (lazy pages)
    void* user = raw + guard;
    static const bool undef = env_int("HIPSIM_MEM_UNDEF", 0) != 0;
    g_allocs[(uint64_t)(uintptr_t)user] = Alloc{n, raw, host, undef && !host ? (uint8_t*)calloc(n, 1) : nullptr};
    g_allocated += n;
    g_gen++;
    return user;
}
static hipError_t dev_free(void* p) {
    if (!p) return hipSuccess;
    auto it = g_allocs.find((uint64_t)(uintptr_t)p);
    if (it == g_allocs.end()) { fprintf(stderr, "gfx950sim: free of %p which is not a live allocation\n", p); return fail(hipErrorInvalidValue); }
    g_allocated -= it->second.size;
    free(it->second.raw);
    free(it->second.shadow);
    g_allocs.erase(it);
    g_gen++;
    return hipSuccess;
}

extern "C" {

void** __hipRegisterFatBinary(const void* data) {
    struct Wrapper { uint32_t magic, version; const void* binary; const void* unused; };
    const Wrapper* w = (const Wrapper*)data;
    Lock lk(g_mu);
    FatBin* fb = new FatBin();
    fb->bundle = (const uint8_t*)w->binary;
    return (void**)fb;
}
void __hipUnregisterFatBinary(void** /*modules*/) { dump_stats(); }
void __hipRegisterFunction(void** modules, const void* hostFunction, char* /*deviceFunction*/, const char* deviceName, unsigned, void*, void*, void*, void*, int*) {
    Lock lk(g_mu);
    g_funcs[hostFunction] = Registered{(FatBin*)modules, deviceName};
}
void __hipRegisterVar(void**, void*, char*, const char* name, int, size_t, int, int) { fprintf(stderr, "gfx950sim: device variable %s is not modelled\n", name); }
void __hipRegisterManagedVar(void*, void**, void*, const char* name, size_t, unsigned) { fprintf(stderr, "gfx950sim: managed variable %s is not modelled\n", name); }

hipError_t __hipPushCallConfiguration(dim3 grid, dim3 block, size_t shmem, hipStream_t stream) { t_cfg.push_back(CallCfg{grid, block, shmem, stream}); return hipSuccess; }
hipError_t __hipPopCallConfiguration(dim3* grid, dim3* block, size_t* shmem, hipStream_t* stream) {
    if (t_cfg.empty()) return fail(hipErrorInvalidValue);
    CallCfg c = t_cfg.back(); t_cfg.pop_back();
    *grid = c.grid; *block = c.block; *shmem = c.shmem; *stream = c.stream;
    return hipSuccess;
}

}  // extern "C"

// Resolve the kernel and snapshot its arguments (HIP copies them at the launch call, whenever the kernel runs).
static hipError_t prepare_launch(const void* func, dim3 grid, dim3 block, void** args, size_t shmem, Op& op) {
    if (!g_fault.empty()) return fail(hipErrorLaunchFailure);
    auto it = g_funcs.find(func);
    if (it == g_funcs.end()) { fprintf(stderr, "gfx950sim: launch of an unregistered function %p\n", func); return fail(hipErrorInvalidDeviceFunction); }
    std::string err;
    FatBin& fb = *it->second.fb;
    if (!load_fatbin(fb, err)) { fprintf(stderr, "%s\n", err.c_str()); g_fault = err; return fail(hipErrorInvalidDeviceFunction); }
    auto kit = fb.kernels.find(it->second.name);
    if (kit == fb.kernels.end()) { fprintf(stderr, "gfx950sim: kernel %s not in its code object\n", it->second.name.c_str()); return fail(hipErrorInvalidDeviceFunction); }
    sim::Kernel& k = kit->second;
    size_t kbytes = ((size_t)k.kernarg + 255) & ~(size_t)255;
    uint8_t* ka = (uint8_t*)dev_alloc(kbytes, true);
    memset(ka, 0, kbytes);
    int ai = 0;
    uint32_t dims = grid.z > 1 ? 3 : grid.y > 1 ? 2 : 1;
    for (auto& a : k.args) {
        uint8_t* p = ka + a.offset;
        auto put = [&](uint64_t v) { memcpy(p, &v, a.size); };
        if (a.kind.rfind("hidden_", 0) != 0) { memcpy(p, args[ai++], a.size); continue; }
        if (a.kind == "hidden_block_count_x") put(grid.x);
        else if (a.kind == "hidden_block_count_y") put(grid.y);
        else if (a.kind == "hidden_block_count_z") put(grid.z);
        else if (a.kind == "hidden_group_size_x") put(block.x);
        else if (a.kind == "hidden_group_size_y") put(block.y);
        else if (a.kind == "hidden_group_size_z") put(block.z);
        else if (a.kind == "hidden_grid_dims") put(dims);
        else if (a.kind == "hidden_dynamic_lds_size") put(shmem);
        else if (a.kind == "hidden_private_base") put(0x00020000u);
        else if (a.kind == "hidden_shared_base") put(0x00010000u);
        // remainders, global offsets, printf / hostcall / heap / queue pointers: zero
    }
    op.kind = Op::LAUNCH; op.k = &k; op.ka = ka; op.grid = grid; op.block = block; op.shmem = shmem;
    return hipSuccess;
}

static void copy_now(void* dst, const void* src, size_t n) {
    if (!n) return;
    memmove(dst, src, n);
    // defined-ness travels with the bytes
    uint8_t* ds = sim::mem_ok((uint64_t)(uintptr_t)dst, n) ? sim::mem_shadow((uint64_t)(uintptr_t)dst) : nullptr;
    if (ds) {
        uint8_t* ss = sim::mem_ok((uint64_t)(uintptr_t)src, n) ? sim::mem_shadow((uint64_t)(uintptr_t)src) : nullptr;
        if (ss) memmove(ds, ss, n); else memset(ds, 1, n);
    }
}
static bool copy_ranges_ok(void* dst, const void* src, size_t n, hipMemcpyKind kind) {
    bool ok = true;
    if (kind == hipMemcpyHostToDevice || kind == hipMemcpyDeviceToDevice) ok &= check_dev(dst, n, "copy destination");
    if (kind == hipMemcpyDeviceToHost || kind == hipMemcpyDeviceToDevice) ok &= check_dev(src, n, "copy source");
    return ok;
}

static void exec_op(Op& op) {
    switch (op.kind) {
    case Op::LAUNCH: {
        sim::Kernel& k = *op.k;
        if (g_fault.empty()) {
            if (env_int("HIPSIM_TRACE", 0))
                fprintf(stderr, "gfx950sim: launch %s grid (%u,%u,%u) block (%u,%u,%u) lds %u+%zu\n", k.name.c_str(), op.grid.x, op.grid.y, op.grid.z, op.block.x, op.block.y,
                        op.block.z, k.lds, op.shmem);
            std::string f = sim::run_kernel(k, sim::Dim3{op.grid.x, op.grid.y, op.grid.z}, sim::Dim3{op.block.x, op.block.y, op.block.z}, op.ka, (uint32_t)op.shmem);
            if (!f.empty()) {
                g_fault = f;
                fprintf(stderr, "gfx950sim: DEVICE FAULT: %s\n", f.c_str());
                if (env_int("HIPSIM_ABORT", 0)) abort();
            }
        }
        (void)dev_free(op.ka);
        break;
    }
    case Op::COPY:      // ranges were live at the call; a buffer freed under a queued copy is a fault of its own
        if (!copy_ranges_ok(op.dst, op.src, op.n, op.ck)) { if (g_fault.empty()) g_fault = "a queued copy ran after its device buffer was freed"; break; }
        copy_now(op.dst, op.src, op.n);
        break;
    case Op::SET:
        if (!check_dev(op.dst, op.n, "memset")) { if (g_fault.empty()) g_fault = "a queued memset ran after its buffer was freed"; break; }
        if (op.n) { memset(op.dst, op.val, op.n); uint8_t* ds = sim::mem_shadow((uint64_t)(uintptr_t)op.dst); if (ds) memset(ds, 1, op.n); }
        break;
    case Op::REC:
        if (op.ev) { op.ev->t = std::chrono::steady_clock::now(); op.ev->recorded = true; }
        A().pending.erase(op.ticket);
        break;
    case Op::WAIT: break;
    }
}

// ---- HIPSIM_ASYNC: the scheduler ----
static bool runnable(SimStream* s) { return !s->q.empty() && !(s->q.front().kind == Op::WAIT && A().pending.count(s->q.front().ticket)); }
static void step(SimStream* s) { Op op = s->q.front(); s->q.pop_front(); exec_op(op); }
// One scheduling decision. `goal`: the stream whose progress the caller waits for (nullptr: any). false = nothing can run.
static bool advance(SimStream* goal, int policy) {
    std::vector<SimStream*> run;
    for (SimStream* s : A().streams) if (runnable(s)) run.push_back(s);
    if (run.empty()) return false;
    SimStream* need = goal;                       // follow the event edges to the stream that can make the goal move
    for (int hop = 0; need && !runnable(need) && hop < 64; hop++) {
        if (need->q.empty()) { need = nullptr; break; }
        auto it = A().pending.find(need->q.front().ticket);
        need = it == A().pending.end() ? nullptr : it->second;
    }
    if (need && !runnable(need)) need = nullptr;
    auto any = [&](std::vector<SimStream*>& v) { return v[A().rng() % v.size()]; };
    SimStream* pick;
    if (policy == 1 && need) pick = need;
    else if (policy == 2 && need) {
        std::vector<SimStream*> others;
        for (SimStream* s : run) if (s != need) others.push_back(s);
        pick = others.empty() ? need : any(others);
    } else pick = any(run);
    step(pick);
    return true;
}
static void stuck(const char* what) {
    if (g_fault.empty()) g_fault = std::string("gfx950sim: deadlock: ") + what + " waits on an event whose record can never run";
    fprintf(stderr, "%s\n", g_fault.c_str());
    for (SimStream* s : A().streams) s->q.clear();
    A().pending.clear();
}
static void drain_stream(SimStream* s) { while (!s->q.empty()) if (!advance(s, A().policy)) { stuck("a stream"); break; } }
static void wait_ticket(uint64_t t) {
    for (;;) {
        auto it = A().pending.find(t);
        if (it == A().pending.end()) break;
        if (!advance(it->second, A().policy)) { stuck("an event synchronisation"); break; }
    }
}
static void drain_all() {
    for (;;) {
        bool left = false;
        for (SimStream* s : A().streams) left |= !s->q.empty();
        if (!left) break;
        if (!advance(nullptr, 0)) { stuck("a device synchronisation"); break; }
    }
}
// the legacy null stream: an operation on it starts after everything already queued on the blocking streams
static void drain_blocking() {
    std::vector<SimStream*> b;
    for (SimStream* s : A().streams) if (s->blocking) b.push_back(s);
    std::shuffle(b.begin(), b.end(), A().rng);
    for (SimStream* s : b) drain_stream(s);
}
static hipError_t submit(hipStream_t h, Op& op) {
    if (!A().on) { exec_op(op); return sync_status(); }
    if (!h) { drain_blocking(); exec_op(op); return sync_status(); }
    ((SimStream*)h)->q.push_back(op);
    return hipSuccess;
}

extern "C" {

hipError_t hipLaunchKernel(const void* func, dim3 grid, dim3 block, void** args, size_t shmem, hipStream_t stream) {
    Lock lk(g_mu);
    Op op;
    hipError_t e = prepare_launch(func, grid, block, args, shmem, op);
    if (e != hipSuccess) return e;
    return submit(stream, op);
}

// HIPSIM_LDS_LIMIT=<bytes>: a runtime that grants less dynamic LDS per workgroup than gfx950 has (tests of the product's fallbacks)
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute a, int v) {
    static const int limit = env_int("HIPSIM_LDS_LIMIT", 160 << 10);
    if (a == hipFuncAttributeMaxDynamicSharedMemorySize && v > limit) return fail(hipErrorInvalidValue);
    return hipSuccess;
}
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : fail(hipErrorInvalidDevice); }
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_tR0600* p, int d) {
    if (d != 0) return fail(hipErrorInvalidDevice);
    memset(p, 0, sizeof *p);
    snprintf(p->name, sizeof p->name, "gfx950sim (CPU interpreter)");
    snprintf(p->gcnArchName, sizeof p->gcnArchName, "gfx950:sramecc+:xnack-");
    p->totalGlobalMem = (size_t)env_int("HIPSIM_MEM_GB", 16) << 30;
    p->sharedMemPerBlock = 64 << 10;
    p->maxSharedMemoryPerMultiProcessor = 160 << 10;
    p->sharedMemPerBlockOptin = 160 << 10;
    p->regsPerBlock = 65536;
    p->warpSize = 64;
    p->maxThreadsPerBlock = 1024;
    p->maxThreadsDim[0] = p->maxThreadsDim[1] = p->maxThreadsDim[2] = 1024;
    p->maxGridSize[0] = p->maxGridSize[1] = p->maxGridSize[2] = 2147483647;
    p->clockRate = 2400000;
    p->memoryClockRate = 2000000;
    p->memoryBusWidth = 8192;
    p->multiProcessorCount = env_int("HIPSIM_CUS", 8);
    p->l2CacheSize = 4 << 20;
    p->maxThreadsPerMultiProcessor = 2048;
    p->major = 9; p->minor = 5;
    p->concurrentKernels = 1;
    p->asyncEngineCount = 2;
    p->canMapHostMemory = 1;
    return hipSuccess;
}
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t a, int) {
    switch (a) {
        case hipDeviceAttributeMultiprocessorCount: *v = env_int("HIPSIM_CUS", 8); break;
        case hipDeviceAttributeWarpSize: *v = 64; break;
        case hipDeviceAttributeMaxSharedMemoryPerBlock: *v = 64 << 10; break;
        default: *v = 0; break;
    }
    return hipSuccess;
}
const char* hipGetErrorString(hipError_t e) {
    switch (e) {
        case hipSuccess: return "no error";
        case hipErrorLaunchFailure: return g_fault.empty() ? "unspecified launch failure" : g_fault.c_str();
        case hipErrorInvalidValue: return "invalid argument";
        case hipErrorOutOfMemory: return "out of memory";
        case hipErrorInvalidDeviceFunction: return "invalid device function";
        case hipErrorInvalidDevice: return "invalid device ordinal";
        default: return "gfx950sim: error";
    }
}
const char* hipGetErrorName(hipError_t e) { return hipGetErrorString(e); }
hipError_t hipGetLastError() { hipError_t e = g_last; g_last = hipSuccess; return e; }
hipError_t hipPeekAtLastError() { return g_last; }

hipError_t hipMalloc(void** p, size_t n) {
    Lock lk(g_mu);
    uint64_t cap = (uint64_t)env_int("HIPSIM_MEM_GB", 16) << 30;
    if (g_allocated + n > cap) { *p = nullptr; return fail(hipErrorOutOfMemory); }
    *p = dev_alloc(n, false);
    return *p ? hipSuccess : fail(hipErrorOutOfMemory);
}
hipError_t hipFree(void* p) { Lock lk(g_mu); if (A().on && p) drain_all(); return dev_free(p); }      // hipFree synchronises the device
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { Lock lk(g_mu); *p = dev_alloc(n, true); g_allocated -= n ? n : 1; return *p ? hipSuccess : fail(hipErrorOutOfMemory); }
hipError_t hipHostFree(void* p) {
    Lock lk(g_mu);
    if (A().on && p) drain_all();
    auto it = g_allocs.find((uint64_t)(uintptr_t)p);
    if (it != g_allocs.end()) g_allocated += it->second.size;
    return dev_free(p);
}
hipError_t hipMemGetInfo(size_t* fr, size_t* tot) {
    Lock lk(g_mu);
    uint64_t cap = (uint64_t)env_int("HIPSIM_MEM_GB", 16) << 30;
    *tot = cap; *fr = cap > g_allocated ? cap - g_allocated : 0;
    return hipSuccess;
}
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t n, hipMemcpyKind kind, hipStream_t stream) {
    Lock lk(g_mu);
    if (!copy_ranges_ok(dst, src, n, kind)) return fail(hipErrorInvalidValue);
    Op op; op.kind = Op::COPY; op.dst = dst; op.src = src; op.n = n; op.ck = kind;
    return submit(stream, op);
}
hipError_t hipMemcpy(void* dst, const void* src, size_t n, hipMemcpyKind kind) { return hipMemcpyAsync(dst, src, n, kind, nullptr); }   // null stream, and the host waits
hipError_t hipMemsetAsync(void* dst, int v, size_t n, hipStream_t stream) {
    Lock lk(g_mu);
    if (!check_dev(dst, n, "memset")) return fail(hipErrorInvalidValue);
    Op op; op.kind = Op::SET; op.dst = dst; op.val = v; op.n = n;
    return submit(stream, op);
}
hipError_t hipMemset(void* dst, int v, size_t n) { return hipMemsetAsync(dst, v, n, nullptr); }

hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags) {
    Lock lk(g_mu);
    SimStream* st = new SimStream{(flags & hipStreamNonBlocking) == 0, {}};
    A().streams.insert(st);
    *s = (hipStream_t)st;
    return hipSuccess;
}
hipError_t hipStreamCreate(hipStream_t* s) { return hipStreamCreateWithFlags(s, 0); }
hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned flags, int) { return hipStreamCreateWithFlags(s, flags); }
hipError_t hipStreamDestroy(hipStream_t s) {
    Lock lk(g_mu);
    SimStream* st = (SimStream*)s;
    if (!st || !A().streams.count(st)) return fail(hipErrorInvalidValue);
    drain_stream(st);                        // queued work completes; the handle is gone at once
    A().streams.erase(st);
    delete st;
    return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t s) {
    Lock lk(g_mu);
    if (A().on) { if (s) drain_stream((SimStream*)s); else drain_blocking(); }
    return sync_status();
}
hipError_t hipStreamQuery(hipStream_t s) {
    Lock lk(g_mu);
    if (A().on && s && !((SimStream*)s)->q.empty()) { advance(nullptr, 0); if (!((SimStream*)s)->q.empty()) return hipErrorNotReady; }
    return sync_status();
}
hipError_t hipDeviceSynchronize() { dump_stats(); Lock lk(g_mu); if (A().on) drain_all(); return sync_status(); }
hipError_t hipDeviceReset() { return hipSuccess; }

hipError_t hipEventCreate(hipEvent_t* e) { *e = (hipEvent_t) new SimEvent{std::chrono::steady_clock::now(), false, 0}; return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) {
    Lock lk(g_mu);
    for (SimStream* s : A().streams) for (Op& op : s->q) if (op.ev == (SimEvent*)e) op.ev = nullptr;     // a queued record still completes its ticket
    delete (SimEvent*)e;
    return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t stream) {
    Lock lk(g_mu);
    Op op; op.kind = Op::REC; op.ev = (SimEvent*)e;
    if (A().on) {
        op.ticket = op.ev->ticket = ++A().next_ticket;
        if (stream) A().pending[op.ticket] = (SimStream*)stream;
    }
    (void)submit(stream, op);
    return hipSuccess;
}
// the wait is for the record that was the event's latest when this call was made (none yet: no wait)
hipError_t hipStreamWaitEvent(hipStream_t stream, hipEvent_t e, unsigned) {
    Lock lk(g_mu);
    if (!A().on) return hipSuccess;
    uint64_t t = ((SimEvent*)e)->ticket;
    if (!t || !A().pending.count(t)) return hipSuccess;
    if (!stream) { drain_blocking(); wait_ticket(t); return hipSuccess; }
    Op op; op.kind = Op::WAIT; op.ticket = t;
    ((SimStream*)stream)->q.push_back(op);
    return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t e) { Lock lk(g_mu); if (A().on) wait_ticket(((SimEvent*)e)->ticket); return sync_status(); }
hipError_t hipEventQuery(hipEvent_t e) {
    Lock lk(g_mu);
    if (A().on && A().pending.count(((SimEvent*)e)->ticket)) { advance(nullptr, 0); if (A().pending.count(((SimEvent*)e)->ticket)) return hipErrorNotReady; }
    return sync_status();
}
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
    Lock lk(g_mu);
    auto* x = (SimEvent*)a; auto* y = (SimEvent*)b;
    if (A().on && (A().pending.count(x->ticket) || A().pending.count(y->ticket))) return hipErrorNotReady;
    double d = std::chrono::duration<double, std::milli>(y->t - x->t).count();
    *ms = (float)(d > 1e-6 ? d : 1e-6);
    return hipSuccess;
}

// diagnostics for tools and tests
const char* hipsim_fault() { return g_fault.c_str(); }
void hipsim_clear_fault() { Lock lk(g_mu); g_fault.clear(); g_last = hipSuccess; }
void hipsim_dump_stats() { dump_stats(); }
unsigned long long hipsim_live_allocations() { Lock lk(g_mu); return g_allocs.size(); }

}  // extern "C"
