"""The COMPILED gfx950 kernels, checked on a machine without a GPU (round 5: GPU access for this repository has been closed
from outside the build since the middle of round 4).  tools/gfx950sim interprets the code objects inside
swipe_amd/libswipe_amd.so - the binary that ships - behind a stand-in for the HIP runtime, bounds-checks every device
access and counts wave instructions.  These tests run a bounded slice of the `-m gpu` parity suite that way (the whole
suite under the interpreter is profiles/r05_sim_gpu_suite.txt), pin the interpreter against itself (scalar definitions vs
the AVX-512 fast forms) and against a deliberate out-of-bounds kernel, and keep the settled drain experiment settled.

Nothing here replaces hardware: `-m gpu` on an MI355X remains the parity gate; this is what can be shown meanwhile, and what
let items 4-8 of VERDICT r4 be developed against real kernel executions instead of blind."""
import json
import os
import re
import subprocess
import sys

import pytest

from conftest import ROOT

SIM = os.path.join(ROOT, "tools", "gfx950sim")
RUN = os.path.join(SIM, "run.sh")
HIPCC = "/opt/rocm/bin/hipcc"
pytestmark = pytest.mark.skipif(not (os.path.exists("/opt/rocm/lib/llvm/bin/clang++") and os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump")),
                                reason="tools/gfx950sim needs ROCm's clang++ and llvm-objdump")


def _sim(cmd, timeout, **env):
    e = dict(os.environ, **{k: str(v) for k, v in env.items()})
    e.pop("LD_PRELOAD", None)
    return subprocess.run([RUN] + cmd, capture_output=True, text=True, timeout=timeout, env=e, cwd=ROOT)


GOLDEN_SLICE = ("scores_equal_reference_for_every_sequence or hit_list_equals_reference_cli or width_escalation or "
                "cli_output_equals_reference_cli or alignment_end_points_equal or translated_scores_equal or "
                "dual_query_kernel_both_strands or empty_inputs")


def test_shipped_kernels_reproduce_the_reference_goldens_under_the_interpreter():
    """every sequence of the reference's goldens (P07327 vs 1 k + planted homologs, edge lengths, the 75 000 score,
    the asymmetric matrix, nucleotide both strands, three volumes), the width escalation 16 -> 32 -> 64, end points vs
    search16s, translated searches, and the CLI's bytes - computed by the gfx950 code objects of libswipe_amd.so"""
    r = _sim([sys.executable, "-m", "pytest", "tests/test_gpu_parity.py", "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider", "-k", GOLDEN_SLICE], 1500)
    tail = (r.stdout + r.stderr)[-1500:]
    assert r.returncode == 0, tail
    m = re.search(r"(\d+) passed", r.stdout)
    assert m and int(m.group(1)) >= 40 and "failed" not in r.stdout, tail


ROUND5_SLICE = [
    # nucleotide volumes and masked aliases through the pipelined open, searched while they load
    ("tests/test_gpu_loading.py", "nucleotide_loader_edge_cases or masked_alias_streams_in or search_before_wait_on_a_corrupt_volume"),
    # inclusion sets on shards over their HBM budget: the reference's masked / taxid goldens through the CLI
    ("tests/test_gpu_group.py", "masks_and_taxid_lists_with_an_hbm_budget and (masked_taxlist or plain_taxid)"),
    # the re-queue behind the first pass, a wave and a block of four waves per sequence
    ("tests/test_gpu_parity.py", "requeue_by_batches_and_by_wave and 600"),
    # .nsq ambiguity tables in both forms against the reference's own output (tests/golden/ntamb.json): old reader, loader, budget
    ("tests/test_gpu_parity.py", "cli_nucleotide_ambiguity_tables_equal_reference_cli"),
    # round 6: both forms of the device-driven re-queue pinned by the option that selects them
    ("tests/test_gpu_parity.py", "both_device_requeue_forms and (129 or 256 or 257 or 1024 or 1025)"),
    # round 6: the bound build with sequences back to back, twin profile on and off (8-lane chains, 5 sets per item), one- and two-query
    ("tests/test_gpu_parity.py", "bound_build_with_sequences_back_to_back and 8-5"),
    ("tests/test_gpu_parity.py", "two_query_kernel_with_sequences_back_to_back and 16"),
    # round 6: ambiguity runs that overlap, in file order (old reader, device unpack, budgeted shard); the budgeted open straight from the files
    ("tests/test_gpu_loading.py", "overlap_are_applied_in_file_order or budgeted_open_fills"),
]


def test_round_5_paths_under_the_interpreter():
    """what round 5 added, on the shipped kernels without a GPU: .nsq entries unpacked on the device (empty sequences, both
    ambiguity table forms, an entry larger than a staging chunk), an OID mask through the loader, a corrupt volume refused
    by a search that follows the loader, taxid lists and masks on budgeted shards byte for byte against the reference's
    goldens, and both forms of the device-driven re-queue; round 6: both re-queue forms by name, the bound builds with sequences back
    to back (one- and two-query), overlapping ambiguity runs in file order, the budgeted open that fills its parts from the files"""
    for path, expr in ROUND5_SLICE:
        r = _sim([sys.executable, "-m", "pytest", path, "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider", "-k", expr], 1500)
        tail = (r.stdout + r.stderr)[-1500:]
        assert r.returncode == 0 and re.search(r"\d+ passed", r.stdout) and "failed" not in r.stdout, (path, expr, tail)


_SMOKE = r"""
import sys, json
sys.path.insert(0, %r)
import numpy as np
np.seterr(over="ignore")
import oracle, swipe_amd
from swipe_amd import blastdb, synth
q = blastdb.encode_protein(synth.QUERY_P07327)
res, off = swipe_amd.synth_db(1, 1500, query=q)
db = swipe_amd.Database.from_arrays(res, off)
db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
s, c = db.search(q)
db.set_option("bound", "1")
hits = db.search_topk(q, keep=50, minscore=60)[:2]
ref = oracle.search_all63(res, off, q, oracle.matrix_builtin("BLOSUM62"), 12, 1, threads=4)
print(json.dumps({"equal": bool((s == ref).all()), "sum": int(s.sum()), "hits": hits, "form": c["narrow_shifted"]}))
"""


def test_scalar_definitions_and_avx512_forms_of_the_interpreter_agree(tmp_path):
    """sim_fast.cpp (AVX-512 FP16 / BW / VBMI) must be an optimisation of sim_isa.cpp's scalar semantics, nothing else; both
    must give the oracle's scores; and the per-kernel instruction counts come out"""
    outs = []
    for fast in ("0", "1"):
        stats = str(tmp_path / f"stats{fast}.jsonl")
        r = _sim([sys.executable, "-c", _SMOKE % ROOT], 600, HIPSIM_FAST=fast, HIPSIM_STATS=stats)
        assert r.returncode == 0, (r.stdout + r.stderr)[-1500:]
        outs.append(json.loads(r.stdout.strip().splitlines()[-1]))
        rows = [json.loads(l) for l in open(stats)]
        first = max((d for d in rows if d["vop3p"] > 0), key=lambda d: d["vop3p"])
        # the first-pass kernels are packed-f16 code: that is where the instructions are
        assert first["vop3p"] > 3 * first["valu"] > 0, first
    assert outs[0]["equal"] and outs[0] == outs[1], outs


def test_interpreter_binary16_arithmetic_equals_numpy():
    """the interpreter's own ground: IEEE binary16 add / mul / fma rounded once and the 754-2019 maximum of three, scalar
    definitions and AVX-512 forms, on two million random bit patterns plus every pair of the special values (zeros of both
    signs, infinities, NaN, smallest and largest subnormal, smallest normal, largest finite) - against numpy's float16"""
    import ctypes
    import numpy as np
    subprocess.run(["make", "-s", "-C", SIM, "libhipsim.so"], check=True, capture_output=True, timeout=600)
    L = ctypes.CDLL(os.path.join(SIM, "libhipsim.so"))
    P = ctypes.POINTER(ctypes.c_uint16)
    L.hipsim_f16_op.argtypes = [ctypes.c_int, ctypes.c_int, P, P, P, P, ctypes.c_long]
    rng = np.random.default_rng(1)
    n = 1 << 21
    a, b, c = (rng.integers(0, 65536, n).astype(np.uint16) for _ in range(3))
    sp = np.array([0, 0x8000, 0x7c00, 0xfc00, 0x7e00, 0x0001, 0x8001, 0x03ff, 0x0400, 0x7bff, 0xfbff, 0x3c00, 0xbc00, 0x6800, 0x6801], np.uint16)
    k = len(sp)
    a[:k * k], b[:k * k], c[:k * k] = np.repeat(sp, k), np.tile(sp, k), np.tile(sp[::-1], k)
    af, bf, cf = (x.view(np.float16).astype(np.float64) for x in (a, b, c))
    with np.errstate(all="ignore"):
        want = {0: (af + bf).astype(np.float16).view(np.uint16), 1: (af * bf).astype(np.float16).view(np.uint16),
                2: (af * bf + cf).astype(np.float16).view(np.uint16)}
    key = lambda h: np.where(h & 0x8000, (~h) & 0xffff, h | 0x8000).astype(np.int64)          # sign-magnitude -> ordered, -0 below +0
    ka, kb, kc = key(a), key(b), key(c)
    best = np.where(ka >= kb, a, b)
    best = np.where(np.maximum(ka, kb) >= kc, best, c)
    anynan = ((a & 0x7fff) > 0x7c00) | ((b & 0x7fff) > 0x7c00) | ((c & 0x7fff) > 0x7c00)
    want[3] = np.where(anynan, 0x7e00, best).astype(np.uint16)
    for op in (0, 1, 2, 3):
        wnan = (want[op] & 0x7fff) > 0x7c00
        for fast in (0, 1):
            out = np.zeros(n, np.uint16)
            rc = L.hipsim_f16_op(op, fast, a.ctypes.data_as(P), b.ctypes.data_as(P), c.ctypes.data_as(P), out.ctypes.data_as(P), n)
            if rc:
                continue                                         # no AVX-512 FP16 on this host: the scalar form is all there is
            onan = (out & 0x7fff) > 0x7c00
            assert np.array_equal(onan, wnan) and np.array_equal(out[~wnan], want[op][~wnan]), (op, fast, int((out != want[op]).sum()))


_OOB = r"""
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void touch(int* p, int n, int at) { if ((int)threadIdx.x == 0) p[at] = n; }
int main(int argc, char** argv) {
  int* d = nullptr;
  if (hipMalloc(&d, 100 * sizeof(int)) != hipSuccess) return 2;
  hipLaunchKernelGGL(touch, dim3(1), dim3(64), 0, 0, d, 7, argc > 1 ? atoi(argv[1]) : 0);
  hipError_t e = hipDeviceSynchronize();
  std::printf("%s\n", e == hipSuccess ? "clean" : hipGetErrorString(e));
  return e == hipSuccess ? 0 : 1;
}
"""


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_interpreter_faults_on_the_first_byte_outside_an_allocation(tmp_path):
    """device memory is checked against the exact byte range of every live allocation - what red zones and compute-sanitizer
    approximate on hardware (device code had never run under a memory checker)"""
    src = tmp_path / "oob.hip"
    src.write_text("#include <cstdlib>\n" + _OOB)
    exe = str(tmp_path / "oob")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O2", str(src), "-o", exe], check=True, capture_output=True, timeout=300)
    ok = _sim([exe, "99"], 120)
    assert ok.returncode == 0 and "clean" in ok.stdout, ok.stdout + ok.stderr
    bad = _sim([exe, "100"], 120)
    assert bad.returncode == 1 and "out of bounds" in bad.stdout + bad.stderr and "0 bytes past the end of the 400-byte allocation" in bad.stdout + bad.stderr, bad.stdout + bad.stderr


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_drain_experiment_root_cause_stays_settled(tmp_path):
    """tools/ubench/drain_test.hip.  The wrong scores of the in-kernel re-queue drain are DETERMINISTIC
    under the interpreter (one wave at a time, single-wave mode included), i.e. not a race: hipcc threads the divergent
    lane-0 branch at the end of the claim loop across the back-edge and lanes 1..63 leave the loop after the first
    sequence (sw_wave_dp_experiment.cuh has the ISA).  With no divergent branch at the latch (-DSWA_DRAIN_FIXED) the same
    inlined drain is right.  Pinned: FIXED must be right; what the unfixed inline build does is reported, not asserted -
    another compiler may well not thread it."""
    inc = ["-I" + os.path.join(ROOT, "swipe_amd", "csrc"), "-I" + os.path.join(ROOT, "tools", "ubench")]
    src = os.path.join(ROOT, "tools", "ubench", "drain_test.hip")
    res = {}
    for name, flags in (("fixed", ["-DSWA_DRAIN_INLINE", "-DSWA_DRAIN_FIXED"]), ("inline", ["-DSWA_DRAIN_INLINE"])):
        exe = str(tmp_path / ("drain_" + name))
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3"] + inc + flags + [src, "-o", exe], check=True, capture_output=True, timeout=600)
        for mode in ("0", "3"):
            r = _sim([exe, "750", mode], 300, HIPSIM_THREADS=1)
            m = re.search(r"wrong (\d+) of 64", r.stdout)
            assert m, r.stdout + r.stderr
            res[(name, mode)] = int(m.group(1))
    print("drain experiment under the interpreter (wrong of 64):", res)
    assert res[("fixed", "0")] == 0 and res[("fixed", "3")] == 0, res
    assert res[("inline", "0")] == res[("inline", "3")], res        # whatever it is, it does not depend on how many waves take part


_STREAM_ORDER = r"""
import ctypes, sys
H = ctypes.CDLL(sys.argv[1])
edge = sys.argv[2] == "edge"
N = 4096
vp = ctypes.c_void_p
for f in ("hipMalloc", "hipHostMalloc", "hipStreamCreateWithFlags", "hipEventCreate", "hipMemsetAsync", "hipMemcpyAsync", "hipEventRecord",
          "hipStreamWaitEvent", "hipStreamSynchronize", "hipDeviceSynchronize", "hipMemset", "hipMemcpy"):
    getattr(H, f).restype = ctypes.c_int
H.hipMemsetAsync.argtypes = [vp, ctypes.c_int, ctypes.c_size_t, vp]
H.hipMemcpyAsync.argtypes = [vp, vp, ctypes.c_size_t, ctypes.c_int, vp]
H.hipMemcpy.argtypes = [vp, vp, ctypes.c_size_t, ctypes.c_int]
H.hipMemset.argtypes = [vp, ctypes.c_int, ctypes.c_size_t]
H.hipEventRecord.argtypes = [vp, vp]
H.hipStreamWaitEvent.argtypes = [vp, vp, ctypes.c_uint]
H.hipStreamSynchronize.argtypes = [vp]
dev, out, a, b, ev = vp(), vp(), vp(), vp(), vp()
assert H.hipMalloc(ctypes.byref(dev), N) == 0 and H.hipHostMalloc(ctypes.byref(out), N, 0) == 0
assert H.hipStreamCreateWithFlags(ctypes.byref(a), 1) == 0 and H.hipStreamCreateWithFlags(ctypes.byref(b), 1) == 0      # hipStreamNonBlocking
assert H.hipEventCreate(ctypes.byref(ev)) == 0
assert H.hipMemset(dev, 0x22, N) == 0
ctypes.memset(out, 0, N)
# producer on stream a, consumer on stream b: without the event edge nothing orders the copy behind the memset
H.hipMemsetAsync(dev, 0x11, N, a)
if edge:
    H.hipEventRecord(ev, a)
    H.hipStreamWaitEvent(b, ev, 0)
H.hipMemcpyAsync(out, dev, N, 2, b)          # device to host
assert H.hipStreamSynchronize(b) == 0
first = ctypes.string_at(out, N)
H.hipDeviceSynchronize()
# a host source is read when the copy EXECUTES: rewriting it before the synchronisation changes what arrives
src = ctypes.create_string_buffer(b"\x33" * N, N)
H.hipMemcpyAsync(dev, src, N, 1, a)          # host to device
ctypes.memset(src, 0x44, N)
H.hipStreamSynchronize(a)
H.hipMemcpy(out, dev, N, 2)
second = ctypes.string_at(out, N)
print("consumer", "fresh" if first == b"\x11" * N else "stale", "source", "call" if second == b"\x33" * N else "execution")
"""


def test_interpreter_async_mode_shows_a_missing_event_edge(tmp_path):
    """HIPSIM_ASYNC=<seed> defers launches, memsets and async copies and runs them in a seeded order that
    only HIP's own ordering promises constrain.  A consumer stream that does not wait for the producer's event reads stale
    data under at least one policy (the lazy one leaves the producer's stream untouched); with hipEventRecord +
    hipStreamWaitEvent every policy sees the producer's bytes; an async copy reads its host source at execution time.
    Without HIPSIM_ASYNC everything runs at the call, as before."""
    subprocess.run(["make", "-s", "-C", SIM, "libhipsim.so"], check=True, timeout=600)
    lib = os.path.join(SIM, "libhipsim.so")
    script = tmp_path / "order.py"
    script.write_text(_STREAM_ORDER)

    def run(mode, seed):
        env = dict(os.environ)
        env.pop("LD_PRELOAD", None)
        env.pop("HIPSIM_ASYNC", None)
        if seed is not None:
            env["HIPSIM_ASYNC"] = str(seed)
        r = subprocess.run([sys.executable, str(script), lib, mode], capture_output=True, text=True, timeout=120, env=env)
        assert r.returncode == 0, r.stdout + r.stderr
        return r.stdout.split()

    assert run("noedge", None) == ["consumer", "fresh", "source", "call"]          # synchronous streams hide both
    with_edge = [run("edge", s) for s in (0, 1, 2, 3, 4, 5)]
    assert all(w[1] == "fresh" and w[3] == "execution" for w in with_edge), with_edge
    without = [run("noedge", s)[1] for s in (0, 1, 2, 3, 4, 5)]
    assert "stale" in without and without[1] == "stale", without                   # seed % 3 == 1: the lazy policy


_CNDMASK_DPP = r"""
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32;
template <int G> __global__ void k(const u32* in, u32* fused, u32* plain)
{
  const int lane = threadIdx.x & 63;
  const u32 cur = in[lane], fresh = in[64 + lane];
  constexpr u32 heads = G == 8 ? 0x01010101u : G == 4 ? 0x11111111u : 0x55555555u;
  u32 out;
  asm("s_mov_b32 vcc_lo, %3\n\ts_mov_b32 vcc_hi, %3\n\tv_cndmask_b32_dpp %0, %1, %2, vcc row_ror:1 row_mask:0xf bank_mask:0xf"
      : "=v"(out) : "v"(cur), "v"(fresh), "n"(heads) : "vcc");
  fused[lane] = out;
  const u32 rot = (u32)__shfl((int)cur, (lane & 48) | ((lane + 15) & 15));       // lane l of a row of 16 takes lane l - 1, lane 0 takes lane 15
  plain[lane] = (lane % G) == 0 ? fresh : rot;
}
int main()
{
  u32 h[128], *d, *a, *b, ra[64], rb[64];
  for (int i = 0; i < 128; ++i) h[i] = 0x9E3779B9u * (i + 1);
  hipMalloc(&d, sizeof h); hipMalloc(&a, 256); hipMalloc(&b, 256);
  hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
  int bad = 0;
  for (int g = 0; g < 3; ++g) {
    if (g == 0) hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, d, a, b);
    if (g == 1) hipLaunchKernelGGL(k<4>, dim3(1), dim3(64), 0, 0, d, a, b);
    if (g == 2) hipLaunchKernelGGL(k<8>, dim3(1), dim3(64), 0, 0, d, a, b);
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    hipMemcpy(ra, a, 256, hipMemcpyDeviceToHost); hipMemcpy(rb, b, 256, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) {
      const u32 want = (l % (2 << g)) == 0 ? h[64 + l] : h[(l & 48) | ((l + 15) & 15)];
      bad += ra[l] != want || rb[l] != want;
    }
  }
  std::printf("mismatches %d\n", bad);
  return bad != 0;
}
"""


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_interpreter_cndmask_through_dpp_equals_rotate_then_select(tmp_path):
    """round 6: the residue shift register of a chain advances with ONE instruction, v_cndmask_b32_dpp (sw_common.cuh
    chain_advance: vcc ? fresh : row_ror:1(cur)), written out in inline assembly.  The interpreter's definition of it against the
    host's statement of what it must do and against the two-instruction form through ds_bpermute, for chains of 2, 4 and 8 lanes"""
    src = tmp_path / "cnd.hip"
    src.write_text(_CNDMASK_DPP)
    exe = str(tmp_path / "cnd")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O2", str(src), "-o", exe], check=True, capture_output=True, timeout=300)
    r = _sim([exe], 120)
    assert r.returncode == 0 and "mismatches 0" in r.stdout, r.stdout + r.stderr


def test_bench_py_runs_end_to_end_on_the_interpreter():
    """bench.py has changed in rounds 5 and 6 without a GPU to run it on (budgeted open in cold_open, the round-3 A/B block).
    tools/bench_dryrun.py runs the driver's script itself on the interpreter (torch.cuda's three calls stubbed): here a bounded
    run - headline step, oracle verification, cpu_baseline against the compiled reference, the cold-open section with the
    pipelined and the budgeted open - and the line must carry the blocks the contract names.  Numbers mean nothing here."""
    r = _sim([sys.executable, "tools/bench_dryrun.py", "--nseq", "20000", "--steps", "1", "--warmup", "0", "--no-live-traffic", "--quick",
              "--secondary-nt-nseq", "4000", "--secondary-protein-nseq", "8000"], 1500)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and lines, (r.stdout + r.stderr)[-2000:]
    line = json.loads(lines[-1])
    assert line["unit"] == "GCUPS" and line["value"] > 0 and line["n_gpus"] == 1 and line["verified_vs_oracle"] >= 1
    assert line["roofline"]["bound"] == "hbm" and line["roofline"]["peak"] == 8000.0 and line["valu_roofline"]
    assert line["cpu_baseline"]["kind"] in ("reference", "port") and line["cpu_baseline"]["value"] > 0
    cold = line["cold_open"]
    assert cold["open_s"] is not None and cold["budgeted_open_s"] is not None, cold      # (at this size the budget holds the whole shard)
    assert len(line["secondary"]) >= 3


_TWIN_FALLBACK = r"""
import sys
sys.path.insert(0, %r)
import numpy as np
np.seterr(over="ignore")
import oracle, swipe_amd
from swipe_amd import blastdb, synth
q = blastdb.encode_protein(synth.QUERY_P07327)
res, off = swipe_amd.synth_db(1, 6000, query=q)
ref = oracle.search_all63(res, off, q, oracle.matrix_builtin("BLOSUM62"), 12, 1, threads=8)
db = swipe_amd.Database.from_arrays(res, off); db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
db.set_option("bound", 1); db.set_option("concat", 4); db.set_option("concat_tail", 50)
order = sorted(((int(s), i) for i, s in enumerate(ref) if s >= 70), key=lambda t: (-t[0], -t[1]))[:100]
hits, tot, obv, c = db.search_topk(q, keep=100, minscore=70)
print("HITS", "ok" if hits == [(i, s) for s, i in order] and c["narrow_shifted"] == 8 else "WRONG")
"""


def test_twin_profile_build_falls_back_where_the_runtime_grants_less_lds():
    """round 6: the bound build at two waves per SIMD asks for 96-128 KB of dynamic LDS per block (two copies of the profile).  No
    hardware has seen that request yet; a runtime that refuses it must get the one-copy form, not a failed search.  The
    interpreter's runtime plays both: all of gfx950's 160 KB, and 64 KB (HIPSIM_LDS_LIMIT) - same hits, different kernel"""
    seen = {}
    for limit in (160 << 10, 64 << 10):
        r = _sim([sys.executable, "-c", _TWIN_FALLBACK % ROOT], 600, HIPSIM_LDS_LIMIT=limit, HIPSIM_TRACE=1)
        assert r.returncode == 0 and "HITS ok" in r.stdout, (r.stdout + r.stderr)[-1500:]
        seen[limit] = re.findall(r"launch _Z23swa_narrow_bound_kernelILi47ELi2ELi8ELi16ELb0ELb([01])E", r.stderr)
    assert seen[160 << 10] and set(seen[160 << 10]) == {"1"} and seen[64 << 10] and set(seen[64 << 10]) == {"0"}, seen
