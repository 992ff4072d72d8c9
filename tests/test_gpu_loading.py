"""A database that streams into HBM (swa_db_open_async, csrc/sw_loading.inc): the reference maps its sequence files a chunk
at a time (db_mapsequences, database.cc:1082-1131) so its first search overlaps the page-in; here the first search runs
over the parts of the shard as the loader publishes them.  Whatever arrives when, hit list, totalhits and every score
must be those of the resident shard - and of the oracle.

Every test drives the C ABI (through swipe_amd/_lib.py); the loader is slowed down with its own test knobs
(SWA_LOAD_PART / SWA_LOAD_CHUNK / SWA_LOAD_DELAY_MS, read when the open begins) so that the searches provably start while
parts are still missing."""
import os
import struct
import sys

import numpy as np
import pytest

import oracle
import swipe_amd
from swipe_amd import blastdb, synth
from conftest import ROOT

pytestmark = pytest.mark.gpu

Q = blastdb.encode_protein(synth.QUERY_P07327)
M = None


def _matrix():
    global M
    if M is None:
        M = swipe_amd.matrix_builtin("BLOSUM62")
    return M


def _expected_topk(scores, keep, minscore, first=0):
    order = sorted(((int(s), i + first) for i, s in enumerate(scores) if s >= minscore), key=lambda t: (-t[0], -t[1]))
    return [(i, s) for s, i in order[:keep]], int((scores >= minscore).sum())


class _Env:
    """loader knobs for the opens inside the block"""

    def __init__(self, **kv):
        self.kv = {k: str(v) for k, v in kv.items()}

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        os.environ.update(self.kv)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.fixture(scope="module")
def volumes(tmp_path_factory):
    """60 000 synthetic proteins (planted homologs of the query) as one volume and as two volumes behind an alias"""
    d = tmp_path_factory.mktemp("loading")
    res, off = swipe_amd.synth_db(5, 60_000, query=Q)
    one = str(d / "one")
    swipe_amd.write_blastdb(one, res, off, first_id=0)
    cut = 23_457
    swipe_amd.write_blastdb(str(d / "v0"), res[:off[cut]], off[:cut + 1], first_id=0)
    swipe_amd.write_blastdb(str(d / "v1"), res[off[cut]:], off[cut:] - off[cut], first_id=cut)
    both = str(d / "both")
    blastdb.write_alias(both, [str(d / "v0"), str(d / "v1")])
    ref = oracle.search_all63(res, off, Q, oracle.matrix_builtin("BLOSUM62"), 12, 1, threads=os.cpu_count() or 1)
    return {"one": one, "both": both, "res": res, "off": off, "ref": ref, "dir": d}


SLOW = dict(SWA_LOAD_PART=2 << 20, SWA_LOAD_CHUNK=1 << 20, SWA_LOAD_DELAY_MS=15)     # ~10 parts, ~20 chunks, >= 0.3 s


@pytest.mark.parametrize("which", ["one", "both"])
def test_search_while_loading_gives_the_resident_results(volumes, which):
    want_hits, want_tot = _expected_topk(volumes["ref"], 100, 60)
    with _Env(**SLOW):
        db = swipe_amd.Database.open(volumes[which], wait=False)
    try:
        before = db.load_progress()
        assert before["parts_total"] >= 8 and before["parts_ready"] < before["parts_total"], before
        db.set_scoring(_matrix(), 11, 1)
        # top-K with the bound first pass, part by part
        hits, tot, obv, c = db.search_topk(Q, keep=100, minscore=60)
        assert c["loading_parts"] == before["parts_total"], c
        assert (hits, tot, obv) == (want_hits, want_tot, 0)
        # a second search, maybe still loading, maybe not: same answer; then every score
        hits2, tot2, _, c2 = db.search_topk(Q, keep=100, minscore=60)
        assert (hits2, tot2) == (want_hits, want_tot)
        db.wait()
        after = db.load_progress()
        assert after == {"bytes_loaded": 0, "bytes_total": 0, "parts_ready": 0, "parts_total": 0}
        scores, c3 = db.search(Q)
        assert c3["loading_parts"] == 0 and np.array_equal(scores, volumes["ref"])
        hits3, tot3, _, c4 = db.search_topk(Q, keep=100, minscore=60)
        assert c4["loading_parts"] == 0 and (hits3, tot3) == (want_hits, want_tot)
        info = db.info()
        assert info["seqcount"] == 60_000 and info["symcount"] == int(volumes["off"][-1])
    finally:
        db.close()


def test_all_scores_while_loading(volumes):
    with _Env(**SLOW):
        db = swipe_amd.Database.open(volumes["both"], wait=False)
    try:
        db.set_scoring(_matrix(), 11, 1)
        scores, c = db.search(Q)
        assert c["loading_parts"] >= 8 and np.array_equal(scores, volumes["ref"])
        # the exact first pass, forced, on whatever is still loading or not
        db.set_option("bound", 0)
        hits, tot, _, c = db.search_topk(Q, keep=50, minscore=45)
        assert (hits, tot) == _expected_topk(volumes["ref"], 50, 45)
    finally:
        db.close()


def test_short_and_long_queries_on_a_loading_handle(volumes):
    """one-lane builds go part by part too; a query beyond the single-pass kernels waits for the shard"""
    res, off = volumes["res"], volumes["off"]
    Mo = oracle.matrix_builtin("BLOSUM62")
    for qlen, parts_expected in ((9, True), (40, True), (200, True), (1100, False)):
        q = (np.arange(qlen) * 7 % 20 + 1).astype(np.uint8) if qlen != 200 else res[off[777]:off[777] + 200]
        want = oracle.search_all63(res, off, q, Mo, 12, 1, threads=os.cpu_count() or 1)
        with _Env(**SLOW):
            db = swipe_amd.Database.open(volumes["one"], wait=False)
        try:
            db.set_scoring(_matrix(), 11, 1)
            scores, c = db.search(q)
            assert (c["loading_parts"] > 0) == parts_expected, (qlen, c)
            assert np.array_equal(scores, want), qlen
        finally:
            db.close()


def test_entry_points_that_need_the_whole_shard_wait(volumes):
    res, off, ref = volumes["res"], volumes["off"], volumes["ref"]
    top = [i for i, _ in _expected_topk(ref, 5, 60)[0]]
    with _Env(**SLOW):
        db = swipe_amd.Database.open(volumes["one"], wait=False)
    try:
        db.set_scoring(_matrix(), 11, 1)
        sc, pos, qpos = db.search_endpoints(Q, top)
        assert [int(s) for s in sc] == [int(ref[i]) for i in top]
        assert db.load_progress()["parts_total"] == 0            # it had to wait
        seq = db.sequence(top[0])
        assert np.array_equal(seq, res[off[top[0]]:off[top[0] + 1]])
    finally:
        db.close()


def test_pipelined_open_equals_the_old_reader(volumes):
    """swa_db_open (= async + wait) against SWA_PIPELINED=0, whole and a range that starts and ends inside volumes"""
    for first, last in ((0, -1), (20_001, 41_234)):
        got = []
        for env in (dict(), dict(SWA_PIPELINED=0), dict(SWA_LOAD_PART=3 << 20, SWA_LOAD_CHUNK=1 << 20, SWA_LOAD_THREADS=3)):
            with _Env(**env):
                db = swipe_amd.Database.open(volumes["both"], first_seqno=first, last_seqno=last)
            db.set_scoring(_matrix(), 11, 1)
            scores, _ = db.search(Q)
            info = db.info()
            hits = db.search_topk(Q, keep=30, minscore=50)[:2]
            got.append((scores, info, hits))
            db.close()
        hi = 60_000 if last < 0 else last + 1
        assert np.array_equal(got[0][0], volumes["ref"][first:hi])
        for other in got[1:]:
            assert np.array_equal(got[0][0], other[0]) and got[0][2] == other[2]
            a, b = dict(got[0][1]), dict(other[1])
            a.pop("hbm_bytes"), b.pop("hbm_bytes")
            assert a == b
        assert got[0][1]["first_seqno"] == first and got[0][1]["total_seqcount"] == 60_000


def test_residue_code_out_of_range_is_reported(volumes, tmp_path):
    """a .psq byte >= 32 would index outside the LDS profile: the old reader's OR over the bytes runs on the device now"""
    import shutil
    for ext in ("pin", "psq", "phr"):
        shutil.copy(volumes["one"] + "." + ext, str(tmp_path / ("bad." + ext)))
    with open(str(tmp_path / "bad.psq"), "r+b") as f:
        f.seek(int(volumes["off"][30_000]) + 30_000 + 5)
        f.write(bytes([77]))
    with pytest.raises(swipe_amd.SwaError, match="out of range"):
        swipe_amd.Database.open(str(tmp_path / "bad"))
    with _Env(SWA_LOAD_PART=2 << 20, SWA_LOAD_CHUNK=1 << 20):
        db = swipe_amd.Database.open(str(tmp_path / "bad"), wait=False)
    try:
        with pytest.raises(swipe_amd.SwaError, match="out of range"):
            db.wait()
        db.set_scoring(_matrix(), 11, 1)
        with pytest.raises(swipe_amd.SwaError, match="load failed"):
            db.search_topk(Q, keep=10, minscore=50)
    finally:
        db.close()


@pytest.mark.late
def test_search_before_wait_on_a_corrupt_volume_fails(volumes, tmp_path):
    """a search that FOLLOWS the loader must not return scores computed from residue codes >= 32 (they index outside
    the LDS profile); the flag the unterminate kernels OR together comes back with the search's own counters"""
    import shutil
    for ext in ("pin", "psq", "phr"):
        shutil.copy(volumes["one"] + "." + ext, str(tmp_path / ("bad2." + ext)))
    with open(str(tmp_path / "bad2.psq"), "r+b") as f:
        f.seek(int(volumes["off"][59_000]) + 59_000 + 5)          # in the LAST part: every earlier part is searched first
        f.write(bytes([77]))
    for entry in ("topk", "all"):
        with _Env(**SLOW):
            db = swipe_amd.Database.open(str(tmp_path / "bad2"), wait=False)
        try:
            db.set_scoring(_matrix(), 11, 1)
            # hardware: an LDS read beyond the profile returns 0, the kernel ends, the flag refuses the result ("load failed");
            # tools/gfx950sim: the out-of-range LDS read itself is the fault
            with pytest.raises(swipe_amd.SwaError, match="load failed|LDS access out of range"):
                if entry == "topk":
                    db.search_topk(Q, keep=10, minscore=50)
                else:
                    db.search(Q)
        finally:
            db.close()
            from conftest import interpreter_clear_fault
            interpreter_clear_fault()


def test_truncated_sequence_file_is_reported(volumes, tmp_path):
    import shutil
    for ext in ("pin", "psq", "phr"):
        shutil.copy(volumes["one"] + "." + ext, str(tmp_path / ("cut." + ext)))
    size = os.path.getsize(str(tmp_path / "cut.psq"))
    with open(str(tmp_path / "cut.psq"), "r+b") as f:
        f.truncate(size - 1_000_000)
    with pytest.raises(swipe_amd.SwaError):
        swipe_amd.Database.open(str(tmp_path / "cut"))


def test_close_while_loading(volumes):
    for _ in range(3):
        with _Env(**SLOW):
            db = swipe_amd.Database.open(volumes["one"], wait=False)
        assert db.load_progress()["parts_ready"] < db.load_progress()["parts_total"]
        db.close()                                        # stops and joins the loader; nothing hangs, nothing leaks a thread


@pytest.fixture(scope="module")
def nt_volumes(tmp_path_factory):
    """12 000 synthetic nucleotide sequences in THREE volumes behind a .nal, ~4 % of them with runs of ambiguity codes (the
    .nsq entries carry ambiguity tables behind the packed bases, database.cc:1284-1323), lengths of every residue mod 4"""
    d = tmp_path_factory.mktemp("ntloading")
    res, off = swipe_amd.synth_db(6, 12_000, protein=False)
    res = res.copy()
    rng = np.random.default_rng(3)
    n = len(off) - 1
    for s in rng.choice(n, 500, replace=False):
        L = int(off[s + 1] - off[s])
        for _ in range(int(rng.integers(1, 4))):
            if L < 8:
                break
            a = int(rng.integers(0, L - 4))
            k = int(rng.integers(1, min(40, L - a)))
            res[off[s] + a: off[s] + a + k] = int(rng.choice([15, 5, 10, 3, 12, 7, 14]))
    seqs = [res[off[i]:off[i + 1]] for i in range(n)]
    base = str(d / "nt3")
    blastdb.write_db(base, seqs, protein=False, volumes=3)
    q = synth._random_residues(21, 1, 1000, synth.residue_table_nucleotide())      # BASELINE config 4's shape: 16 lanes x 63 rows
    Mo = oracle.matrix_nucleotide(1, -3)
    cpus = os.cpu_count() or 1
    ref = (oracle.search_all63(res, off, q, Mo, 7, 2, threads=cpus), oracle.search_all63(res, off, blastdb.revcomp_nt16(q), Mo, 7, 2, threads=cpus))
    return {"base": base, "res": res, "off": off, "q": q, "ref": ref}


NT_SLOW = dict(SWA_LOAD_PART=1 << 18, SWA_LOAD_CHUNK=1 << 17, SWA_LOAD_DELAY_MS=15)


@pytest.mark.late
def test_nucleotide_volumes_stream_in(nt_volumes):
    """nucleotide volumes take the pipelined open too - the .nsq as it lies (whole entries per chunk), 2-bit
    -> one-hot nibbles and the ambiguity runs on the device, 4-bit one-sequence-per-row parts merged into the set both-strand
    searches stream.  Residues, both-strand scores and hit lists equal the old reader's and the oracle's."""
    v = nt_volumes
    q, qr = v["q"], blastdb.revcomp_nt16(v["q"])
    M = swipe_amd.matrix_nucleotide(1, -3)
    with _Env(SWA_PIPELINED=0):
        old = swipe_amd.Database.open(v["base"], symtype=0)
    with _Env(**NT_SLOW):
        new = swipe_amd.Database.open(v["base"], symtype=0, wait=False)
    try:
        p = new.load_progress()
        assert p["parts_total"] >= 3 and p["parts_ready"] < p["parts_total"], p
        for d in (old, new):
            d.set_scoring(M, 5, 2)
        # every residue of a sample of sequences, ambiguity runs included (before anything else forces the load to end)
        s1, s2, c = new.search2(q, qr)
        assert c["loading_parts"] >= 3, c                     # both strands in one pass, part by part behind the loader
        assert np.array_equal(s1, v["ref"][0]) and np.array_equal(s2, v["ref"][1])
        new.wait()
        rng = np.random.default_rng(9)
        for s in list(rng.choice(len(v["off"]) - 1, 300, replace=False)) + [0, len(v["off"]) - 2]:
            want = v["res"][v["off"][s]:v["off"][s + 1]]
            assert np.array_equal(new.sequence(int(s)), want), s
            assert np.array_equal(old.sequence(int(s)), want), s
        assert new.info()["symcount"] == old.info()["symcount"] == int(v["off"][-1])
        a, b = old.search2_topk(q, qr, keep=80, minscore=22), new.search2_topk(q, qr, keep=80, minscore=22)
        assert a[:3] == b[:3]
        o1, o2, _ = old.search2(q, qr)
        assert np.array_equal(o1, s1) and np.array_equal(o2, s2)
        # a short query takes chains of fewer than 16 lanes, which stream the pair format (built on demand): same answers
        qs = np.ascontiguousarray(q[:200])
        x, y = old.search2_topk(qs, blastdb.revcomp_nt16(qs), keep=40, minscore=18), new.search2_topk(qs, blastdb.revcomp_nt16(qs), keep=40, minscore=18)
        assert x[:3] == y[:3]
        # a single-strand search on the streamed-in shard
        assert np.array_equal(new.search(q)[0], v["ref"][0])
    finally:
        old.close()
        new.close()


_PACK_OLD = blastdb.pack_nucleotide


def _pack_nucleotide_new_format(codes):
    """blastdb.pack_nucleotide with the ambiguity table in the 64-bit form (header bit 31; entries code:4 | run-1:12 | pad:4 |
    position:44, database.cc:1284-1305) - runs of up to 4 096 bases in one entry"""
    import struct
    body, _ = _PACK_OLD(codes)
    codes = np.asarray(codes, dtype=np.uint8)
    amb = ~np.isin(codes, (1, 2, 4, 8))
    entries, i, n = [], 0, len(codes)
    while i < n:
        if amb[i]:
            j = i
            while j + 1 < n and amb[j + 1] and codes[j + 1] == codes[i] and j + 1 - i < 4096:
                j += 1
            entries.append((int(codes[i]) << 60) | ((j - i) << 48) | i)
            i = j + 1
        else:
            i += 1
    table = b"" if not entries else struct.pack(">I", 0x80000000 | (2 * len(entries))) + b"".join(struct.pack(">Q", e) for e in entries)
    return body, table


_BUDGET_OPEN = r"""
import json, os, sys, time
sys.path.insert(0, %r)
import numpy as np
np.seterr(over="ignore")
import swipe_amd
from swipe_amd import blastdb, synth

def status(key):
    for line in open("/proc/self/status"):
        if line.startswith(key + ":"):
            return int(line.split()[1]) * 1024

base, budget, mode = sys.argv[1], int(sys.argv[2]), sys.argv[3]
q = blastdb.encode_protein(synth.QUERY_P07327)
M = swipe_amd.matrix_builtin("BLOSUM62")
warm = swipe_amd.Database.from_arrays(np.array([1, 2, 3], np.uint8), np.array([0, 3], np.int64))      # the runtime's own allocations
warm.set_scoring(M, 11, 1); warm.search(q[:20]); warm.close()
out = {"rss_before": status("VmRSS")}
t0 = time.perf_counter()
db = swipe_amd.Database.open(base, hbm_budget=budget)
out["open_s"] = time.perf_counter() - t0
out["progress_at_open"] = db.load_progress()
if mode == "rss":
    db.wait()
    out["wait_s"] = time.perf_counter() - t0
    out["progress_after"] = db.load_progress()
    out["hwm_after"] = status("VmHWM")
    out["hbm_bytes"] = db.info()["hbm_bytes"]
    out["first"] = [int(x) for x in db.sequence(0)[:8]]
else:
    db.set_scoring(M, 11, 1)
    hits, tot, obv, c = db.search_topk(q, keep=30, minscore=55)        # binds every part as the loader delivers it
    out["search_s"] = time.perf_counter() - t0
    out["hits"], out["tot"] = hits, tot
    out["progress_after"] = db.load_progress()
db.close()
print(json.dumps(out))
"""


@pytest.mark.late
def test_budgeted_open_fills_the_parts_straight_from_the_files(tmp_path):
    """swa_db_open_streamed no longer reads the database into a host vector and copies that into the parts
    (2 x the database in host memory, a serial open): the call returns when the index is read and the parts are planned, a
    loader fills each part's page-locked block out of the sequence files (database.cc:1082-1131: the reference maps what it
    is about to search), and the first search binds the parts as they arrive.  Asserted on /proc/self/status of a fresh
    process: open + wait raise the resident set by at most 1.15 x the page-locked footprint (under the interpreter the
    device slots are host memory too and are allowed for)."""
    import json
    import subprocess
    from conftest import under_interpreter
    nseq = 300_000 if under_interpreter() else 3_000_000
    res, off = swipe_amd.synth_db(31, nseq, query=Q, threads=os.cpu_count() or 1)
    base = str(tmp_path / "big")
    swipe_amd.write_blastdb(base, res, off)
    want = oracle.search_all63(res[:off[40_000]], off[:40_001], Q, oracle.matrix_builtin("BLOSUM62"), 12, 1, threads=os.cpu_count() or 1)
    nsym = int(off[-1])
    first8 = [int(x) for x in res[:8]]
    budget = int((2.04 * nsym + 77 * nseq) / 4)                     # a quarter of what the shard would take resident
    # the whole database's expected hit list: from a resident shard in this process (itself checked against the oracle on a slice)
    db = swipe_amd.Database.from_arrays(res, off)
    db.set_scoring(_matrix(), 11, 1)
    scores, _ = db.search(Q)
    assert np.array_equal(scores[:40_000], want)
    exp_hits, exp_tot = _expected_topk(scores, 30, 55)
    db.close()
    del res, off, scores
    script = tmp_path / "budget_open.py"
    script.write_text(_BUDGET_OPEN % ROOT)

    def run(mode):
        r = subprocess.run([sys.executable, str(script), base, str(budget), mode], capture_output=True, text=True, timeout=3000)
        assert r.returncode == 0, r.stdout[-500:] + r.stderr[-1500:]
        return json.loads(r.stdout.strip().splitlines()[-1])

    a = run("rss")
    pinned = a["progress_after"]["bytes_total"]
    assert a["progress_after"]["parts_ready"] == a["progress_after"]["parts_total"] >= 7 and a["first"] == first8
    assert nsym < pinned < 1.1 * nsym + 64 * nseq
    grown = a["hwm_after"] - a["rss_before"] - (a["hbm_bytes"] if under_interpreter() else 0)
    assert grown <= 1.15 * pinned, (grown, pinned, a)
    assert a["progress_at_open"]["parts_ready"] < a["progress_at_open"]["parts_total"], a      # the open did not wait for the residues
    b = run("search")
    assert ([tuple(h) for h in b["hits"]], b["tot"]) == (exp_hits, exp_tot)
    print("budgeted open: %.3f s to return, %.3f s until every part is in place (%.2f GB page-locked); first search done %.3f s after the open began"
          % (a["open_s"], a["wait_s"], pinned / 1e9, b["search_s"]))


def _apply_table_in_file_order(codes, table):
    """the residues a .nsq entry stands for: its one-hot bases with the ambiguity entries applied one after the other, the last
    writer wins (database.cc:1296-1321); runs are cut at the end of the sequence"""
    out = np.asarray(codes, dtype=np.uint8).copy()
    out[~np.isin(out, (1, 2, 4, 8))] = 1                      # what the 2-bit body holds under an ambiguity code: A
    if table:
        big = (struct.unpack(">I", table[:4])[0] >> 31) != 0
        es = 8 if big else 4
        for i in range(4, len(table), es):
            v = int.from_bytes(table[i:i + es], "big")
            code, run, pos = (v >> 60, ((v >> 48) & 0xfff) + 1, v & 0xfffffffffff) if big else (v >> 28, ((v >> 24) & 15) + 1, v & 0xffffff)
            out[pos:pos + run] = code
    return out


@pytest.mark.late
def test_nucleotide_ambiguity_runs_that_overlap_are_applied_in_file_order(tmp_path, monkeypatch):
    """VERDICT r5 item 6 / ADVICE r5: the reference applies a sequence's ambiguity entries in file order, so where runs overlap the
    last one wins (database.cc:1296-1321).  Tables written back to front, with runs that overlap each other, in both table
    forms: every base from the old reader (host), from the pipelined open (swa_unpack_nt on the device) and from a shard over
    its HBM budget equals the entries applied one after the other.  (tests/golden/ntamb_overlap.json is the same thing through
    the reference's own output.)"""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_ntamb_golden as G
    rng = np.random.default_rng(29)
    acgt = np.array([1, 2, 4, 8], np.uint8)
    seqs = []
    for k in range(700):
        n = int(rng.integers(30, 900))
        sq = acgt[rng.integers(0, 4, n)]
        for _ in range(int(rng.integers(0, 4))):                   # a few ambiguous runs; the wrapper adds the overlapping ones
            a = int(rng.integers(2, n - 20))
            sq[a:a + int(rng.integers(1, 18))] = int(rng.choice([3, 5, 6, 9, 10, 12, 14, 15]))
        seqs.append(sq)
    packs = {"old": G.disordered(blastdb.pack_nucleotide), "new": G.disordered(lambda c: _pack_nucleotide_new_format(c))}
    half = len(seqs) // 2
    want = [_apply_table_in_file_order(sq, packs["old" if k < half else "new"](sq)[1]) for k, sq in enumerate(seqs)]
    assert sum(1 for w, sq in zip(want, seqs) if not np.array_equal(w, sq)) > 100        # the overlaps do change bases
    a, b = str(tmp_path / "va"), str(tmp_path / "vb")
    monkeypatch.setattr(blastdb, "pack_nucleotide", packs["old"])
    blastdb.write_volume(a, seqs[:half], protein=False, ids=[f"s{i}" for i in range(half)])
    monkeypatch.setattr(blastdb, "pack_nucleotide", packs["new"])
    blastdb.write_volume(b, seqs[half:], protein=False, ids=[f"s{i}" for i in range(half, len(seqs))])
    monkeypatch.undo()
    base = str(tmp_path / "ovl")
    blastdb.write_alias(base, [a, b], protein=False)
    tiny = dict(SWA_LOAD_PART=1 << 14, SWA_LOAD_CHUNK=1 << 12, SWA_LOAD_THREADS=3)
    for how in ("old reader", "loader", "budget"):
        if how == "old reader":
            with _Env(SWA_PIPELINED=0):
                d = swipe_amd.Database.open(base, symtype=0)
        elif how == "loader":
            with _Env(**tiny):
                d = swipe_amd.Database.open(base, symtype=0)
        else:
            with _Env(SWA_STREAM_RESERVE=4096):
                d = swipe_amd.Database.open(base, symtype=0, hbm_budget=150_000)
        try:
            for k in range(len(seqs)):
                assert np.array_equal(d.sequence(k), want[k]), (how, k)
        finally:
            d.close()


@pytest.mark.late
def test_nucleotide_loader_edge_cases(tmp_path, monkeypatch):
    """what the .nsq format can hold and the unpack kernel must survive: empty and 1..3-base sequences (up to eight sequences
    share an output dword), ambiguity codes at the first and the last base, whole sequences of N, runs in both table forms
    (32-bit entries of at most 16 bases, 64-bit entries of up to 4 096), an entry several times larger than a staging chunk,
    a range that starts and ends inside volumes, and an OID mask on top.  Every base of every sequence, and both strands'
    scores, against the old reader and the source arrays"""
    rng = np.random.default_rng(17)
    acgt = np.array([1, 2, 4, 8], np.uint8)
    lens = [0, 1, 2, 3, 4, 5, 7, 8, 9, 0, 1, 15, 16, 17, 31, 32, 33, 2, 2, 2, 1, 1, 1, 1, 3, 0, 0, 6] + [int(x) for x in rng.integers(0, 40, 600)] + \
           [int(x) for x in rng.integers(40, 900, 900)] + [50_000, 3, 0, 12_345]
    seqs = [acgt[rng.integers(0, 4, n)] for n in lens]
    for k, sq in enumerate(seqs):
        n = len(sq)
        if n == 0:
            continue
        if k % 5 == 0:
            sq[0] = 15
        if k % 7 == 0:
            sq[-1] = int(rng.choice([5, 10, 14]))
        if k % 41 == 0:
            sq[:] = 15                                             # all N
        if n > 200 and k % 3 == 0:
            a = int(rng.integers(0, n - 150))
            sq[a:a + int(rng.integers(17, 140))] = int(rng.choice([3, 6, 9, 12, 15]))     # longer than one 32-bit entry holds
    seqs[-4][20_000:23_500] = 15                                  # 3 500 N: one 64-bit entry, 219 32-bit ones
    half = len(seqs) // 2
    a, b = str(tmp_path / "va"), str(tmp_path / "vb")
    blastdb.write_volume(a, seqs[:half], protein=False, ids=[f"s{i}" for i in range(half)])
    monkeypatch.setattr(blastdb, "pack_nucleotide", _pack_nucleotide_new_format)
    blastdb.write_volume(b, seqs[half:], protein=False, ids=[f"s{i}" for i in range(half, len(seqs))])
    monkeypatch.undo()
    base = str(tmp_path / "edge")
    blastdb.write_alias(base, [a, b], protein=False)
    inc = rng.random(half) < 0.6
    blastdb.write_mask_alias(str(tmp_path / "edgemask"), a, inc, memb_bit=1, length=int(sum(len(x) for x, k in zip(seqs[:half], inc) if k)), protein=False)
    q = acgt[rng.integers(0, 4, 1000)]
    qr = blastdb.revcomp_nt16(q)
    M = swipe_amd.matrix_nucleotide(1, -3)
    tiny = dict(SWA_LOAD_PART=1 << 14, SWA_LOAD_CHUNK=1 << 12, SWA_LOAD_THREADS=3)      # (a 50 000-base entry is 12.5 KiB: three chunks' worth)
    for name, first, last, count in ((base, 0, -1, len(seqs)), (base, 11, len(seqs) - 3, len(seqs) - 13), (str(tmp_path / "edgemask"), 0, -1, half)):
        with _Env(SWA_PIPELINED=0):
            old = swipe_amd.Database.open(name, symtype=0, first_seqno=first, last_seqno=last)
        with _Env(**tiny):
            new = swipe_amd.Database.open(name, symtype=0, first_seqno=first, last_seqno=last, wait=False)
        try:
            for d in (old, new):
                d.set_scoring(M, 5, 2)
            s1, s2, c = new.search2(q, qr)                         # follows the loader (16 x 63 rows)
            o1, o2, _ = old.search2(q, qr)
            assert np.array_equal(s1, o1) and np.array_equal(s2, o2), name
            new.wait()
            assert new.info() == dict(old.info(), hbm_bytes=new.info()["hbm_bytes"])
            for k in range(count):
                want = seqs[first + k]
                got = new.sequence(first + k)
                assert np.array_equal(got, want), (name, first + k, len(want))
            t1, t2 = old.search2_topk(q, qr, keep=30, minscore=12), new.search2_topk(q, qr, keep=30, minscore=12)
            assert t1[:3] == t2[:3]
        finally:
            old.close()
            new.close()


@pytest.mark.late
def test_two_queries_per_pass_on_a_loading_protein_shard(volumes):
    """swa_search_pair_topk / swa_search2 follow the loader too (chains of 2 / 4 / 8 lanes stream the parts' pair format); the
    16-lane two-query builds (long queries) wait for the shard"""
    res, off = volumes["res"], volumes["off"]
    qa, qb = np.ascontiguousarray(Q[:230]), np.ascontiguousarray(Q[60:270])          # 8 lanes x 29 rows: the parts' own format
    Mo = oracle.matrix_builtin("BLOSUM62")
    cpus = os.cpu_count() or 1
    refa, refb = oracle.search_all63(res, off, qa, Mo, 12, 1, threads=cpus), oracle.search_all63(res, off, qb, Mo, 12, 1, threads=cpus)
    with _Env(**SLOW):
        db = swipe_amd.Database.open(volumes["one"], wait=False)
    try:
        db.set_scoring(_matrix(), 11, 1)
        a, b, c = db.search_pair_topk(qa, qb, keep=(100, 60), minscore=(50, 45))
        assert c["loading_parts"] >= 8, c
        assert a[:2] == _expected_topk(refa, 100, 50) and b[:2] == _expected_topk(refb, 60, 45)
        db.wait()
        a2, b2, c2 = db.search_pair_topk(qa, qb, keep=(100, 60), minscore=(50, 45))
        assert c2["loading_parts"] == 0 and a2[:2] == a[:2] and b2[:2] == b[:2]
    finally:
        db.close()
    with _Env(**SLOW):
        db = swipe_amd.Database.open(volumes["one"], wait=False)
    try:
        db.set_scoring(_matrix(), 11, 1)
        s1, s2, c = db.search2(qa, np.ascontiguousarray(Q[100:330]))
        assert c["loading_parts"] >= 8, c
        assert np.array_equal(s1, refa) and np.array_equal(s2, oracle.search_all63(res, off, Q[100:330], Mo, 12, 1, threads=cpus))
        # a pair of 375-row queries: whatever has or has not arrived by now, same answers
        a3 = db.search_pair_topk(Q, Q[::-1].copy(), keep=(50, 50), minscore=(60, 60))
        assert a3[0][:2] == _expected_topk(volumes["ref"], 50, 60)
    finally:
        db.close()
    # With a threshold clear of the bound's slack the pair has a bound build on 8 lanes (8 x 47 rows), which streams the parts' pair
    # format: the search follows the loader.  (Round 6: where the table would pick a 16-lane build for a loading protein shard and a
    # shorter chain has a build for the query, the shorter chain is taken instead of waiting for the whole shard.)
    with _Env(**SLOW):
        db = swipe_amd.Database.open(volumes["one"], wait=False)
    try:
        db.set_scoring(_matrix(), 11, 1)
        a4, b4, c4 = db.search_pair_topk(Q, Q[::-1].copy(), keep=(50, 50), minscore=(70, 70))      # (70: clear of the bound's slack)
        assert c4["loading_parts"] >= 8 and c4["narrow_rows"] * 8 >= 375 and c4["narrow_shifted"] == 10, c4
        assert a4[:2] == _expected_topk(volumes["ref"], 50, 70)
        db.wait()
        a5, b5, c5 = db.search_pair_topk(Q, Q[::-1].copy(), keep=(50, 50), minscore=(70, 70))
        assert c5["loading_parts"] == 0 and a5[:2] == a4[:2] and b5[:2] == b4[:2]
    finally:
        db.close()


@pytest.mark.late
@pytest.mark.parametrize("early", [True, False])
def test_masked_alias_streams_in(volumes, tmp_path, early):
    """an OID-mask alias through the pipelined open: a search that follows the loader marks the excluded sequences, the
    adopted tables hold the members only; hit lists, counts and the statistics' totals are the old reader's"""
    n = 60_000
    inc = (np.arange(n) * 7919 % 10) < 4
    length = int(sum(int(volumes["off"][i + 1] - volumes["off"][i]) for i in np.nonzero(inc)[0]))
    alias = str(tmp_path / "msk")
    for ext in ("pin", "psq", "phr"):
        os.symlink(volumes["one"] + "." + ext, str(tmp_path / ("one." + ext)))
    blastdb.write_mask_alias(alias, str(tmp_path / "one"), inc, memb_bit=1, length=length)
    ref = np.where(inc, volumes["ref"], -1)
    want_hits, want_tot = _expected_topk(ref, 100, 60)
    with _Env(SWA_PIPELINED=0):
        old = swipe_amd.Database.open(alias)
    with _Env(**SLOW):
        new = swipe_amd.Database.open(alias, wait=False)
    try:
        for d in (old, new):
            d.set_scoring(_matrix(), 11, 1)
        if not early:
            new.wait()
        hits, tot, obv, c = new.search_topk(Q, keep=100, minscore=60)
        assert (c["loading_parts"] > 0) == early
        assert (hits, tot) == (want_hits, want_tot)
        assert old.search_topk(Q, keep=100, minscore=60)[:2] == (want_hits, want_tot)
        new.wait()
        io, inw = old.info(), new.info()
        assert all(io[k] == inw[k] for k in ("seqcount", "symcount", "total_seqcount", "total_symcount")) and inw["total_seqcount"] == int(inc.sum())
        sc, c2 = new.search(Q)
        assert c2["loading_parts"] == 0 and np.array_equal(np.where(inc, sc, -1), ref) and c2["cells"] == length * len(Q)
        assert new.search_topk(Q, keep=100, minscore=60)[:2] == (want_hits, want_tot)
    finally:
        old.close()
        new.close()


@pytest.mark.late
def test_masked_alias_opened_as_six_translations_through_the_loader(tmp_path):
    """round 6 (missing 4): a masked nucleotide alias held as its six translations (-p 3 / -p 4 on an OID-masked
    database) opens through the pipelined loader too - the mask is taken off the loader and becomes the translated shard's
    inclusion set.  Same frame-tagged hits, counts, scores of every frame and statistics totals as through the old reader"""
    rng = np.random.default_rng(41)
    res, off = swipe_amd.synth_db(3, 2500, protein=False)
    seqs = [res[off[i]:off[i + 1]] for i in range(2500)]
    vol = str(tmp_path / "ntv")
    blastdb.write_volume(vol, seqs, protein=False, ids=[f"s{i}" for i in range(len(seqs))])
    inc = rng.random(len(seqs)) < 0.55
    alias = str(tmp_path / "ntmask")
    blastdb.write_mask_alias(alias, vol, inc, memb_bit=1, length=int(sum(len(x) for x, k in zip(seqs, inc) if k)), protein=False)
    q = Q[:180]
    got = []
    for env in (dict(SWA_PIPELINED=0), dict(SWA_LOAD_PART=1 << 15, SWA_LOAD_CHUNK=1 << 13, SWA_LOAD_THREADS=3)):
        with _Env(**env):
            db = swipe_amd.Database.open_translated(alias, db_gencode=1)
        try:
            db.set_scoring(_matrix(), 11, 1)
            scores, _ = db.search(q)
            hits = db.search_frames_topk([q], keep=60, minscore=30)[:3]
            info = db.info()
            got.append((scores, hits, {k: info[k] for k in ("seqcount", "symcount", "total_seqcount", "total_symcount", "frames")}))
        finally:
            db.close()
    assert np.array_equal(got[0][0], got[1][0]) and got[0][1] == got[1][1] and got[0][2] == got[1][2]
    assert got[0][2]["frames"] == 6 and got[0][2]["total_seqcount"] == int(inc.sum())
    excluded = np.repeat(~inc, 6)
    assert (got[1][0][excluded] == -1).all() and (got[1][0][~excluded] >= 0).all() and got[1][1][1] > 0


_REDZONE_SCRIPT = r"""
import os, sys
sys.path.insert(0, %r)
import numpy as np
import swipe_amd
from swipe_amd import blastdb, synth
rng = np.random.default_rng(7)
q0 = blastdb.encode_protein(synth.QUERY_P07327)
res, off = swipe_amd.synth_db(9, 30_000, query=q0)
# three long subjects so that window views are built
extra = [rng.integers(1, 21, n).astype(np.uint8) for n in (30_000, 12_000, 9_000)]
res = np.concatenate([res] + extra); off = np.concatenate([off, off[-1] + np.cumsum([len(e) for e in extra])])
M = swipe_amd.matrix_builtin("BLOSUM62")
db = swipe_amd.Database.from_arrays(res, off)
db.set_scoring(M, 11, 1)
searches = 0
for qlen in list(range(1, 70)) + [95, 96, 97, 124, 128, 191, 192, 193, 248, 255, 256, 375, 384, 385, 496, 500, 640, 767, 768, 769, 928, 929, 1000, 1100, 2300]:
    q = q0[:qlen] if qlen <= len(q0) else rng.integers(1, 21, qlen).astype(np.uint8)
    db.search(q, want_scores=False); searches += 1
    for bound in ("0", "1"):
        db.set_option("bound", bound)
        db.search_topk(q, keep=50, minscore=60); searches += 1
    db.set_option("bound", None)
    if qlen %% 7 == 0 or qlen > 300:
        db.search_pair_topk(q, q[::-1].copy(), keep=20, minscore=(60, 60)); searches += 1
hits = db.search_topk(q0, keep=60, minscore=50)[0]
db.align(q0, [h[0] for h in hits]); db.search_endpoints(q0, [h[0] for h in hits])
# a matrix that leaves the packed range: 32-bit and 64-bit re-queues
big = np.array(M, dtype=np.int64).copy(); big[big > 0] *= 40
db.set_scoring(big, 400, 40); db.search(q0, want_scores=False); db.search_topk(q0, keep=10, minscore=1000); searches += 2
db.close()
# budgeted shard (two slots), nucleotide shard (both strands, 4-bit stream), a shard that streams in from disk
sdb = swipe_amd.Database.from_arrays(res, off, hbm_budget=24 << 20); sdb.set_scoring(M, 11, 1)    # two slots of 4 MiB beside the 2 x 8 MiB reserve
sdb.search_topk(q0, keep=50, minscore=60); sdb.search(q0[:40], want_scores=False); sdb.close(); searches += 2
nres, noff = swipe_amd.synth_db(3, 20_000, protein=False)
ndb = swipe_amd.Database.from_arrays(nres, noff, symtype=0); ndb.set_scoring(swipe_amd.matrix_nucleotide(1, -3), 5, 2)
for qlen in (18, 30, 48, 100, 300, 480, 1000, 1500):
    qn = synth._random_residues(5, 1, qlen, synth.residue_table_nucleotide())
    ndb.search2_topk(qn, blastdb.revcomp_nt16(qn), keep=20, minscore=20); searches += 1
ndb.close()
base = os.path.join(%r, "rz")
swipe_amd.write_blastdb(base, res, off, first_id=0)
os.environ.update(SWA_LOAD_PART=str(1 << 20), SWA_LOAD_CHUNK=str(1 << 20), SWA_LOAD_DELAY_MS="5")
ldb = swipe_amd.Database.open(base, wait=False); ldb.set_scoring(M, 11, 1)
c = ldb.search_topk(q0, keep=50, minscore=60)[3]; ldb.search(q0[:33], want_scores=False); ldb.wait(); ldb.search_topk(q0, keep=50, minscore=60); searches += 3
n, bad, report = swipe_amd.redzones_check()
print("REDZONES searches=%%d allocations=%%d touched=%%d loading_parts=%%d %%s" %% (searches, n, bad, c["loading_parts"], report))
ldb.close()
"""


@pytest.mark.late
def test_kernels_stay_inside_their_buffers(tmp_path):
    """SWA_REDZONES=1: every device allocation of the library has a 4 KiB guard in front and one right behind its last
    requested byte; after ~300 searches that reach the one-lane, chain, long-lane, pass, bound, two-query, window, re-queue,
    budgeted and streaming-in paths not one guard byte may have changed (the reference's only mention of a memory checker
    is Valgrind, CHANGES:54).  A child process: the switch is read when the library is first used."""
    import subprocess
    import sys
    from conftest import ROOT, under_interpreter
    env = dict(os.environ, SWA_REDZONES="1", SWA_WATCHDOG_S="60")
    r = subprocess.run([sys.executable, "-c", _REDZONE_SCRIPT % (ROOT, str(tmp_path))], capture_output=True, text=True, timeout=2400 if under_interpreter() else 600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("REDZONES")][-1]
    f = dict(kv.split("=") for kv in line.split()[1:5])
    assert int(f["searches"]) > 250 and int(f["allocations"]) >= 12 and int(f["touched"]) == 0, line     # (the loader's leftovers are released by wait())
    assert int(line.split("loading_parts=")[1].split()[0]) > 0, line
