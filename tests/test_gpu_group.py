"""GPU: the C++ multi-device layer (swa_group, swipe_amd_cli -a N -g LIST) - the reference's run_threads / worker
(swipe.cc:1599-1699) and the MPI master's re-entry of reported hits (swipe.cc:1951-1974) with a device as the unit.

Shards go to the devices conftest.shard_devices names (SWA_TEST_DEVICES, default every device the library sees, cycled): on
the one-GPU test box every shard lives on device 0 (`-g 0,0,0`: three handles, three host threads, three sets of
streams) - the orchestration, the routing of the alignment phase and the merge are exactly what N devices run - and a
box with several devices (or a partitioned one) runs the same tests on distinct ordinals.
Everything must equal the single-shard result and the reference's golden CLI output byte for byte."""
import os
import re
import subprocess

import numpy as np
import pytest

import cases
import oracle
import swipe_amd
from conftest import ROOT, case_matrix, load_golden, shard_devices
from swipe_amd import blastdb

pytestmark = pytest.mark.gpu
EXE = os.path.join(ROOT, "swipe_amd", "swipe_amd_cli")
THREADS = os.cpu_count() or 1


def shard_args(k):
    return ["-a", str(k), "-g", ",".join(str(d) for d in shard_devices(k))]


def write_case(tmp_path, name):
    case, g = cases.get(name), load_golden(name)
    base = str(tmp_path / name)
    blastdb.write_db(base, case.seqs, protein=case.protein, volumes=case.volumes)
    alpha = blastdb.NCBI4NA if (case.sym in (0, 2, 4)) else blastdb.NCBISTDAA
    qf = str(tmp_path / "q.fa")
    open(qf, "w").write(">query test\n" + "".join(alpha[c] for c in case.query) + "\n")
    return case, g, base, qf


@pytest.mark.parametrize("shards", [2, 3])
@pytest.mark.parametrize("name", ["p1k", "multivol", "nt", "asym", "edges", "limit16"])
def test_cli_sharded_over_devices_equals_reference_cli(tmp_path, name, shards):
    """-a N: hit list, alignments (-m 7 / 8 / 0) of N shards = the reference's output for the whole database; the hits of
    a list come from different shards and every alignment is made by the shard that owns the sequence."""
    case, g, base, qf = write_case(tmp_path, name)
    args = [EXE, "-d", base, "-i", qf, "-p", "1" if case.protein else "0", "-G", str(case.gapopen), "-E", str(case.gapextend),
            "-v", str(case.keep), "-e", "10"] + shard_args(shards)
    if case.protein:
        mat = case.matrix
        if mat == "@text":
            mat = str(tmp_path / "matrix.txt")
            open(mat, "w").write(case.matrix_text)
        args += ["-M", mat]
    else:
        args += ["-r", str(case.match), "-q", str(case.mismatch)]
    run = lambda extra, env=None: subprocess.run(args + extra, capture_output=True, text=True, check=True, env=env).stdout
    assert run(["-m", "7", "-b", str(g["nalign"])]) == g["xml_align"]
    assert run(["-m", "8", "-b", str(case.keep)]) == g["tsv"]
    plain = run(["-m", "0", "-b", str(g["nalign"])])
    assert plain[plain.index("Sequences producing"):] == g["plain_align"]
    assert f"Threads:           {shards}" in plain
    # and with the bound build of the first pass forced on every shard
    assert run(["-m", "8", "-b", str(case.keep)], dict(os.environ, SWA_BOUND="1")) == g["tsv"]
    # hit list without alignments
    xml = run(["-m", "7", "-b", "0"])
    strip = lambda t: re.sub(r"\s*<len>\d+</len>", "", t)
    assert strip(xml) == strip(g["xml"])


@pytest.mark.parametrize("name", [f.__name__[5:] for f in cases.TRANSLATED])
def test_cli_translated_searches_sharded(tmp_path, name):
    case, g, base, qf = write_case(tmp_path, name)
    args = [EXE, "-d", base, "-i", qf, "-p", str(case.sym), "-G", str(case.gapopen), "-E", str(case.gapextend), "-v", str(case.keep),
            "-e", "10", "-M", case.matrix, "-Q", str(case.query_gencode), "-D", str(case.db_gencode)] + shard_args(2)
    run = lambda extra: subprocess.run(args + extra, capture_output=True, text=True, check=True).stdout
    assert run(["-m", "7", "-b", str(g["nalign"])]) == g["xml_align"]
    assert run(["-m", "8", "-b", str(case.keep)]) == g["tsv"]
    plain = run(["-m", "0", "-b", str(g["nalign"])])
    assert plain[plain.index("Sequences producing"):] == g["plain_align"]


@pytest.mark.parametrize("variant", ["plain_gis_taxid", "masked", "masked_taxlist", "taxlist_gis_taxid"])
def test_cli_masks_and_taxid_lists_sharded(tmp_path, variant):
    """an OID-mask alias and a -x taxid list cut across shard boundaries: each shard applies its slice"""
    from test_host_cpu import build_headers_db, HEADER_VARIANTS
    case, vol, masked, tx = build_headers_db(tmp_path)
    ref = load_golden("headers")["variants"][variant]
    dbn, flags, taxlist = HEADER_VARIANTS[variant]
    qf = str(tmp_path / "q.fa")
    open(qf, "w").write(">query test\n" + "".join(blastdb.NCBISTDAA[c] for c in case.query) + "\n")
    args = [EXE, "-d", vol if dbn == "vol" else masked, "-i", qf, "-v", str(case.keep), "-e", "1e6"] + shard_args(3)
    args += (["-I"] if flags & 1 else []) + (["-H"] if flags & 2 else []) + (["-x", tx] if taxlist else [])
    run = lambda extra: subprocess.run(args + extra, capture_output=True, text=True, check=True).stdout
    assert run(["-m", "7", "-b", "5"]) == ref["m7"]
    assert run(["-m", "8", "-b", str(case.keep)]) == ref["m8"]
    plain = run(["-m", "0", "-b", "5"])
    assert plain[plain.index("Sequences producing"):] == ref["m0"]


def test_cli_multi_query_file_sharded(tmp_path):
    """queries of a file two per pass (swa_group_search_pair_topk) on every shard"""
    g = load_golden("multiquery")
    case = cases.get("edges")
    base = str(tmp_path / "db")
    blastdb.write_db(base, case.seqs, protein=True)
    qf = str(tmp_path / "q.fa")
    open(qf, "w").write(g["query_text"])
    r = subprocess.run([EXE, "-d", base, "-i", qf, "-m", "8", "-b", "10", "-v", "12", "-e", "1000"] + shard_args(2), capture_output=True, text=True)
    assert r.returncode == g["rc8"], r.stderr
    assert r.stdout == g["m8"]


def test_cli_thread_and_device_arguments(tmp_path):
    case, g, base, qf = write_case(tmp_path, "p1k")
    common = [EXE, "-d", base, "-i", qf, "-m", "8", "-b", "5"]
    r = subprocess.run(common + ["-a", "0"], capture_output=True, text=True)
    assert r.returncode == 1 and "Illegal number of threads specified" in r.stderr          # swipe.cc:1131
    r = subprocess.run(common + ["-g", "4099"], capture_output=True, text=True)
    assert r.returncode == 1 and "no such HIP device" in r.stderr
    r = subprocess.run(common + ["-g", "0,x"], capture_output=True, text=True)
    assert r.returncode == 1 and "Illegal device list" in r.stderr
    # more threads than devices listed: as many shards as devices; more shards than sequences: the empty ones are dropped
    one = subprocess.run(common, capture_output=True, text=True, check=True).stdout
    assert subprocess.run(common + ["-a", "16"], capture_output=True, text=True, check=True).stdout == one
    tiny = str(tmp_path / "tiny")
    blastdb.write_db(tiny, case.seqs[:3], protein=True)
    a = subprocess.run([EXE, "-d", tiny, "-i", qf, "-m", "8", "-b", "5", "-e", "1e9"], capture_output=True, text=True, check=True).stdout
    b = subprocess.run([EXE, "-d", tiny, "-i", qf, "-m", "8", "-b", "5", "-e", "1e9"] + shard_args(8), capture_output=True, text=True, check=True).stdout
    assert a == b and a


@pytest.mark.parametrize("shards", [2, 5])
def test_group_api_equals_one_shard(shards):
    """swa_group_* through ctypes on 30 000 synthetic sequences with planted homologs: every score, the top-K with ties
    cut by keep, counts, the two-queries-per-pass form and the alignments equal the single-handle calls."""
    q = blastdb.encode_protein(swipe_amd.synth.QUERY_P07327)
    res, off = swipe_amd.synth_db(7, 30000, query=q)
    M = swipe_amd.matrix_builtin("BLOSUM62")
    one = swipe_amd.Database.from_arrays(res, off, first_seqno=11)
    grp = swipe_amd.Group.from_arrays(res, off, devices=tuple(shard_devices(shards)), first_seqno=11)
    one.set_scoring(M, 11, 1)
    grp.set_scoring(M, 11, 1)
    gi, oi = grp.info(), one.info()
    assert gi["nshards"] == shards and all(gi[k] == oi[k] for k in ("seqcount", "symcount", "longest", "first_seqno", "total_seqcount", "total_symcount"))
    sizes = [grp.shard_info(k)["symcount"] for k in range(shards)]
    assert max(sizes) - min(sizes) <= 2 * oi["longest"]                      # residue-balanced
    s1, c1 = one.search(q)
    s2, c2 = grp.search(q)
    assert np.array_equal(s1, s2) and c1["cells"] == c2["cells"] and c2["narrow"] == 30000
    for keep, lo, hi in ((250, 40, 1 << 62), (1, 1, 1 << 62), (37, 45, 300), (1000, 30, 1 << 62)):
        assert one.search_topk(q, keep, lo, hi)[:3] == grp.search_topk(q, keep, lo, hi)[:3], (keep, lo, hi)
    grp.set_option("bound", 1)
    assert one.search_topk(q, 250, 80)[:3] == grp.search_topk(q, 250, 80)[:3]
    grp.set_option("bound", None)
    q2 = np.ascontiguousarray(q[20:340])
    a, b, _ = one.search_pair_topk(q, q2, keep=(100, 50), minscore=(40, 35))
    x, y, _ = grp.search_pair_topk(q, q2, keep=(100, 50), minscore=(40, 35))
    assert a == x and b == y
    hits = grp.search_topk(q, 60, 40)[0]
    assert len({int(np.searchsorted([grp.shard_info(k)["first_seqno"] for k in range(shards)], h[0], side="right")) for h in hits}) > 1, \
        "hits of one list should come from several shards"
    ids = [h[0] for h in hits]
    assert one.align(q, ids) == grp.align(q, ids)
    for s in ids[:5]:
        assert np.array_equal(one.sequence(s), grp.sequence(s))
    with pytest.raises(swipe_amd.SwaError):
        grp.sequence(10)
    # an inclusion set that silences whole shards
    inc = np.ones(30000, np.uint8)
    inc[: 30000 * 3 // 5] = 0
    one.set_inclusion(inc)
    grp.set_inclusion(inc)
    assert one.search_topk(q, 250, 40)[:3] == grp.search_topk(q, 250, 40)[:3]
    assert np.array_equal(one.search(q)[0], grp.search(q)[0])
    one.close()
    grp.close()


@pytest.mark.late
@pytest.mark.parametrize("variant", ["masked", "masked_gis_taxid", "taxlist", "plain_taxid", "masked_taxlist"])
def test_cli_masks_and_taxid_lists_with_an_hbm_budget(tmp_path, variant):
    """OID-mask aliases and -x taxid lists on shards that walk their parts through two device slots
    (db_check_inclusion works on whatever range the reference has mapped, database.cc:670-772, 1403-1481) - the reference's
    golden output, byte for byte, at about a half and a quarter of the footprint; SWA_STREAM_RESERVE shrinks the fixed
    allowance of a slot so that this small database really is cut into parts"""
    from test_host_cpu import build_headers_db, HEADER_VARIANTS
    case, vol, masked, tx = build_headers_db(tmp_path)
    ref = load_golden("headers")["variants"][variant]
    dbn, flags, taxlist = HEADER_VARIANTS[variant]
    qf = str(tmp_path / "q.fa")
    open(qf, "w").write(">query test\n" + "".join(blastdb.NCBISTDAA[c] for c in case.query) + "\n")
    nsym, nseq = sum(len(x) for x in case.seqs), len(case.seqs)
    footprint = int(2.04 * nsym + 77 * nseq) + 4096
    env = dict(os.environ, SWA_STREAM_RESERVE="2048")
    for shards in (1, 2):
        for frac in (0.5, 0.25):
            budget = int(frac * footprint / shards) + 2 * 2048 + 2 * int(2.04 * max(len(x) for x in case.seqs) + 77)
            args = [EXE, "-d", vol if dbn == "vol" else masked, "-i", qf, "-v", str(case.keep), "-e", "1e6", "--hbm-budget", str(budget)] + shard_args(shards)
            args += (["-I"] if flags & 1 else []) + (["-H"] if flags & 2 else []) + (["-x", tx] if taxlist else [])
            run = lambda extra: subprocess.run(args + extra, capture_output=True, text=True, check=True, env=env).stdout
            assert run(["-m", "7", "-b", "5"]) == ref["m7"], (shards, frac)
            assert run(["-m", "8", "-b", str(case.keep)]) == ref["m8"], (shards, frac)
            plain = run(["-m", "0", "-b", "5"])
            assert plain[plain.index("Sequences producing"):] == ref["m0"], (shards, frac)


@pytest.mark.late
@pytest.mark.parametrize("nt", [False, True])
def test_inclusion_set_on_a_budgeted_shard(nt):
    """swa_db_set_inclusion on a handle opened with an HBM budget: every part re-plans its tables for the admitted sequences;
    all scores, top-K with counts, pairs / both strands, end points and alignments equal the resident shard's under the same
    set - including sets that empty whole parts - and lifting the set restores the full answers"""
    from swipe_amd import synth
    rng = np.random.default_rng(11)
    if nt:
        res, off = swipe_amd.synth_db(4, 12_000, protein=False)
        q = synth._random_residues(17, 1, 300, synth.residue_table_nucleotide())
        M, go, ge, sym = swipe_amd.matrix_nucleotide(1, -3), 5, 2, 0
    else:
        q = blastdb.encode_protein(synth.QUERY_P07327)
        res, off = swipe_amd.synth_db(8, 12_000, query=q)
        M, go, ge, sym = swipe_amd.matrix_builtin("BLOSUM62"), 11, 1, 1
    n = len(off) - 1
    footprint = int((1.02 if nt else 2.04) * int(off[-1]) + 90 * n)
    os.environ["SWA_STREAM_RESERVE"] = "65536"
    try:
        sdb = swipe_amd.Database.from_arrays(res, off, symtype=sym, hbm_budget=footprint // 3 + 2 * 65536)
    finally:
        os.environ.pop("SWA_STREAM_RESERVE", None)
    one = swipe_amd.Database.from_arrays(res, off, symtype=sym)
    assert sdb.info()["hbm_bytes"] < one.info()["hbm_bytes"]          # it really is over budget: two slots of a part each
    for d in (one, sdb):
        d.set_scoring(M, go, ge)
    sets = [rng.random(n) < 0.5, np.arange(n) >= n * 2 // 3, np.arange(n) % 997 == 5, None]
    for inc in sets:
        u8 = None if inc is None else inc.astype(np.uint8)
        one.set_inclusion(u8)
        sdb.set_inclusion(u8)
        if nt:
            qr = blastdb.revcomp_nt16(q)
            assert one.search2_topk(q, qr, keep=60, minscore=25)[:3] == sdb.search2_topk(q, qr, keep=60, minscore=25)[:3]
            a, b = one.search2(q, qr)[:2], sdb.search2(q, qr)[:2]
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        else:
            assert np.array_equal(one.search(q)[0], sdb.search(q)[0])
            for keep, lo in ((100, 45), (7, 1)):
                assert one.search_topk(q, keep, lo)[:3] == sdb.search_topk(q, keep, lo)[:3], (keep, lo)
            q2 = np.ascontiguousarray(q[30:330])
            x, y = one.search_pair_topk(q, q2, keep=(50, 20), minscore=(45, 40)), sdb.search_pair_topk(q, q2, keep=(50, 20), minscore=(45, 40))
            assert x[0] == y[0] and x[1] == y[1]
            hits = [h[0] for h in sdb.search_topk(q, 30, 45)[0]]
            assert one.align(q, hits) == sdb.align(q, hits)
    one.close()
    sdb.close()


@pytest.mark.late
def test_group_wait_and_load_errors(tmp_path):
    """swa_group_open streams its shards in behind the call; swa_group_wait / swa_group_load_progress are how a caller
    waits for them and sees a load error (here: a residue code >= 32 in the second shard) before the first search"""
    q = blastdb.encode_protein(swipe_amd.synth.QUERY_P07327)
    res, off = swipe_amd.synth_db(3, 20000, query=q)
    base = str(tmp_path / "g")
    swipe_amd.write_blastdb(base, res, off, first_id=0)
    grp = swipe_amd.Group.open(base, devices=tuple(shard_devices(3)))
    p = grp.load_progress()
    assert set(p) == {"bytes_loaded", "bytes_total", "parts_ready", "parts_total"}
    grp.wait()
    assert grp.load_progress() == {"bytes_loaded": 0, "bytes_total": 0, "parts_ready": 0, "parts_total": 0}
    grp.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
    one = swipe_amd.Database.from_arrays(res, off)
    one.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
    assert grp.search_topk(q, 100, 50)[:3] == one.search_topk(q, 100, 50)[:3]
    one.close()
    grp.close()
    with open(base + ".psq", "r+b") as f:
        f.seek(int(off[15000]) + 15000 + 3)
        f.write(bytes([99]))
    bad = swipe_amd.Group.open(base, devices=tuple(shard_devices(3)))
    try:
        with pytest.raises(swipe_amd.SwaError, match="out of range"):
            bad.wait()
    finally:
        bad.close()


def test_group_of_a_translated_nucleotide_database():
    case = cases.get("tblastn")
    res, off = oracle.pack(case.seqs)
    M = case_matrix(case, swipe_amd)
    one = swipe_amd.Database.from_arrays(res, off, translate_gencode=case.db_gencode)
    grp = swipe_amd.Group.from_arrays(res, off, devices=tuple(shard_devices(3)), translate_gencode=case.db_gencode)
    one.set_scoring(M, case.gapopen, case.gapextend)
    grp.set_scoring(M, case.gapopen, case.gapextend)
    q = np.asarray(case.query, dtype=np.uint8)
    assert np.array_equal(one.search(q)[0], grp.search(q)[0])
    h1 = one.search_frames_topk([q], keep=case.keep, minscore=20)
    h2 = grp.search_frames_topk([q], keep=case.keep, minscore=20)
    assert h1[:3] == h2[:3] and h1[0]
    ids, ds, df = [h[0] for h in h1[0]], [h[4] for h in h1[0]], [h[5] for h in h1[0]]
    assert one.align(q, ids, ds, df) == grp.align(q, ids, ds, df)
    one.close()
    grp.close()


def _option_runs():
    g = load_golden("options")
    return [(name, i) for name in g for i in range(len(g[name]["runs"]))]


@pytest.mark.parametrize("name,i", _option_runs())
def test_cli_options_that_move_thresholds_and_strands_equal_reference_cli(tmp_path, name, i):
    """tests/golden/options.json: the reference CLI under -c / -u (score window), -e / -k (E-value window), -z (effective
    database size), -v / -b (list lengths), -S (query strands, also of translated queries), nucleotide rewards, other
    matrices and gap systems with and without Karlin-Altschul parameters - hits_init's thresholds (hits.cc:283-511) and
    search_chunk's strand loops (swipe.cc:277-337) end to end.  swipe_amd_cli on one shard and on two must print the same
    bytes."""
    g = load_golden("options")[name]
    case = cases.get(name)
    assert g["checksum"] == case.checksum()
    rec = g["runs"][i]
    base = str(tmp_path / name)
    blastdb.write_db(base, case.seqs, protein=case.protein, volumes=case.volumes)
    alpha = blastdb.NCBI4NA if case.query_is_nt else blastdb.NCBISTDAA
    qf = str(tmp_path / "q.fa")
    open(qf, "w").write(">query test\n" + "".join(alpha[c] for c in case.query) + "\n")
    args = [EXE, "-d", base, "-i", qf, "-p", str(case.sym)]
    if case.sym >= 2:
        args += ["-Q", str(case.query_gencode), "-D", str(case.db_gencode)]
    args += rec["options"]
    strip = lambda t: re.sub(r"\s*<len>\d+</len>", "", t)
    for shards in ([], shard_args(2)):
        r = subprocess.run(args + shards + ["-m", "8"], capture_output=True, text=True)
        assert r.returncode == rec["m8_rc"] and r.stdout == rec["m8"], (rec["options"], shards, r.stderr)
    r = subprocess.run(args + ["-m", "7", "-b", "0"], capture_output=True, text=True)
    assert r.returncode == rec["m7_rc"] and strip(r.stdout) == strip(rec["m7"]), (rec["options"], r.stderr)


@pytest.mark.parametrize("shards", [1, 3])
def test_cli_nucleotide_database_in_three_volumes_behind_an_alias(tmp_path, shards):
    """.nal alias over three .nin/.nsq volumes (ambiguity tables per volume): the reference prints for it what it prints for
    the single volume (checked on the CPU in test_oracle_vs_reference), and so must swipe_amd_cli - also when the shard
    boundaries of -a 3 fall inside volumes"""
    case, g = cases.get("nt"), load_golden("nt")
    base = str(tmp_path / "ntv")
    blastdb.write_db(base, case.seqs, protein=False, volumes=3)
    qf = str(tmp_path / "q.fa")
    open(qf, "w").write(">query test\n" + "".join(blastdb.NCBI4NA[c] for c in case.query) + "\n")
    args = [EXE, "-d", base, "-i", qf, "-p", "0", "-G", str(case.gapopen), "-E", str(case.gapextend), "-v", str(case.keep), "-e", "10",
            "-r", str(case.match), "-q", str(case.mismatch)] + shard_args(shards)
    run = lambda extra: subprocess.run(args + extra, capture_output=True, text=True, check=True).stdout
    assert run(["-m", "8", "-b", str(case.keep)]) == g["tsv"]
    assert run(["-m", "7", "-b", str(g["nalign"])]) == g["xml_align"]


@pytest.mark.late
@pytest.mark.parametrize("nt", [False, True])
def test_cli_and_group_over_shards_with_an_hbm_budget(tmp_path, nt):
    """swipe_amd_cli -a 3 --hbm-budget N and swa_group_open_streamed: every shard walks its parts through two device slots
    (the reference maps any range of a database a chunk at a time with any thread count, database.cc:1082-1131); hit lists,
    alignments and every printed byte equal the resident group's, at about a half and a quarter of a shard's footprint."""
    from swipe_amd import synth
    if nt:
        res, off = swipe_amd.synth_db(3, 150_000, protein=False)
        q = synth._random_residues(99, 1, 400, synth.residue_table_nucleotide())
        alpha, sym, extra = blastdb.NCBI4NA, 0, ["-p", "0", "-r", "1", "-q", "-3", "-G", "5", "-E", "2"]
    else:
        q = cases.Q375
        res, off = swipe_amd.synth_db(1, 150_000, query=q)
        alpha, sym, extra = blastdb.NCBISTDAA, 1, []
    base = str(tmp_path / "db")
    swipe_amd.write_blastdb(base, res, off, symtype=sym, first_id=0)
    qf = str(tmp_path / "q.fa")
    open(qf, "w").write(">query test\n" + "".join(alpha[c] for c in q) + "\n")
    devs = shard_devices(3)
    # what one of the three shards needs resident (residues + formatted stream + tables + the fixed allowance of 8 MB)
    shard_bytes = int((1.02 if nt else 2.04) * int(off[-1]) / 3 + 100 * 50_000 + (8 << 20))
    outs = {}
    for view in ("8", "0"):
        common = [EXE, "-d", base, "-i", qf, "-m", view, "-b", "25", "-v", "30"] + extra + shard_args(3)
        body = lambda text: text[text.index("Sequences producing"):] if view == "0" else text      # (-m 0 starts with the time of day)
        outs[view] = body(subprocess.run(common, capture_output=True, text=True, check=True).stdout)
        assert outs[view].count("\n") > (5 if nt else 20)              # (a random 400-nt query has a handful of hits with E <= 10)
        for frac in (0.55, 0.3):
            budget = max(int(frac * shard_bytes), 20 << 20)
            r = subprocess.run(common + ["--hbm-budget", str(budget)], capture_output=True, text=True)
            assert r.returncode == 0, r.stderr[-800:]
            assert body(r.stdout) == outs[view], (view, frac)
    # the same through the library: the group's shards really are over budget, and say so
    grp = swipe_amd.Group.open(base, symtype=sym, devices=tuple(devs), hbm_budget=max(int(0.3 * shard_bytes), 20 << 20))
    assert max(grp.shard_info(k)["hbm_bytes"] for k in range(3)) < shard_bytes        # two slots of a part each, not the shard
    grp.close()
