"""GPU: the drop-in claim, executed.  oracle/_ref/swipe_bound_scores and oracle/_ref/swipe_bound_topk are the REFERENCE
program - its own main(), option parsing, database reader, hits_init / hits_enter, alignment phase and output code,
compiled from /root/reference by oracle/Makefile - with the body of search_chunk() (swipe.cc:1365-1596) replaced by the
binding of INTEGRATION.md section 2 (cut out of the document at build time) and linked against libswipe_amd.so.  Their
output must equal the golden output of the unmodified reference byte for byte: with one worker thread and with eight
(chunks arriving concurrently; swipe_bound_group = binding B over swa_group: -a N is then N shards, here all on device 0), for protein / nucleotide / multi-volume / custom-matrix / 64-bit-score / translated
databases, OID masks and taxid lists, several queries per file."""
import os
import re
import subprocess

import pytest

import cases
from conftest import ROOT, load_golden
from swipe_amd import blastdb

pytestmark = pytest.mark.gpu
BOUND = {v: os.path.join(ROOT, "oracle", "_ref", "swipe_bound_" + v) for v in ("scores", "topk", "group")}


def need(variant):
    if not os.path.exists(BOUND[variant]):
        pytest.skip("oracle/_ref/swipe_bound_* is built in the container that holds /root/reference (oracle/Makefile)")
    return BOUND[variant]


def case_args(tmp_path, name):
    case, g = cases.get(name), load_golden(name)
    base = str(tmp_path / name)
    blastdb.write_db(base, case.seqs, protein=case.protein, volumes=case.volumes)
    alpha = blastdb.NCBI4NA if case.query_is_nt else blastdb.NCBISTDAA
    qf = str(tmp_path / "q.fa")
    open(qf, "w").write(">query test\n" + "".join(alpha[c] for c in case.query) + "\n")
    args = ["-d", base, "-i", qf, "-p", str(case.sym), "-G", str(case.gapopen), "-E", str(case.gapextend), "-v", str(case.keep), "-e", "10"]
    if case.sym != 0:
        mat = case.matrix
        if mat == "@text":
            mat = str(tmp_path / "matrix.txt")
            open(mat, "w").write(case.matrix_text)
        args += ["-M", mat]
    else:
        args += ["-r", str(case.match), "-q", str(case.mismatch)]
    if case.sym >= 2:
        args += ["-Q", str(case.query_gencode), "-D", str(case.db_gencode)]
    return case, g, args


@pytest.mark.parametrize("threads", [1, 8])
@pytest.mark.parametrize("variant", ["scores", "topk", "group"])
@pytest.mark.parametrize("name", ["p1k", "nt", "multivol", "asym", "edges", "limit16", "blastx", "tblastn", "tblastx"])
def test_reference_bound_to_the_library_prints_the_reference_output(tmp_path, name, variant, threads):
    exe = need(variant)
    case, g, args = case_args(tmp_path, name)
    run = lambda extra: subprocess.run([exe] + args + ["-a", str(threads)] + extra, capture_output=True, text=True, check=True).stdout
    assert run(["-m", "8", "-b", str(case.keep)]) == g["tsv"]
    assert run(["-m", "7", "-b", str(g["nalign"])]) == g["xml_align"]
    if threads > 1:
        return                                   # the other views add nothing about concurrency
    assert run(["-m", "7", "-b", "0"]) == g["xml"]
    plain = run(["-m", "0", "-b", str(g["nalign"])])
    assert plain[plain.index("Sequences producing"):] == g["plain_align"]
    t9 = run(["-m", "9", "-b", str(case.keep)]).split("\n")[1:]           # first line carries the compile date
    want = g["tsv9"].split("\n")
    assert t9[0] == want[0] and t9[1].startswith("# Database: ") and t9[2:] == want[2:]      # the database path differs


@pytest.mark.parametrize("variant", ["scores", "topk", "group"])
@pytest.mark.parametrize("hv", ["plain_gis_taxid", "masked", "masked_gis_taxid", "taxlist", "masked_taxlist"])
def test_bound_reference_with_masks_and_taxid_lists(tmp_path, hv, variant):
    from test_host_cpu import build_headers_db, HEADER_VARIANTS
    exe = need(variant)
    case, vol, masked, tx = build_headers_db(tmp_path)
    ref = load_golden("headers")["variants"][hv]
    dbn, flags, taxlist = HEADER_VARIANTS[hv]
    qf = str(tmp_path / "q.fa")
    open(qf, "w").write(">query test\n" + "".join(blastdb.NCBISTDAA[c] for c in case.query) + "\n")
    args = [exe, "-d", vol if dbn == "vol" else masked, "-i", qf, "-v", str(case.keep), "-e", "1e6", "-a", "3"]      # group: 3 shards on device 0
    args += (["-I"] if flags & 1 else []) + (["-H"] if flags & 2 else []) + (["-x", tx] if taxlist else [])
    run = lambda extra: subprocess.run(args + extra, capture_output=True, text=True, check=True).stdout
    assert run(["-m", "7", "-b", "5"]) == ref["m7"]
    assert run(["-m", "8", "-b", str(case.keep)]) == ref["m8"]
    plain = run(["-m", "0", "-b", "5"])
    assert plain[plain.index("Sequences producing"):] == ref["m0"]


@pytest.mark.parametrize("variant", ["scores", "topk", "group"])
def test_bound_reference_with_a_query_file(tmp_path, variant):
    """the per-query reset (amd_queryno) over a file of several queries, an empty one among them"""
    exe = need(variant)
    g = load_golden("multiquery")
    case = cases.get("edges")
    base = str(tmp_path / "db")
    blastdb.write_db(base, case.seqs, protein=True)
    qf = str(tmp_path / "q.fa")
    open(qf, "w").write(g["query_text"])
    for threads in ("1", "4"):
        r = subprocess.run([exe, "-d", base, "-i", qf, "-m", "8", "-b", "10", "-v", "12", "-e", "1000", "-a", threads], capture_output=True, text=True)
        assert r.returncode == g["rc8"], r.stderr
        assert r.stdout == g["m8"]


def _option_runs():
    g = load_golden("options")
    return [(name, i) for name in g for i in range(len(g[name]["runs"]))]


@pytest.mark.parametrize("name,i", _option_runs())
def test_bound_reference_under_threshold_and_strand_options(tmp_path, name, i):
    """the options of tests/golden/options.json through the reference bound to the library: hits_init computes the
    thresholds (reference code), binding B hands them to the device-side acceptance test, binding A feeds hits_enter"""
    g = load_golden("options")[name]
    case = cases.get(name)
    rec = g["runs"][i]
    base = str(tmp_path / name)
    blastdb.write_db(base, case.seqs, protein=case.protein, volumes=case.volumes)
    alpha = blastdb.NCBI4NA if case.query_is_nt else blastdb.NCBISTDAA
    qf = str(tmp_path / "q.fa")
    open(qf, "w").write(">query test\n" + "".join(alpha[c] for c in case.query) + "\n")
    args = ["-d", base, "-i", qf, "-p", str(case.sym)] + (["-Q", str(case.query_gencode), "-D", str(case.db_gencode)] if case.sym >= 2 else [])
    for variant, tail, key in (("topk", ["-m", "8", "-a", "2"], "m8"), ("topk", ["-m", "7", "-b", "0"], "m7"), ("scores", ["-m", "8"], "m8")):
        r = subprocess.run([need(variant)] + args + rec["options"] + tail, capture_output=True, text=True)
        assert r.returncode == rec[key + "_rc"] and r.stdout == rec[key], (variant, rec["options"], r.stderr)


def test_one_query_file_through_the_reference_the_bound_reference_and_the_cli_at_scale():
    """the fixtures above hold a thousand sequences; this is the same comparison where searches take long enough for kernels
    to meet on the device (tools/probe.py dropin, which found round 3's follower hang at 10 M sequences): a 2 M-sequence
    database written as BLAST v4 volumes, an 8-query file of 332..410-aa database sequences, -m 8 with 250 alignments,
    through the unmodified reference, the reference bound to the library three ways and swipe_amd_cli (which pairs adjacent
    queries) - the tool fails unless all five outputs are identical"""
    import sys
    for v in BOUND:
        need(v)
    from conftest import under_interpreter
    nseq = "30000" if under_interpreter() else "2000000"     # (interpreted kernels: the comparison, not the scale)
    r = subprocess.run([sys.executable, "-u", os.path.join(ROOT, "tools", "probe.py"), "dropin", "--nseq", nseq, "--nq", "8",
                        "--ref-queries", "8", "--reps", "1"], capture_output=True, text=True, timeout=2400 if under_interpreter() else 600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    assert "output: identical" in r.stdout and "swipe_amd_cli" in r.stdout
