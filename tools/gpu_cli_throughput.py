"""End-to-end CLI on a query file: wall time per query against the search kernel's own time.
Writes a synthetic BLAST v4 protein database to local disk, runs swipe_amd_cli on 1 and then N queries
(-m 8 tabular and -m 0 with alignments) and reports (T_N - T_1) / (N - 1) = steady-state seconds per query."""
import os, sys, time, tempfile, subprocess, numpy as np
np.seterr(over='ignore')
sys.path.insert(0, '.')
import swipe_amd
from swipe_amd import synth, blastdb
nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 16
q = blastdb.encode_protein(synth.QUERY_P07327)
res, off = swipe_amd.synth_db(1, nseq, query=q)
d = tempfile.mkdtemp(prefix="cli_", dir="/tmp")
nvol = max(1, int(np.ceil((off[-1] + nseq) / 3.5e9)))
names = []
for v in range(nvol):
    lo, hi = nseq * v // nvol, nseq * (v + 1) // nvol
    name = os.path.join(d, "db.%02d" % v)
    blastdb.write_protein_volume_arrays(name, res, off[lo:hi + 1], first_id=lo)
    names.append(name)
blastdb.write_alias(os.path.join(d, "db"), names, protein=True)
# queries: database sequences of about the bench query's length (so every one has real hits)
lens = np.diff(off)
pick = np.nonzero((lens > 330) & (lens < 420))[0][:: max(1, nseq // 200)][:nq]
sym = "-ABCDEFGHIKLMNPQRSTVWXYZU*OJ"
def fasta(ids, path):
    with open(path, "w") as f:
        for i in ids:
            f.write(">q%d\n%s\n" % (i, "".join(sym[c] for c in res[off[i]:off[i + 1]])))
fasta(pick[:1], os.path.join(d, "q1.fa"))
fasta(pick, os.path.join(d, "qn.fa"))
cli = os.path.join(os.path.dirname(swipe_amd.__file__), "swipe_amd_cli")
db = swipe_amd.Database.from_arrays(res, off)
db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
kms = []
for i in pick:
    db.search_topk(res[off[i]:off[i + 1]], keep=250, minscore=80)
    kms.append(db.search_topk(res[off[i]:off[i + 1]], keep=250, minscore=80)[3]["total_ms"])
db.close()
print("library search step (device time, top-250): mean %.1f ms per query over %d queries" % (np.mean(kms), len(pick)))
for mode, extra in (("-m 8", ["-m", "8"]), ("-m 0 (250 alignments)", ["-m", "0"]), ("-m 7 xml", ["-m", "7"])):
    ts = []
    for qf in ("q1.fa", "qn.fa"):
        best = 1e9
        for _ in range(2):
            t = time.time()
            r = subprocess.run([cli, "-d", os.path.join(d, "db"), "-i", os.path.join(d, qf), "-o", os.path.join(d, "out.txt")] + extra,
                               capture_output=True, text=True)
            best = min(best, time.time() - t)
            if r.returncode:
                print(r.stderr[-500:]); sys.exit(1)
        ts.append(best)
    per = (ts[1] - ts[0]) / (len(pick) - 1)
    print("%-24s 1 query %.2f s, %d queries %.2f s -> %.1f ms per query in steady state (search step alone %.1f ms: %.0f %% of it)" % (
        mode, ts[0], len(pick), ts[1], per * 1e3, np.mean(kms), 100 * np.mean(kms) / (per * 1e3)))
subprocess.run(["rm", "-rf", d])
