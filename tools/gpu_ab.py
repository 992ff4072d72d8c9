"""A/B: plain vs row-shifted narrow kernel, interleaved rounds in one process (parity + speed)."""
import os, sys, numpy as np
np.seterr(over='ignore')
sys.path.insert(0, '.')
import swipe_amd, oracle
from swipe_amd import synth, blastdb
q = blastdb.encode_protein(synth.QUERY_P07327)
nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
res, off = swipe_amd.synth_db(1, nseq, query=q)
M = swipe_amd.matrix_builtin("blosum62")
VARIANTS = [int(x) for x in os.environ.get('AB_VARIANTS', '2,3').split(',')]
dbs = {}
for v in VARIANTS:
    dbs[v] = swipe_amd.Database.from_arrays(res, off)
    dbs[v].set_option("narrow_variant", str(v))
    dbs[v].set_scoring(M, 11, 1)
s1, c1 = dbs[VARIANTS[0]].search(q)
s2, c2 = dbs[VARIANTS[-1]].search(q)
print("variants agree:", np.array_equal(s1, s2), "requeued", c1["wide"], c2["wide"], "shifted flags", c1["narrow_shifted"], c2["narrow_shifted"])
pick = np.random.default_rng(1).integers(0, nseq, 3000)
r2, o2 = oracle.pack([res[off[i]:off[i+1]] for i in pick])
want = oracle.search_all63(r2, o2, q, oracle.matrix_builtin("BLOSUM62"), 12, 1, threads=os.cpu_count())
print("sample vs oracle:", np.array_equal(s2[pick], want))
cells = c1["cells"]
for rnd in range(5):
    out = []
    for v in VARIANTS:
        _, c = dbs[v].search(q, want_scores=False)
        out.append("v%d %.0f GCUPS (%.2f ms)" % (v, cells / c["kernel_ms"] / 1e6, c["kernel_ms"]))
    print("  ".join(out))
