"""Config 4 probe: 1 kb DNA query, both strands, synthetic nt db (+1/-3, gap 5/2)."""
import os, sys, time, numpy as np
np.seterr(over='ignore')
sys.path.insert(0, '.')
import swipe_amd, oracle
from swipe_amd import synth, blastdb
nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
rtab = synth.residue_table_nucleotide()
q = synth._random_residues(99, 1, 1000, rtab)
qm = blastdb.revcomp_nt16(q)
res, off = swipe_amd.synth_db(3, nseq, protein=False)
t = time.time()
db = swipe_amd.Database.from_arrays(res, off, symtype=0)
print("load", time.time() - t, db.info())
db.set_scoring(swipe_amd.matrix_nucleotide(1, -3), 5, 2)
s1, s2, c = db.search2(q, qm)
pick = np.random.default_rng(2).integers(0, nseq, 1500)
r2, o2 = oracle.pack([res[off[i]:off[i+1]] for i in pick])
Mo = oracle.matrix_nucleotide(1, -3)
print("parity", np.array_equal(s1[pick], oracle.search_all63(r2, o2, q, Mo, 7, 2, threads=os.cpu_count())),
      np.array_equal(s2[pick], oracle.search_all63(r2, o2, qm, Mo, 7, 2, threads=os.cpu_count())))
for _ in range(3):
    _, _, c = db.search2(q, qm, want_scores=False)
    print("dual: %.0f GCUPS kernel (%.2f ms), total %.0f GCUPS" % (c['cells'] / c['kernel_ms'] / 1e6, c['kernel_ms'], c['cells'] / c['total_ms'] / 1e6), c['narrow_rows'])
for _ in range(2):
    _, c = db.search(q, want_scores=False)
    print("single strand pair kernel: %.0f GCUPS kernel (%.2f ms) rows %d" % (c['cells'] / c['kernel_ms'] / 1e6, c['kernel_ms'], c['narrow_rows']))
