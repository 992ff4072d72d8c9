"""Cold path: BLAST v4 volumes on local disk -> swa_db_open (read + PCIe + format) -> first search."""
import os, sys, time, tempfile, subprocess, numpy as np
np.seterr(over='ignore')
sys.path.insert(0, '.')
import swipe_amd
from swipe_amd import synth, blastdb
nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
q = blastdb.encode_protein(synth.QUERY_P07327)
res, off = swipe_amd.synth_db(1, nseq, query=q)
d = tempfile.mkdtemp(prefix="cold_", dir="/tmp")
nvol = max(1, int(np.ceil((off[-1] + nseq) / 3.5e9)))
t = time.time()
names = []
for v in range(nvol):
    lo, hi = nseq * v // nvol, nseq * (v + 1) // nvol
    name = os.path.join(d, "db.%02d" % v)
    blastdb.write_protein_volume_arrays(name, res, off[lo:hi + 1], first_id=lo)
    names.append(name)
blastdb.write_alias(os.path.join(d, "db"), names, protein=True)
print("wrote %d volumes, %.2f GB in %.1f s" % (nvol, (off[-1] + nseq) / 1e9, time.time() - t))
os.sync()
try:
    open("/proc/sys/vm/drop_caches", "w").write("3\n")
    dropped = True
except Exception:
    dropped = False
for rep in range(2):
    t = time.time()
    db = swipe_amd.Database.open(os.path.join(d, "db"))
    t_open = time.time() - t
    db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
    t = time.time()
    hits, tot, obv, c = db.search_topk(q, keep=250, minscore=40)
    t_search = time.time() - t
    print("%s open (disk -> HBM, formatted): %.2f s; first search %.3f s; top hit %s" % ("cold" if rep == 0 and dropped else "warm page cache", t_open, t_search, hits[0]))
    db.close()
subprocess.run(["rm", "-rf", d])
