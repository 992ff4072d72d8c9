"""stage-by-stage smoke with a watchdog: prints where a hang sits (faulthandler dumps the Python stack after 60 s)"""
import faulthandler, os, sys, time
faulthandler.dump_traceback_later(60, exit=True)
sys.path.insert(0, '.')
import numpy as np
np.seterr(over='ignore')
def say(*a): print(time.strftime("%H:%M:%S"), *a, flush=True)
say("import")
import swipe_amd
from swipe_amd import blastdb, synth
say("devices", swipe_amd._lib.load().swa_device_count())
q = blastdb.encode_protein(synth.QUERY_P07327)
res, off = swipe_amd.synth_db(1, 4000, query=q)
say("synth")
db = swipe_amd.Database.from_arrays(res, off, device=0)
say("opened")
db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
say("scoring")
for host, follow in ((1, 0), (0, 0), (0, 1)):
    db.set_option("requeue_host", host)
    db.set_option("requeue_follow", follow)
    say("requeue_host", host, "follow", follow)
    scores, c = db.search(q)
    say("search", c)
    hits, total, obvious, c = db.search_topk(q, keep=10, minscore=40)
    say("topk", hits[:3], total, c)
    db.set_option("bound", 1)
    hits, total, obvious, c = db.search_topk(q, keep=10, minscore=40)
    say("topk bound", hits[:3], total, c)
    db.set_option("bound", None)
db.close()
say("done")
