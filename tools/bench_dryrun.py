"""bench.py end to end WITHOUT a GPU, on tools/gfx950sim: every code path of the default run - headline, exact and round-3 A/B
blocks, oracle verification, cpu_baseline, cold_open (incl. the budgeted open), the secondary sections - at sizes the
interpreter finishes in minutes.  The numbers mean nothing; what it shows is that the script that the driver runs at round end
still runs (bench.py changes in rounds 5 and 6 have never met hardware).  torch.cuda is not there, so its three calls bench.py
makes outside torch.distributed are stubbed here (is_available / set_device / synchronize); everything else is bench.py's own.

    tools/gfx950sim/run.sh python tools/bench_dryrun.py [bench.py arguments]
"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("HIPSIM") != "1":
    raise SystemExit("run under tools/gfx950sim/run.sh")
import torch

torch.cuda.is_available = lambda: True
torch.cuda.set_device = lambda d: None
torch.cuda.synchronize = lambda *a, **k: None
os.environ.setdefault("SWA_BENCH_BACKEND", "gloo")        # collective tensors on the host
args = sys.argv[1:] or ["--nseq", "40000", "--steps", "2", "--warmup", "1", "--no-live-traffic", "--quick",
                        "--secondary-nt-nseq", "20000", "--secondary-protein-nseq", "40000"]
sys.argv = [os.path.join(ROOT, "bench.py")] + args
runpy.run_path(sys.argv[0], run_name="__main__")
