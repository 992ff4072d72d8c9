#!/usr/bin/env python3
"""Headline benchmark of BASELINE.json: GCUPS of a 375-aa query against ONE 10M-sequence synthetic protein
database (configs[1]) on N MI355X.  N > 1: the database is cut into N read-only shards of near-equal RESIDUE
count (parallel.shard_bounds), one per GPU - strong scaling, the metric's "at 1/2/4/8 GPUs" -
and the per-shard top-K lists are merged after one all_gather over RCCL.

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...   (no launcher: re-executes itself as the line above on a free port)

    --weak                      every rank its own --nseq sequences (N x 10M database)
    --workload protein100M      BASELINE.json configs[4]: 100 M proteins over the N ranks (12.5 M each at N = 8)
    --workload nucleotide       configs[3] as the only section (it also runs as a secondary section by default)
    --quick                     secondary sections at reduced size (nucleotide 10 M sequences, no 100 M-protein section)
    --predict-scaling           one GPU: every shard of the 10 M database at N = 1, 2, 4, 8 (parallel.shard_bounds) timed on
                                its own + the measured gather latency -> predicted strong-scaling curve

At N = 1 the default run also carries, as `secondary` entries under the same clock, BASELINE.json configs[3] at its stated
size (1 kb DNA query, both strands, 50 M sequences) and configs[4]'s whole database on the one GPU (100 M proteins).

A step = one complete search of the resident shard: query + scoring upload, first-pass kernel, re-queue kernel,
device-side hit filter, top-250 back on the host, (N>1) all_gather + merge.  The database is already formatted in
HBM when the timed region starts.  Rank 0 prints ONE JSON line.  See DESIGN.md "Measurement".
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np

np.seterr(over="ignore")

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec
KEEP = 250                       # reference default hit list length (max(-v,-b), hits.cc:287)
CLOCK_GHZ = 2.4
# VALU instructions per cell pair of each first-pass form (counters.narrow_shifted, include/swipe_amd.h)
OPS = {0: 8.5, 1: 7.5, 2: 7.5, 3: 7.5, 4: 6.5, 5: 7.5, 6: 6.5, 7: 7.5, 8: 6.0, 9: 6.0, 10: 5.0}
KERNEL = {0: "swa_narrow_kernel<%d>", 1: "swa_narrow_split_kernel<%d, W, 16>", 2: "swa_narrow_split_kernel<%d, W, 8>",
          3: "swa_narrow_split_kernel<%d, W, 4>", 4: "swa_dual_kernel<%d, W, NRES, G>",
          5: "swa_narrow_split_kernel<%d, 2, 16, PIPE, DEFER, MP> (one launch per pass)",
          6: "swa_dual_kernel<%d, W, NRES, 16, MP> (one launch per pass)", 7: "swa_narrow_split_kernel<%d, W, 2>",
          8: "swa_narrow_bound_kernel<%d, W, G, 16> (bound build of the first pass; sequences at or above the score "
             "threshold recomputed exactly by the 32-bit wave kernel inside the step)",
          9: "swa_narrow_bound_kernel<%d, 2, 16, 16, MP> (one launch per pass)", 10: "swa_dual_bound_kernel<%d>"}


def kernel_name(form, rows, qlen):
    """the first-pass kernel as rocprofv3 names it, where the template arguments follow from (form, rows, qlen) alone"""
    if form == 8:                                        # swipe_amd.cpp run_search + sw_cb_kernel.inc cb_waves_for
        if qlen <= 48:
            return "swa_one_bound_kernel<%d, W, true> (bound build, one lane per sequence pair)" % rows
        G = 2 if qlen <= 96 else 4 if qlen <= 192 else 8 if qlen <= 384 else 16
        W = 8 if rows <= 5 else 6 if rows <= 10 else 4 if rows <= 20 else 3 if rows <= 29 else 2
        return ("swa_narrow_bound_kernel<%d, %d, %d, 16, false> (bound build of the first pass; sequences at or above the score "
                "threshold recomputed exactly by the 32-bit wave kernel beside it)" % (rows, W, G))
    return KERNEL.get(form, "first-pass kernel, %d rows per lane") % rows


def roofline_blocks(nsym, nseq, cells, k_ms, form, rows, bytes_per_residue=1.0, traffic=None, traffic_source=None, qlen=375):
    """roofline (HBM, SURVEY.md 8(d): 1 B per residue + 12 B per sequence per launch) and the VALU-issue model"""
    alg_bytes = int(nsym * bytes_per_residue) + 12 * nseq
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9
    ops = OPS.get(form, 7.5)
    r = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
         "kernel": kernel_name(form, rows, qlen), "kernel_ms": round(k_ms, 3),
         "algorithmic_bytes_per_launch": alg_bytes,
         "note": "integer DP at hundreds of cells per residue byte is VALU-issue-bound, not HBM-bound; see valu_roofline"}
    if traffic_source:
        r["traffic_source"] = traffic_source
    v = {"achieved_gcups_kernel": round(cells / (k_ms * 1e-3) / 1e9, 1),
         "peak_gcups": round(256 * 4 * CLOCK_GHZ * 1e9 / 4 * 128 / ops / 1e9, 1),
         "model": "256 CU x 4 SIMD x %.1f GHz / 4 cycles per VOP3P wave64 op x 128 cells / %.1f ops per cell pair" % (CLOCK_GHZ, ops)}
    v["frac"] = round(v["achieved_gcups_kernel"] / v["peak_gcups"], 4)
    return r, v


def committed_traffic(key, nseq):
    """HBM bytes per launch from the committed PMC passes of the default bench command (profiles/hbm_traffic.json, written by
    tools/summarise_profiles.py): counters cannot be collected inside a timed run, so the line says where the number comes
    from.  Looked up by (workload, database size); None when this size was not profiled."""
    tf = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        for rec in json.load(open(tf)).get("records", []):
            if rec.get("workload") == key and rec.get("nseq") == nseq:
                return rec.get("bytes_per_launch"), rec.get("source")
    except Exception:
        pass
    return None, None


def traffic_child(nseq, device):
    """--traffic-child: the headline step twice (one warm-up, one counted) on a database of nseq sequences, nothing else - what
    live_traffic() runs under rocprofv3 --pmc.  Prints the kernel the step ran."""
    import swipe_amd
    from swipe_amd import blastdb, synth
    q = blastdb.encode_protein(synth.QUERY_P07327)
    res, off = swipe_amd.synth_db(1, nseq, query=q, threads=os.cpu_count() or 1)
    db = swipe_amd.Database.from_arrays(res, off, device=device)
    db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
    st = swipe_amd.stats_init(qlen=len(q), db_seqcount=nseq, db_symcount=int(off[-1]))
    for _ in range(2):
        c = db.search_topk(q, keep=KEEP, minscore=st.scorethreshold, maxscore=st.upperscorethreshold)[3]
    db.close()
    print("TRAFFIC_CHILD", c["narrow_shifted"], c["narrow_rows"], flush=True)


def live_traffic(nseq, device, budget_s=45):
    """HBM bytes per launch of the headline step's first-pass kernel, measured NOW: the step re-run in a child process under
    `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and again under `--pmc WRITE_SIZE` (separate passes, counters only, as the
    MI355X guide prescribes), corrected as profiles/r03_fetch_calibration.txt found for this kernel's loads (FETCH_SIZE counts
    half of a coalesced 2-byte-per-lane read: x 2; both counters are in KiB).  Returns (bytes, source) or (None, why)."""
    import csv
    import glob
    import shutil
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 not on PATH"
    got = {}
    t0 = time.time()
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="swa_pmc_", dir="/tmp")
        try:
            env = dict(os.environ, TMPDIR="/tmp")
            # its own session: a pass that overruns is ended as a GROUP (rocprofv3 and the python under it), so that nothing of
            # it is still on the GPU while the sections after this one are timed
            pr = subprocess.Popen([exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable,
                                   os.path.join(ROOT, "bench.py"), "--traffic-child", "--nseq", str(nseq), "--device", str(device)],
                                  stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd="/tmp", env=env, start_new_session=True)
            try:
                # both passes together stay inside 60 s whatever happens: 45 s shared, the second at least 15 s
                so, se = pr.communicate(timeout=max(1.0, min(max(15.0, budget_s - (time.time() - t0)), 60.0 - (time.time() - t0))))
            except subprocess.TimeoutExpired:
                os.killpg(pr.pid, 9)
                pr.communicate()
                return None, "rocprofv3 --pmc %s pass did not finish inside its %d s budget" % (counter, budget_s)
            if pr.returncode != 0 or "TRAFFIC_CHILD" not in so:
                return None, "rocprofv3 --pmc %s failed: %s" % (counter, (se or so)[-200:].replace("\n", " "))
            per = {}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row.get("Counter_Name") == counter and "swa_" in row.get("Kernel_Name", ""):
                        per.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
            first = {k: v for k, v in per.items() if "narrow" in k or "one_" in k}
            if not first:
                return None, "no first-pass kernel in the %s pass" % counter
            name = max(first, key=lambda k: sum(first[k]))
            got[counter] = (name, first[name][-1])              # the counted launch (the last one)
        except Exception as e:       # never lose the bench line over its evidence
            return None, "live traffic failed: %s" % e
        finally:
            subprocess.run(["rm", "-rf", d])
    if got["FETCH_SIZE"][0] != got["WRITE_SIZE"][0]:
        return None, "the two passes ran different kernels"
    b = int(got["FETCH_SIZE"][1] * 1024 * 2 + got["WRITE_SIZE"][1] * 1024)
    return b, ("measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (two passes of the headline step in a child "
               "process, %.0f s), FETCH_SIZE x 2 per profiles/r03_fetch_calibration.txt, kernel %s" % (time.time() - t0, got["FETCH_SIZE"][0][:60]))


def reference_cli_rate(d, base, query_text, threads, extra, cells1, budget_s):
    """GCUPS of oracle/_ref/swipe (the reference, compiled from /root/reference by oracle/Makefile) at `threads`"""
    exe = os.path.join(ROOT, "oracle", "_ref", "swipe")

    def run(reps):
        qf = os.path.join(d, f"q{threads}_{reps}.fa")
        with open(qf, "w") as f:
            for i in range(reps):
                f.write(f">q{i}\n{query_text}\n")
        out = subprocess.run([exe, "-d", base, "-i", qf, "-a", str(threads), "-v", "5", "-b", "0"] + extra,
                             capture_output=True, text=True, check=True).stdout
        return [float(x) for x in re.findall(r"Elapsed:\s+([0-9.]+)s", out)]

    first = run(1)
    per = max(first[0], 0.01)
    reps = int(min(200, max(2, round(budget_s / per))))
    el = run(reps)
    total = sum(el)
    return (cells1 * len(el) / total / 1e9, len(el), total) if total > 0 else (0.0, 0, 0.0)


def cpu_baseline(res, off, query_text, cores, *, protein=True, sample_seqs=3_000_000):
    """Rank 0, N=1 only: the reference SSSE3 path (oracle/_ref/swipe) on the host cores over a bounded sample of the same
    database, swept over its -a thread counts (the reference takes hitsmutex once per database sequence, hits.cc:172, so
    it has a knee): the BEST setting is reported, with the per-thread figure and the whole sweep.
    Falls back to the oracle port if the binary is absent."""
    import swipe_amd
    n = min(sample_seqs, len(off) - 1)
    nres = int(off[n] - off[0])
    cells1 = nres * len(query_text) * (1 if protein else 2)
    exe = os.path.join(ROOT, "oracle", "_ref", "swipe")
    if os.path.exists(exe):
        d = tempfile.mkdtemp(prefix="swa_cpu_")
        try:
            base = os.path.join(d, "sample")
            swipe_amd.write_blastdb(base, res, off[: n + 1], symtype=1 if protein else 0)
            extra = [] if protein else ["-p", "0", "-r", "1", "-q", "-3", "-G", "5", "-E", "2"]
            sweep = {}
            cand = sorted({t for t in (8, 16, 24, 32, 64, 128, 256) if t <= max(cores, 8)} | {min(cores, 256)})
            for t in cand:
                g, k, tot = reference_cli_rate(d, base, query_text, t, extra, cells1, budget_s=3.0)
                sweep[t] = (round(g, 2), k, round(tot, 2))
            best = max(sweep, key=lambda t: sweep[t][0])
            return {"value": sweep[best][0], "unit": "GCUPS", "cores": best, "kind": "reference",
                    "per_thread": round(sweep[best][0] / best, 3), "host_cores": cores,
                    "sweep": {str(t): v[0] for t, v in sweep.items()},
                    "sample": f"oracle/_ref/swipe (SSSE3 path), best of -a {cand}: {sweep[best][1]} x {len(query_text)}-"
                              f"{'aa' if protein else 'nt (both strands)'} query vs the first {n} sequences ({nres} residues) of "
                              f"the same db; sum of its own 'Elapsed' = {sweep[best][2]}s"}
        finally:
            subprocess.run(["rm", "-rf", d])
    import oracle
    from swipe_amd import blastdb as b
    n = min(200_000, len(off) - 1)
    q = b.encode_protein(query_text) if protein else b.encode_nucleotide(query_text)
    M = oracle.matrix_builtin("BLOSUM62") if protein else oracle.matrix_nucleotide(1, -3)
    t = time.time()
    oracle.search_all63(res[: off[n]], off[: n + 1], q, M, 12 if protein else 7, 1 if protein else 2, threads=cores)
    dt = time.time() - t
    return {"value": round(int(off[n]) * len(q) / dt / 1e9, 2), "unit": "GCUPS", "cores": cores, "kind": "port",
            "per_thread": round(int(off[n]) * len(q) / dt / 1e9 / cores, 3),
            "sample": f"oracle scalar 63-bit recurrence (one strand), {cores} threads, first {n} sequences"}


def verify_against_oracle(db, res, off, lo, q, M_name, goe, ge, hits, tot, minscore, maxscore, sample, threads):
    """The checker leg: all scores of the shard from the exact first pass (swa_search), then (1) the hit list and
    totalhits recomputed on the host from those scores over EVERY sequence of the shard, (2) the oracle's scalar 63-bit
    recurrence on the hits' sequences and a seeded random sample.  Returns (verified, mismatches)."""
    import oracle
    scores, _ = db.search(q)
    n = len(off) - 1
    idx = np.flatnonzero(scores >= minscore)
    order = idx[np.lexsort((-idx, -scores[idx]))]
    kept = order[scores[order] <= maxscore][:KEEP]
    bad = 0
    mine = [(int(s), int(v)) for s, v in hits if lo <= s < lo + n]
    want = [(int(lo + i), int(scores[i])) for i in kept]
    # the local list is what this shard contributes: with one shard it must equal the merged list
    if want[: len(mine)] != mine:
        bad += 1
    rng = np.random.default_rng(20260929 + lo)
    pick = np.unique(np.concatenate([rng.integers(0, n, size=min(sample, n)), np.array([s - lo for s, _ in mine], dtype=np.int64)]))
    lens = off[pick + 1] - off[pick]
    o2 = np.zeros(len(pick) + 1, dtype=np.int64)
    np.cumsum(lens, out=o2[1:])
    r2 = np.empty(int(o2[-1]), dtype=np.uint8)
    for k, i in enumerate(pick):
        r2[o2[k]:o2[k + 1]] = res[off[i]:off[i + 1]]
    Mo = oracle.matrix_builtin(M_name) if isinstance(M_name, str) else M_name
    ref = oracle.search_all63(r2, o2, q, Mo, goe, ge, threads=threads)
    bad += int((ref != scores[pick]).sum())
    hit_scores = dict(mine)
    for k, i in enumerate(pick):
        s = int(lo + i)
        if s in hit_scores and hit_scores[s] != int(ref[k]):
            bad += 1
    return len(pick), bad, int((scores >= minscore).sum())


def cold_open(res, off, device, q, minscore, maxscore, want_hits):
    """disk -> HBM -> first hits: the shard written as BLAST v4 volumes to local disk, then (1) page cache dropped if allowed,
    swa_db_open; (2) the same warm; (3) swa_db_open_async + the first top-K search following the loader; (4) swipe_amd_cli with
    that one query, process start to process end (what a SWIPE user types).  Every hit list must equal the timed region's."""
    import swipe_amd
    from swipe_amd import blastdb
    d = tempfile.mkdtemp(prefix="swa_cold_")
    try:
        base = os.path.join(d, "db")
        n = len(off) - 1
        # .psq offsets are 32 bit: volumes of at most ~3.9 G residues behind an alias
        cuts, vols = [0], []
        while cuts[-1] < n:
            hi = int(np.searchsorted(off, off[cuts[-1]] + 3_900_000_000, side="right")) - 1
            cuts.append(min(n, max(hi, cuts[-1] + 1)))
        for v in range(len(cuts) - 1):
            name = f"{base}.{v:02d}"
            swipe_amd.write_blastdb(name, res, off[cuts[v]: cuts[v + 1] + 1], first_id=cuts[v])
            vols.append(name)
        if len(vols) > 1:
            blastdb.write_alias(base, vols, protein=True)
        else:
            base = vols[0]
        gb = (int(off[-1] - off[0]) + n) / 1e9

        def drop():
            os.sync()
            try:
                with open("/proc/sys/vm/drop_caches", "w") as f:
                    f.write("3\n")
                return True
            except OSError:
                return False

        M = swipe_amd.matrix_builtin("BLOSUM62")
        dropped = drop()
        t0 = time.time()
        db = swipe_amd.Database.open(base, device=device)
        cold = time.time() - t0
        db.close()
        t0 = time.time()
        db = swipe_amd.Database.open(base, device=device)
        warm = time.time() - t0
        db.close()
        t0 = time.time()
        db = swipe_amd.Database.open(base, device=device, wait=False)
        t_ret = time.time() - t0
        db.set_scoring(M, 11, 1)
        hits, tot, obv, c = db.search_topk(q, keep=KEEP, minscore=minscore, maxscore=maxscore)
        first_hits = time.time() - t0
        db.wait()
        resident = time.time() - t0
        db.close()
        if [tuple(h) for h in hits] != want_hits:
            raise SystemExit("bench: the search that followed the loader disagrees with the resident shard's hit list")
        out = {"open_s": round(warm, 3), "open_cold_s": round(cold, 3), "page_cache_dropped": dropped, "volumes": len(vols),
               "file_gb": round(gb, 3), "cold_gb_per_s": round(gb / cold, 2), "warm_gb_per_s": round(gb / warm, 2),
               "async_open_returns_s": round(t_ret, 3), "first_hits_s": round(first_hits, 3), "resident_s": round(resident, 3),
               "first_search_parts": int(c["loading_parts"]),
               "what": "swa_db_open of the shard from BLAST v4 volumes on the box's local disk, pipelined (index walk | reader threads -> "
                       "page-locked ring -> copy engine -> per-chunk terminator strip + per-part format kernels): open_cold_s with the "
                       "page cache dropped (disk-bound: cold_gb_per_s is the disk's rate), open_s warm; first_hits_s = swa_db_open_async "
                       "+ swa_set_scoring + the first top-250 search following the loader part by part (hit list identical)"}
        # the same volumes as a shard over its HBM budget (a quarter of its resident footprint): the open returns when the index
        # is read and the parts are planned, a loader fills the parts' page-locked blocks from the files behind it, and the
        # first search binds the parts as they arrive (round 6)
        try:
            budget = max(int((2.04 * int(off[-1] - off[0]) + 77 * n) / 4), 48 << 20)       # (two slots, each with its 8 MiB reserve)
            t0 = time.time()
            db = swipe_amd.Database.open(base, device=device, hbm_budget=budget)
            b_ret = time.time() - t0
            db.set_scoring(M, 11, 1)
            bhits, _, _, _ = db.search_topk(q, keep=KEEP, minscore=minscore, maxscore=maxscore)
            b_first = time.time() - t0
            db.wait()
            b_all = time.time() - t0
            prog = db.load_progress()
            db.close()
            if [tuple(h) for h in bhits] != want_hits:
                raise SystemExit("bench: the budgeted shard's first search disagrees with the resident shard's hit list")
            out.update({"budgeted_open_s": round(b_ret, 3), "budgeted_first_hits_s": round(b_first, 3), "budgeted_all_parts_s": round(b_all, 3),
                        "budgeted_parts": int(prog["parts_total"]), "budgeted_page_locked_gb": round(prog["bytes_total"] / 1e9, 3),
                        "budgeted_what": "swa_db_open_streamed at an HBM budget of a quarter of the resident footprint, warm page cache: the call "
                                         "returns (budgeted_open_s), the first top-250 search over all parts through the two device slots is done "
                                         "(budgeted_first_hits_s, hit list identical), every part is in page-locked memory (budgeted_all_parts_s)"})
        except SystemExit:
            raise
        except Exception as e:
            out["budgeted_open_s"] = None
            out["budgeted_what"] = f"failed: {e}"
        # through the command line: one query, process start -> process end
        cli = os.path.join(os.path.dirname(os.path.abspath(swipe_amd.__file__)), "swipe_amd_cli")
        if os.path.exists(cli):
            from swipe_amd import synth
            qf, of = os.path.join(d, "q1.fa"), os.path.join(d, "out.txt")
            with open(qf, "w") as f:
                f.write(">P07327\n" + synth.QUERY_P07327 + "\n")
            best = None
            for _ in range(2):
                t0 = time.time()
                r = subprocess.run([cli, "-d", base, "-i", qf, "-o", of, "-m", "8", "-v", str(KEEP), "-b", str(KEEP), "-e", "10"],
                                   capture_output=True, text=True)
                dt = time.time() - t0
                if r.returncode:
                    raise SystemExit("bench: swipe_amd_cli failed: " + r.stderr[-300:])
                best = dt if best is None else min(best, dt)
            ids = [int(l.split("\t")[1].split("|")[1][1:]) for l in open(of) if l.strip() and not l.startswith("#")]
            if ids != [h[0] for h in want_hits]:
                raise SystemExit("bench: swipe_amd_cli lists other sequences than the timed region")
            out["first_query_s"] = round(best, 3)
            out["first_query_what"] = ("swipe_amd_cli -d db -i one_query.fa -m 8 -v 250 -b 250 -e 10, warm page cache, process start -> process "
                                       "end (open, search following the loader, 250 alignments, output); best of 2; hit list identical")
        return out
    finally:
        subprocess.run(["rm", "-rf", d])


def group_section(res, off, device, q, minscore, maxscore, want_sha1, steps=3):
    """the C++ multi-device path under the driver's clock: swa_group with 1 and 8 shards, all on this device (one host thread
    per shard inside the library, host merge with the reference comparator); same hit list as the timed region"""
    import hashlib
    import swipe_amd
    out = []
    for shards in (1, 8):
        t0 = time.time()
        g = swipe_amd.Group.from_arrays(res, off, devices=(device,) * shards)
        t_load = time.time() - t0
        g.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
        g.search_topk(q, keep=KEEP, minscore=minscore, maxscore=maxscore)
        t1 = time.perf_counter()
        for _ in range(steps):
            hits, tot, obv, c = g.search_topk(q, keep=KEEP, minscore=minscore, maxscore=maxscore)
        el = (time.perf_counter() - t1) / steps
        g.close()
        sha = hashlib.sha1(np.ascontiguousarray(np.array(hits, dtype=np.int64).reshape(-1, 2)).tobytes()).hexdigest()
        if sha != want_sha1:
            raise SystemExit(f"bench: swa_group with {shards} shard(s) disagrees with the timed region's hit list")
        out.append({"metric": f"GCUPS, swa_group (in-process C++ multi-device layer), {shards} shard(s) on device {device}",
                    "value": round(int(off[-1] - off[0]) * len(q) / el / 1e9, 1), "unit": "GCUPS", "steps": steps,
                    "ms_per_step": round(el * 1e3, 3), "kernel_ms_slowest_shard": round(c["kernel_ms"], 3),
                    "load_s": round(t_load, 2), "hits_sha1": sha, "hits_identical": True})
    return out


def nucleotide_cold_open(res, off, device, q, qm, minscore, nslice=2_000_000):
    """disk -> HBM for nucleotide volumes (round 5: the pipelined open takes the .nsq as it lies - 2 bits per base, ambiguity
    tables - and unpacks on the device): the first `nslice` sequences of the section's database written as a BLAST v4 volume,
    swa_db_open warm, then swa_db_open_async + the first both-strand top-K search FOLLOWING the loader; its hit list must be the
    resident handle's.  A slice, so that the default run stays inside its minutes; rates are per GB of .nsq."""
    import swipe_amd
    n = min(nslice, len(off) - 1)
    d = tempfile.mkdtemp(prefix="swa_ntcold_")
    try:
        base = os.path.join(d, "nt")
        t0 = time.time()
        swipe_amd.write_blastdb(base, res, off[: n + 1], symtype=0, first_id=0)
        t_write = time.time() - t0
        gb = os.path.getsize(base + ".nsq") / 1e9
        M = swipe_amd.matrix_nucleotide(1, -3)
        t0 = time.time()
        db = swipe_amd.Database.open(base, symtype=0, device=device)
        warm = time.time() - t0
        db.set_scoring(M, 5, 2)
        want = db.search2_topk(q, qm, keep=KEEP, minscore=minscore)[:3]
        db.close()
        t0 = time.time()
        db = swipe_amd.Database.open(base, symtype=0, device=device, wait=False)
        t_ret = time.time() - t0
        db.set_scoring(M, 5, 2)
        hits, tot, obv, c = db.search2_topk(q, qm, keep=KEEP, minscore=minscore)
        first_hits = time.time() - t0
        db.wait()
        resident = time.time() - t0
        db.close()
        if (hits, tot, obv) != want:
            raise SystemExit("bench (nucleotide): the search that followed the loader disagrees with the resident shard's hit list")
        return {"sequences": int(n), "bases": int(off[n] - off[0]), "nsq_gb": round(gb, 3), "write_s": round(t_write, 2),
                "open_s": round(warm, 3), "warm_gb_per_s": round(gb / warm, 2), "async_open_returns_s": round(t_ret, 3),
                "first_hits_s": round(first_hits, 3), "resident_s": round(resident, 3), "first_search_parts": int(c["loading_parts"]),
                "what": "swa_db_open of a .nsq volume (2 bits per base) from the box's local disk, warm page cache: lengths out of the "
                        "index + one byte per entry on 16 threads | reader threads -> page-locked ring -> copy engine -> swa_unpack_nt per "
                        "chunk -> 4-bit parts; first_hits_s = swa_db_open_async + swa_set_scoring + the first both-strand top-250 search "
                        "following the loader part by part (hit list identical to the resident handle's)"}
    finally:
        subprocess.run(["rm", "-rf", d])


def nucleotide_section(a, rank, local, world, nseq, steps, want_cpu):
    """BASELINE.json configs[3]: 1 kb DNA query vs a synthetic nucleotide db, +1/-3, gap 5+2, both strands in one pass of
    the two-query kernel (plus strand | reverse complement in the two halves of the packed lanes).  Single shard."""
    import torch
    import swipe_amd
    from swipe_amd import blastdb, synth
    rtab = synth.residue_table_nucleotide()
    q = synth._random_residues(99, 1, 1000, rtab)
    qm = blastdb.revcomp_nt16(q)
    res, off = swipe_amd.synth_db(3, nseq, protein=False, threads=os.cpu_count() or 1)
    t0 = time.time()
    db = swipe_amd.Database.from_arrays(res, off, symtype=0, device=local)
    t_load = time.time() - t0
    db.set_scoring(swipe_amd.matrix_nucleotide(1, -3), 5, 2)
    st = swipe_amd.stats_init(symtype=0, match=1, mismatch=-3, gapopen=5, gapextend=2, qlen=len(q),
                              db_seqcount=nseq, db_symcount=int(off[-1]))
    for _ in range(max(1, a.warmup)):
        db.search2_topk(q, qm, keep=KEEP, minscore=st.scorethreshold)
    torch.cuda.synchronize()
    kms, per = [], []
    t0 = time.perf_counter()
    for _ in range(steps):
        t1 = time.perf_counter()
        hits, tot, obv, c = db.search2_topk(q, qm, keep=KEEP, minscore=st.scorethreshold)
        per.append(time.perf_counter() - t1)
        kms.append(c["kernel_ms"])
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    nsym = int(off[-1])
    cells = 2 * nsym * len(q)
    k_ms = float(np.mean(kms))
    info = db.info()
    traffic, tsrc = committed_traffic("nucleotide", nseq)
    roof, valu = roofline_blocks(nsym, nseq, cells, k_ms, c["narrow_shifted"], c["narrow_rows"], traffic=traffic, traffic_source=tsrc,
                                 qlen=len(q))
    roof["algorithmic_bytes_note"] = ("1 B per base + 12 B per sequence, read ONCE for both strands (the reference makes one pass per "
                                      "strand, swipe.cc:1403); at the .nsq format's 2 bits per base it would be %d" % (nsym // 4 + 12 * nseq))
    out = {"metric": "GCUPS, 1 kb DNA query vs synthetic nt db, both strands (BASELINE.json configs[3])",
           "value": round(cells * steps / el / 1e9, 1), "unit": "GCUPS", "n_gpus": 1, "steps": steps,
           "ms_per_step": round(el / steps * 1e3, 3), "ms_median": round(float(np.median(per)) * 1e3, 3),
           "overhead_ms": round(el / steps * 1e3 - k_ms, 3), "dtype": "f16x2 (plus strand | minus strand)", "data": "synthetic",
           "config": {"workload": f"1000-nt query, both strands, vs {nseq} nt sequences ({nsym} bases), +1/-3, gap 5+2, "
                                  f"top-{KEEP} hits by E<=10 (score >= {st.scorethreshold})"},
           "roofline": roof, "valu_roofline": valu, "totalhits": int(tot),
           "hbm_bytes_per_base": round(info["hbm_bytes"] / max(1, nsym), 3), "setup_s": {"load_format": round(t_load, 2)}}
    if not a.no_verify:
        # checker leg: all scores of both strands (swa_search2), the merged hit list recomputed on the host over every
        # sequence, the hits + a seeded sample of sequences recomputed by the oracle's scalar recurrence
        import oracle
        s1, s2, _ = db.search2(q, qm)
        minscore = st.scorethreshold
        cand = sorted([(int(v), int(i), 0) for i, v in zip(np.flatnonzero(s1 >= minscore), s1[s1 >= minscore])] +
                      [(int(v), int(i), 1) for i, v in zip(np.flatnonzero(s2 >= minscore), s2[s2 >= minscore])],
                      key=lambda t: (-t[0], -t[1], t[2]))[:KEEP]
        bad = int([(i, v, w) for v, i, w in cand] != [tuple(h) for h in hits])
        bad += int(int((s1 >= minscore).sum()) + int((s2 >= minscore).sum()) != tot)
        rng = np.random.default_rng(20260929)
        pick = np.unique(np.concatenate([rng.integers(0, nseq, size=min(max(1, a.verify_sample // 5), nseq)),
                                         np.array([h[0] for h in hits], dtype=np.int64)]))
        lens = off[pick + 1] - off[pick]
        o2 = np.zeros(len(pick) + 1, dtype=np.int64)
        np.cumsum(lens, out=o2[1:])
        r2 = np.empty(int(o2[-1]), dtype=np.uint8)
        for k, i in enumerate(pick):
            r2[o2[k]:o2[k + 1]] = res[off[i]:off[i + 1]]
        Mo = oracle.matrix_nucleotide(1, -3)
        thr = os.cpu_count() or 1
        bad += int((oracle.search_all63(r2, o2, q, Mo, 7, 2, threads=thr) != s1[pick]).sum())
        bad += int((oracle.search_all63(r2, o2, qm, Mo, 7, 2, threads=thr) != s2[pick]).sum())
        if bad:
            raise SystemExit(f"bench (nucleotide): {bad} mismatches against the oracle")
        out["verified_vs_oracle"] = int(len(pick))
    if not a.no_cold:
        try:
            out["cold_open"] = nucleotide_cold_open(res, off, local, q, qm, st.scorethreshold)
        except Exception as e:
            out["cold_open"] = {"open_s": None, "what": f"failed: {e}"}
    if want_cpu:
        try:
            qtext = "".join("-ACMGRSVTWYHKDBN"[int(x)] for x in q)
            out["cpu_baseline"] = cpu_baseline(res, off, qtext, os.cpu_count() or 1, protein=False, sample_seqs=300_000)
        except Exception as e:
            out["cpu_baseline"] = {"value": None, "unit": "GCUPS", "kind": "reference", "sample": f"failed: {e}"}
    db.close()
    return out


def protein100m_section(a, local, nseq=100_000_000, steps=3):
    """BASELINE.json configs[4]'s database - 100 M synthetic proteins, 32.4 G residues - resident on ONE MI355X (about 70 GB
    of its 288 GB): the same top-250 search as the headline, timed the same way, verified the same way (all scores from
    the exact pass recounted on the host over every sequence; hits + a seeded sample recomputed by the oracle).  The
    8-GPU form of this config is `--workload protein100M --gpus 8` (12.5 M sequences per rank)."""
    import torch
    import swipe_amd
    from swipe_amd import blastdb, synth
    q = blastdb.encode_protein(synth.QUERY_P07327)
    t0 = time.time()
    res, off = swipe_amd.synth_db(1, nseq, query=q, threads=os.cpu_count() or 1)
    t_gen = time.time() - t0
    nsym = int(off[-1])
    t0 = time.time()
    db = swipe_amd.Database.from_arrays(res, off, device=local)
    t_load = time.time() - t0
    db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
    st = swipe_amd.stats_init(qlen=len(q), db_seqcount=nseq, db_symcount=nsym)
    minscore, maxscore = st.scorethreshold, st.upperscorethreshold
    db.search_topk_array(q, keep=KEEP, minscore=minscore, maxscore=maxscore)
    torch.cuda.synchronize()
    kms, per = [], []
    t0 = time.perf_counter()
    for _ in range(steps):
        t1 = time.perf_counter()
        hits, tot, obv, c = db.search_topk_array(q, keep=KEEP, minscore=minscore, maxscore=maxscore)
        per.append(time.perf_counter() - t1)
        kms.append(c["kernel_ms"])
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    k_ms = float(np.mean(kms))
    cells = nsym * len(q)
    traffic, tsrc = committed_traffic("protein", nseq)
    roof, valu = roofline_blocks(nsym, nseq, cells, k_ms, c["narrow_shifted"], c["narrow_rows"], traffic=traffic, traffic_source=tsrc)
    info = db.info()
    out = {"metric": "GCUPS, 375-aa query vs 100M-seq protein db (BASELINE.json configs[4]'s database) on ONE MI355X",
           "value": round(cells * steps / el / 1e9, 1), "unit": "GCUPS", "n_gpus": 1, "steps": steps, "warmup": 1,
           "ms_per_step": round(el / steps * 1e3, 3), "ms_median": round(float(np.median(per)) * 1e3, 3),
           "overhead_ms": round(el / steps * 1e3 - k_ms, 3), "dtype": "f16x2 (exact integers; re-queue to i32/i64)", "data": "synthetic",
           "config": {"workload": f"375-aa query (P07327) vs {nseq} synthetic protein sequences ({nsym} residues) resident on one GPU, "
                                  f"BLOSUM62, gap 11+1, top-{KEEP} hits by E<=10 (score >= {minscore})"},
           "roofline": roof, "valu_roofline": valu,
           "search": {"totalhits": int(tot), "top_hit": [int(x) for x in hits[0]] if len(hits) else None,
                      "requeued_32bit": int(c["wide"]), "requeued_64bit": int(c["full"])},
           "hbm_gb": round(info["hbm_bytes"] / 1e9, 1), "setup_s": {"generate": round(t_gen, 2), "load_format": round(t_load, 2)}}
    if not a.no_verify:
        nver, bad, tot_all = verify_against_oracle(db, res, off, 0, q, "BLOSUM62", 12, 1, [tuple(h) for h in hits.tolist()], tot,
                                                   minscore, maxscore, max(1, a.verify_sample // 4), os.cpu_count() or 1)
        if bad or tot_all != tot:
            raise SystemExit(f"bench (100 M proteins): {bad} scores differ from the oracle (totalhits {tot} vs {tot_all} recounted)")
        out["verified_vs_oracle"] = int(nver)
    db.close()
    return out


def predict_scaling(a, local):
    """What the 8-GPU strong-scaling run of the metric will show, measured on ONE GPU: for N = 1, 2, 4, 8 every shard
    parallel.shard_bounds gives rank r of N is generated, loaded and searched on its own with the step of the bench
    (top-250, thresholds of the WHOLE database), and the step of an N-rank run is predicted as the slowest shard's step plus
    the gather - one all_gather_into_tensor of 250 x 2 + 3 int64 per rank and swa_hits_merge of N lists, whose latency is
    measured here over RCCL with one rank (a lower bound: xGMI hops of a 4 KB message add microseconds, not
    milliseconds).  No collective sits on the data path, so nothing else changes with N."""
    import torch
    import torch.distributed as dist
    import swipe_amd
    from swipe_amd import blastdb, parallel, synth
    nseq_total = a.nseq or 10_000_000
    q = blastdb.encode_protein(synth.QUERY_P07327)
    cores = os.cpu_count() or 1
    goff = swipe_amd.synth_offsets(1, nseq_total, query=q, threads=cores)
    tot_sym = int(goff[-1])
    st = swipe_amd.stats_init(qlen=len(q), db_seqcount=nseq_total, db_symcount=tot_sym)
    minscore, maxscore = st.scorethreshold, st.upperscorethreshold
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29591")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local))
    dev = torch.device("cuda", local)
    rows, merged_ref = [], None
    for world in (1, 2, 4, 8):
        bounds = parallel.shard_bounds(goff, world)
        shard_ms, lists, tots = [], [], 0
        for lo, hi in bounds:
            res, off = swipe_amd.synth_db(1, hi - lo, first=lo, query=q, threads=cores)
            db = swipe_amd.Database.from_arrays(res, off, device=local, first_seqno=lo, total_seqcount=nseq_total, total_symcount=tot_sym)
            db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
            for _ in range(4):
                db.search_topk_array(q, keep=KEEP, minscore=minscore, maxscore=maxscore)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                hits, tot, obv, c = db.search_topk_array(q, keep=KEEP, minscore=minscore, maxscore=maxscore)
            torch.cuda.synchronize()
            shard_ms.append((time.perf_counter() - t0) / a.steps * 1e3)
            lists.append(hits.copy())
            tots += tot
            db.close()
            del res, off
        # the gather + merge of one step at this N: measured over RCCL (world 1) on a buffer of the N-rank size is not
        # possible with one rank, so the per-rank all_gather is timed as it is and the merge over N real lists on the host
        t0 = time.perf_counter()
        for _ in range(50):
            parallel.gather_topk_array(lists[0], KEEP, 0, 0, device=dev)
        gather_ms = (time.perf_counter() - t0) / 50 * 1e3
        stack = np.zeros((world, KEEP, 2), dtype=np.int64)
        counts = np.zeros(world, dtype=np.int64)
        for r, h in enumerate(lists):
            stack[r, : len(h)] = h
            counts[r] = len(h)
        t0 = time.perf_counter()
        for _ in range(50):
            merged = swipe_amd.merge_hit_arrays(stack, counts, KEEP)
        merge_ms = (time.perf_counter() - t0) / 50 * 1e3
        if merged_ref is None:
            merged_ref, tot_ref = merged.copy(), tots
        elif not (np.array_equal(merged, merged_ref) and tots == tot_ref):
            raise SystemExit(f"predict-scaling: the merged list of {world} shards differs from the single shard's")
        step = max(shard_ms) + gather_ms + (merge_ms if world > 1 else 0.0)
        rows.append({"n_gpus": world, "shard_ms": [round(x, 3) for x in shard_ms], "slowest_shard_ms": round(max(shard_ms), 3),
                     "gather_ms_rccl_world1": round(gather_ms, 3), "merge_ms": round(merge_ms, 3), "predicted_ms_per_step": round(step, 3),
                     "predicted_gcups": round(tot_sym * len(q) / (step * 1e-3) / 1e9, 1)})
    base = rows[0]["predicted_ms_per_step"]
    for r in rows:
        r["predicted_efficiency"] = round(base / (r["n_gpus"] * r["predicted_ms_per_step"]), 4)
    dist.destroy_process_group()
    return {"what": "predicted strong scaling of `bench.py --gpus N` from one GPU: slowest residue-balanced shard + measured gather + merge",
            "database": {"sequences": nseq_total, "residues": tot_sym, "query_len": len(q), "threshold": int(minscore)},
            "steps_per_shard": a.steps, "merged_lists_identical": True, "rows": rows,
            "not_modelled": "8 processes sharing the host's PCIe / page-locked memory during the 4 KB D2H copy of each step; "
                            "RCCL all_gather across xGMI instead of inside one rank (a 4 KB message: tens of microseconds)"}


def relaunch_command(gpus, env, argv):
    """The command `python bench.py --gpus N` turns itself into when N > 1 and no launcher set WORLD_SIZE: the driver's own
    line (torch.distributed.run, one node, N ranks, rendezvous on 127.0.0.1 - the container's hostname may not resolve) on a
    free port.  None when there is nothing to do."""
    if gpus <= 1 or "WORLD_SIZE" in env:
        return None
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=7)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--nseq", type=int, default=0, help="sequences of the database (default: 10 M; protein100M: 100 M)")
    ap.add_argument("--weak", action="store_true", help="every rank its own --nseq sequences instead of a shard of one database")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the exact-first-pass, nucleotide and cold-open sections")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-cold", action="store_true", help="skip the cold-open (disk -> HBM) section")
    ap.add_argument("--traffic-only", action="store_true", help="the headline step + roofline.traffic measured in the run, nothing else (tests)")
    ap.add_argument("--verify-sample", type=int, default=10_000)
    ap.add_argument("--quick", action="store_true", help="secondary sections at reduced size: nucleotide 10 M sequences, no 100 M-protein section")
    ap.add_argument("--secondary-nt-nseq", type=int, default=0, help="sequences of the nucleotide secondary section (default 50 M; --quick 10 M)")
    ap.add_argument("--secondary-protein-nseq", type=int, default=0, help="sequences of the big-protein secondary section (default 100 M)")
    ap.add_argument("--predict-scaling", action="store_true", help="one GPU: time every shard of N = 1, 2, 4, 8 and predict the scaling curve")
    ap.add_argument("--traffic-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--device", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--no-live-traffic", action="store_true", help="roofline.traffic from the committed PMC passes instead of a live rocprofv3 pass")
    ap.add_argument("--workload", choices=["protein", "protein100M", "nucleotide"], default="protein",
                    help="protein = BASELINE.json configs[1] (the headline); protein100M = configs[4]; nucleotide = configs[3]")
    a = ap.parse_args()
    if a.traffic_child:
        traffic_child(a.nseq or 10_000_000, a.device)
        return
    if a.traffic_only:
        a.no_secondary = a.no_cpu_baseline = a.no_verify = a.no_cold = True

    # `python bench.py --gpus N` without a launcher: become the launcher the driver uses (one rank per GPU over RCCL).  Under
    # torch.distributed.run WORLD_SIZE is set and is what counts; --gpus is then only the caller's statement of it.
    cmd = relaunch_command(a.gpus, os.environ, sys.argv[1:])
    if cmd:
        os.execv(cmd[0], cmd)
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != a.gpus:
        print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={os.environ['WORLD_SIZE']}: running {os.environ['WORLD_SIZE']} ranks", file=sys.stderr)

    rank = int(os.environ.get("RANK", "0"))
    # SWA_BENCH_DEVICE / SWA_BENCH_BACKEND=gloo: tests run the N-rank code path on the ONE GPU a test box has (all ranks on
    # that device, collectives over gloo on host tensors - RCCL refuses two ranks on one GPU); the driver never sets them
    local = int(os.environ.get("SWA_BENCH_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    backend = os.environ.get("SWA_BENCH_BACKEND", "nccl")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    import swipe_amd
    from swipe_amd import blastdb, parallel, synth
    ndev = swipe_amd._lib.load().swa_device_count() if torch.cuda.is_available() else 0
    if ndev == 1 and "SWA_BENCH_DEVICE" not in os.environ:
        local = 0           # a launcher that narrows each rank's view to its own GPU (HIP_VISIBLE_DEVICES per rank)
    if ndev <= local:
        raise SystemExit("bench.py needs a HIP device per rank (swipe_amd has no CPU path)")
    torch.cuda.set_device(local)
    dist = None
    use_dist = world > 1 or os.environ.get("SWA_BENCH_FORCE_DIST") == "1"     # the latter: exercise RCCL with one rank
    if use_dist:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    cdev = "cuda" if backend == "nccl" else "cpu"          # where the collectives' tensors live
    cores = os.cpu_count() or 1

    if a.predict_scaling:
        if world != 1:
            raise SystemExit("--predict-scaling runs on one GPU")
        r = predict_scaling(a, local)
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)                     # RCCL's banner sits in a C buffer: out first, the JSON line last
        print(json.dumps(r), flush=True)
        return

    if a.workload == "nucleotide":
        out = nucleotide_section(a, rank, local, world, a.nseq or (10_000_000 if a.quick else 50_000_000), a.steps, not a.no_cpu_baseline and world == 1)
        if rank == 0:
            out.update({"warmup": a.warmup, "higher_is_better": True, "scaling": "weak", "vs_baseline": None})
            print(json.dumps(out), flush=True)
        return

    nseq_total = a.nseq or (100_000_000 if a.workload == "protein100M" else 10_000_000)
    q = blastdb.encode_protein(synth.QUERY_P07327)
    gen_threads = max(1, cores // max(1, world))
    t0 = time.time()
    if a.weak:
        lo, n_local = rank * nseq_total, nseq_total
        db_seqs = world * nseq_total
    else:
        # ONE database: every rank derives the same length table, takes its residue-balanced slice of sequence numbers
        # and generates only that slice (the generator is counter-based, keyed by the global sequence number)
        goff = swipe_amd.synth_offsets(1, nseq_total, query=q, threads=gen_threads)
        lo, hi = parallel.shard_bounds(goff, world)[rank]
        n_local = hi - lo
        db_seqs = nseq_total
        tot_sym_known = int(goff[-1])
        del goff
    res, off = swipe_amd.synth_db(1, n_local, first=lo, query=q, threads=gen_threads)
    t_gen = time.time() - t0
    nsym = int(off[-1])
    tot_sym = nsym
    if use_dist:
        t = torch.tensor([nsym], dtype=torch.int64, device=cdev)
        dist.all_reduce(t)
        tot_sym = int(t.item())
    if not a.weak and tot_sym != tot_sym_known:
        raise SystemExit("bench: the shards do not add up to the database")
    t0 = time.time()
    db = swipe_amd.Database.from_arrays(res, off, device=local, first_seqno=lo, total_seqcount=db_seqs, total_symcount=tot_sym)
    t_load = time.time() - t0
    db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
    st = swipe_amd.stats_init(qlen=len(q), db_seqcount=db_seqs, db_symcount=tot_sym)
    dev = torch.device("cuda", local) if use_dist and backend == "nccl" else None
    minscore, maxscore = st.scorethreshold, st.upperscorethreshold

    def step():
        hits, tot, obv, c = db.search_topk_array(q, keep=KEEP, minscore=minscore, maxscore=maxscore)
        if use_dist:
            hits, tot, obv = parallel.gather_topk_array(hits, KEEP, tot, obv, device=dev)
        return hits, tot, c

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def all_max(x):
        if not use_dist:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def timed(steps):
        kms, per = [], []
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            t1 = time.perf_counter()
            hits, tot, c = step()
            per.append(time.perf_counter() - t1)
            kms.append(c["kernel_ms"])
        fence()
        el = time.perf_counter() - t0
        return all_max(el), hits, tot, c, float(np.mean(kms)), float(np.median(per))

    for _ in range(a.warmup):
        step()
    elapsed, hits, tot, c, k_ms, med = timed(a.steps)

    # The same step with the exact first pass (every score of the shard exact on the device, what swa_search returns),
    # over the same number of steps, reported beside the headline; the two hit lists must be identical.  The headline step
    # may run the bound build, which computes exact scores only for sequences that can reach the E <= 10 threshold -
    # hits_enter drops every other score unseen (hits.cc:174-184).
    exact = None
    want_exact = int(c["narrow_shifted"] in (8, 9) and not a.no_secondary)
    if use_dist:                                         # every rank takes the same branch (the steps hold collectives)
        t = torch.tensor([want_exact], dtype=torch.int64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        want_exact = int(t.item())
    if want_exact:
        db.set_option("bound", 0)
        step()
        el_x, hits_x, tot_x, c_x, k_x, med_x = timed(a.steps)
        db.set_option("bound", None)
        same = bool(np.array_equal(hits_x, hits) and tot_x == tot)
        if not same:
            raise SystemExit("bench: the bound build and the exact first pass disagree on the hit list")
        tr_x, ts_x = committed_traffic("exact", n_local)
        r_x, v_x = roofline_blocks(nsym, n_local, nsym * len(q), k_x, c_x["narrow_shifted"], c_x["narrow_rows"], traffic=tr_x, traffic_source=ts_x)
        exact = {"value": round(tot_sym * len(q) * a.steps / el_x / 1e9, 1), "unit": "GCUPS", "steps": a.steps,
                 "ms_per_step": round(el_x / a.steps * 1e3, 3), "ms_median": round(med_x * 1e3, 3),
                 "overhead_ms": round(el_x / a.steps * 1e3 - k_x, 3), "hits_identical": same, "roofline": r_x, "valu_roofline": v_x,
                 "note": "the same step with swa_set_option(bound, 0): ALL scores of the shard exact on the device (what "
                         "swa_search returns, 7.5 instructions per cell pair); the headline step recomputes exactly only "
                         "what can reach the threshold"}

    # Round 6 changed the bound build (sequences back to back, twin profile: DESIGN 4.2) without hardware to time it on.  The same
    # step with both switched off IS the round-3 kernel: reported beside the headline so that the first run on an MI355X is its
    # own A/B (single GPU only; same hit list or the bench fails).
    round3 = None
    if want_exact and not use_dist:
        try:
            db.set_option("concat", 1)
            db.set_option("twin", 0)
            step()
            el_3, hits_3, tot_3, c_3, k_3, med_3 = timed(a.steps)
        finally:
            db.set_option("concat", None)
            db.set_option("twin", None)
        if not (np.array_equal(hits_3, hits) and tot_3 == tot):
            raise SystemExit("bench: the round-3 form of the bound build disagrees with the headline's hit list")
        round3 = {"value": round(tot_sym * len(q) * a.steps / el_3 / 1e9, 1), "unit": "GCUPS", "steps": a.steps,
                  "ms_per_step": round(el_3 / a.steps * 1e3, 3), "kernel_ms": round(k_3, 3), "headline_kernel_ms": round(k_ms, 3),
                  "requeued_32bit": int(c_3["wide"]), "headline_requeued_32bit": int(c["wide"]), "hits_identical": True,
                  "note": "the same step with swa_set_option(concat, 1) and (twin, 0): every set of batches drained and reset on its "
                          "own, one copy of the profile, blocks of 4 waves - the kernel that measured 11 614 GCUPS in round 3"}

    verified = None
    if not a.no_verify:
        nver, bad, tot_local = verify_against_oracle(db, res, off, lo, q, "BLOSUM62", 12, 1, [tuple(h) for h in hits.tolist()],
                                                     tot, minscore, maxscore, max(1, a.verify_sample // world), gen_threads)
        v = torch.tensor([nver, bad, tot_local], dtype=torch.int64, device=cdev)
        if use_dist:
            dist.all_reduce(v)
        nver, bad, tot_all = (int(x) for x in v.tolist())
        if bad or tot_all != tot:
            raise SystemExit(f"bench: {bad} scores differ from the oracle (totalhits {tot} vs {tot_all} recounted)")
        verified = nver

    line = None
    if rank == 0:
        cells_per_step = tot_sym * len(q)
        value = cells_per_step * a.steps / elapsed / 1e9
        form = c["narrow_shifted"]
        traffic, tsrc = committed_traffic("protein", n_local)
        if world == 1 and a.workload == "protein" and (a.traffic_only or not a.no_secondary) and not a.no_live_traffic:
            try:
                live, why = live_traffic(n_local, local)         # after the timed region, in a child process under rocprofv3
            except BaseException as e:                           # whatever it is, the line and the sections after it survive
                if isinstance(e, KeyboardInterrupt):
                    raise
                live, why = None, "live traffic raised %s: %s" % (type(e).__name__, e)
            if live:
                traffic, tsrc = live, why
            elif tsrc:
                tsrc += " (live pass not available: %s)" % why
            else:
                tsrc = "not measured: %s" % why
        roof, valu = roofline_blocks(nsym, n_local, nsym * len(q), k_ms, form, c["narrow_rows"], traffic=traffic, traffic_source=tsrc)
        what = "top-%d search, bound first pass" % KEEP if form in (8, 9, 10) else "top-%d search, exact first pass" % KEEP
        out = {
            "metric": "GCUPS, 375-aa query vs 10M-seq protein db at 1/2/4/8 GPUs; bit-exact scores",
            "value": round(value, 1), "unit": "GCUPS", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 3), "ms_median": round(med * 1e3, 3),
            "overhead_ms": round(elapsed / a.steps * 1e3 - k_ms, 3),
            "higher_is_better": True, "scaling": "weak" if a.weak else "strong",
            "vs_baseline": None, "dtype": "f16x2 (exact integers; re-queue to i32/i64)", "data": "synthetic",
            "value_is": what + ": hit list, totalhits and every listed score bit-exact (verified_vs_oracle); the all-scores-"
                               "exact rate of swa_search is exact_first_pass.value",
            "collectives": (backend if use_dist else None),
            "config": {"workload": f"375-aa query (P07327) vs ONE database of {db_seqs} synthetic protein sequences "
                                   f"({tot_sym} residues), BLOSUM62, gap 11+1, top-{KEEP} hits by E<=10 (score >= {minscore}); "
                                   f"value = {what} (hit list, totalhits, every listed score bit-exact); the rate with EVERY score of "
                                   f"the database exact on the device is exact_first_pass.value",
                       "sequences_total": db_seqs, "residues_total": tot_sym, "query_len": len(q),
                       "sequences_rank0": n_local, "residues_rank0": nsym,
                       "sharding": (f"{world} read-only shards of the one database by residue count (parallel.shard_bounds); "
                                    f"one all_gather of {KEEP}x2+3 int64 per step" if not a.weak else
                                    f"{world} shards of {nseq_total} sequences each (weak)") if world > 1 else "single shard"},
            "roofline": roof, "valu_roofline": valu,
            "search": {"totalhits": int(tot), "top_hit": [int(x) for x in hits[0]] if len(hits) else None,
                       "requeued_32bit": int(c["wide"]), "requeued_64bit": int(c["full"])},
            "setup_s": {"generate": round(t_gen, 2), "load_format": round(t_load, 2)},
            "hits_sha1": __import__("hashlib").sha1(np.ascontiguousarray(hits, dtype=np.int64).tobytes()).hexdigest(),
        }
        if verified is not None:
            out["verified_vs_oracle"] = verified
            out["verified_how"] = ("after the timed region: all scores of every shard from the exact pass; hit list and totalhits "
                                   "recomputed on the host over every sequence; the hits + a seeded random sample recomputed by "
                                   "oracle/ (scalar 63-bit recurrence, search63.cc:28-89) - 0 mismatches or the bench fails")
        if exact:
            out["exact_first_pass"] = exact
        if round3:
            out["round3_first_pass"] = round3
        line = out
    pair, group = None, []
    if world == 1 and rank == 0 and not a.no_secondary and a.workload == "protein":
        # a query FILE (the reference's unit of work, swipe.cc:2561-2575): two different 375-aa queries per pass
        # (swa_search_pair_topk), each with its own E <= 10 window; both hit lists must equal the one-per-pass searches
        try:
            q2 = synth._random_residues(4242, 1, len(q), synth.residue_table_protein())
            ones = [db.search_topk_array(x, keep=KEEP, minscore=minscore, maxscore=maxscore) for x in (q, q2)]
            db.search_pair_topk(q, q2, keep=KEEP, minscore=(minscore, minscore), maxscore=(maxscore, maxscore))
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(3):
                r = db.search_pair_topk(q, q2, keep=KEEP, minscore=(minscore, minscore), maxscore=(maxscore, maxscore))
            el_p = (time.perf_counter() - t1) / 3
            same = all([tuple(h) for h in ones[k][0].tolist()] == r[k][0] and ones[k][1] == r[k][1] for k in (0, 1))
            if not same:
                raise SystemExit("bench: the paired search and the one-per-pass searches disagree")
            pair = {"metric": "GCUPS aggregate, two different 375-aa queries per pass over the same database (swa_search_pair_topk)",
                    "value": round(2 * nsym * len(q) / el_p / 1e9, 1), "unit": "GCUPS", "steps": 3,
                    "ms_per_pair": round(el_p * 1e3, 3), "kernel_ms": round(r[2]["kernel_ms"], 3), "hits_identical": True,
                    "kernel": KERNEL.get(r[2]["narrow_shifted"], "%d") % r[2]["narrow_rows"]}
        except SystemExit:
            raise
        except Exception as e:
            pair = {"metric": "pair section", "value": None, "error": str(e)}
    if world == 1 and rank == 0 and not a.no_secondary and not a.no_cold and a.workload == "protein":
        try:
            line["cold_open"] = cold_open(res, off, local, q, minscore, maxscore, [tuple(int(x) for x in h) for h in hits.tolist()])
        except Exception as e:
            line["cold_open"] = {"open_s": None, "what": f"failed: {e}"}
        try:
            group = group_section(res, off, local, q, minscore, maxscore, line["hits_sha1"])
        except Exception as e:
            group = [{"metric": "swa_group section", "value": None, "error": str(e)}]
    if world == 1 and rank == 0 and not a.no_cpu_baseline:
        try:
            line["cpu_baseline"] = cpu_baseline(res, off, synth.QUERY_P07327, cores)
        except Exception as e:   # a missing baseline must not lose the measurement
            line["cpu_baseline"] = {"value": None, "unit": "GCUPS", "cores": cores, "kind": "reference", "sample": f"failed: {e}"}
    db.close()
    del res, off
    if world == 1 and rank == 0 and not a.no_secondary and a.workload == "protein":
        try:
            import gc
            gc.collect()
            line["secondary"] = [nucleotide_section(a, rank, local, world, a.secondary_nt_nseq or (10_000_000 if a.quick else 50_000_000), 3,
                                                    not a.no_cpu_baseline)]
        except Exception as e:
            line["secondary"] = [{"metric": "nucleotide section", "value": None, "error": str(e)}]
        if not a.quick:
            try:
                gc.collect()
                line["secondary"].append(protein100m_section(a, local, a.secondary_protein_nseq or 100_000_000))
            except Exception as e:
                line["secondary"].append({"metric": "100 M-protein section", "value": None, "error": str(e)})
        if pair:
            line["secondary"].append(pair)
        line["secondary"].extend(group)
    if use_dist:
        dist.destroy_process_group()
    if rank == 0:
        # RCCL writes its version banner to C stdout, block-buffered when piped, i.e. at exit - AFTER anything Python
        # printed.  Drain the C buffers first so that the JSON line is the last line on stdout.
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
