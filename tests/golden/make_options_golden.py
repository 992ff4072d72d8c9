#!/usr/bin/env python3
"""tests/golden/options.json: the reference CLI under the options that move hits_init's thresholds and search_chunk's
strand loops (swipe.cc:1088-1161, hits.cc:283-511) - score window (-c / -u), E-value window (-e / -k), effective database
size (-z), list lengths (-v / -b), query strands (-S), nucleotide rewards, other matrices and gap penalties (with and
without Karlin-Altschul parameters), query strands of a translated search.  Build container only (needs oracle/_ref/swipe)."""
import json, os, subprocess, sys, tempfile
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))
import cases
from swipe_amd import blastdb

REF = os.path.join(ROOT, "oracle", "_ref", "swipe")

# (case, extra options); every run is taken with -m 8 (alignments of the whole list) and -m 7 -b 0 (hit list only)
VARIANTS = {
    "p1k": [["-c", "100"], ["-u", "300"], ["-c", "60", "-u", "200"], ["-k", "1e-20"], ["-e", "1e-5"], ["-e", "1e-5", "-k", "1e-50"],
            ["-z", "1000000"], ["-z", "100000000000", "-e", "1000"], ["-v", "5", "-b", "3"], ["-v", "3", "-b", "7"],
            ["-M", "BLOSUM50", "-G", "13", "-E", "2"], ["-M", "BLOSUM62", "-G", "9", "-E", "2"], ["-M", "PAM250", "-G", "14", "-E", "2"],
            ["-M", "BLOSUM62", "-G", "3", "-E", "3", "-c", "40"],          # no K-A parameters for this gap system: raw scores
            ["-e", "1e-300"]],
    "nt": [["-S", "1"], ["-S", "2"], ["-S", "plus"], ["-S", "minus"], ["-S", "both"], ["-r", "2", "-q", "-5"], ["-r", "1", "-q", "-1", "-G", "3", "-E", "1"],
           ["-c", "30"], ["-u", "40"], ["-e", "1e-3"], ["-z", "5000000"]],
    "blastx": [["-S", "1"], ["-S", "2"], ["-c", "50"]],
    "tblastx": [["-S", "1"], ["-e", "1e-3"]],
}


def base_args(case, base, qf):
    a = ["-d", base, "-i", qf, "-p", str(case.sym)]
    if case.sym >= 2:
        a += ["-Q", str(case.query_gencode), "-D", str(case.db_gencode)]
    return a


def main():
    out = {}
    for name, variants in VARIANTS.items():
        case = cases.get(name)
        d = tempfile.mkdtemp(prefix="golden_opt_")
        base = os.path.join(d, name)
        blastdb.write_db(base, case.seqs, protein=case.protein, volumes=case.volumes)
        alpha = blastdb.NCBI4NA if case.query_is_nt else blastdb.NCBISTDAA
        qf = os.path.join(d, "q.fa")
        open(qf, "w").write(">query test\n" + "".join(alpha[c] for c in case.query) + "\n")
        rows = []
        for extra in variants:
            rec = {"options": extra}
            for key, tail in (("m8", ["-m", "8"]), ("m7", ["-m", "7", "-b", "0"])):
                r = subprocess.run([REF] + base_args(case, base, qf) + extra + tail, capture_output=True, text=True)
                rec[key] = r.stdout
                rec[key + "_rc"] = r.returncode
                rec[key + "_err"] = r.stderr
            rows.append(rec)
            print(name, " ".join(extra), "->", rec["m8_rc"], len(rec["m8"].splitlines()), "tsv lines,", rec["m7"].count("<hit>"), "hits", rec["m8_err"].strip()[:80])
        out[name] = {"checksum": case.checksum(), "runs": rows}
    json.dump(out, open(os.path.join(HERE, "options.json"), "w"), separators=(",", ":"))


if __name__ == "__main__":
    main()
