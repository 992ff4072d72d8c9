#!/usr/bin/env python3
"""tests/golden/multiquery.json: the reference CLI on a FASTA file holding several queries in awkward layouts
(no description on the first, wrapped lines, lower case, blank lines, digits and '*' inside, CR LF, an
empty query), against the `edges` database.  Build container only."""
import json, os, subprocess, sys, tempfile
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))
import cases
from swipe_amd import blastdb

REF = os.path.join(ROOT, "oracle", "_ref", "swipe")


def query_file_text():
    a = blastdb.NCBISTDAA
    q = "".join(a[c] for c in cases.Q375)
    wrapped = "\n".join(q[i:i + 60] for i in range(0, len(q), 60))
    return (q[:50] + "\n" + q[50:120] + "\n"                      # first query: no description line at all
            + ">second query with  double spaces and a long description " + "x" * 90 + "\n" + wrapped.lower() + "\n\n"
            + ">third\n" + q[200:260] + " 123 " + q[260:300] + "*\n" + "\r\n"
            + ">fourth_empty\n"
            + ">fifth|with|bars some text\n" + q[100:140] + "\n")


def main():
    case = cases.get("edges")
    d = tempfile.mkdtemp(prefix="golden_mq_")
    base = os.path.join(d, "db")
    blastdb.write_db(base, case.seqs, protein=True)
    qf = os.path.join(d, "q.fa")
    open(qf, "w").write(query_file_text())
    out = {"checksum": case.checksum(), "query_text": query_file_text()}
    for m, b in (("7", "3"), ("8", "10"), ("9", "10"), ("0", "2")):
        r = subprocess.run([REF, "-d", base, "-i", qf, "-m", m, "-b", b, "-v", "12", "-e", "1000"], capture_output=True, text=True)
        out["m" + m] = r.stdout
        out["rc" + m] = r.returncode
        out["err" + m] = r.stderr
        print("-m", m, "rc", r.returncode, len(r.stdout), "bytes", r.stderr[:200])
    json.dump(out, open(os.path.join(HERE, "multiquery.json"), "w"), separators=(",", ":"))


if __name__ == "__main__":
    main()
