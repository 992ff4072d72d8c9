"""Register budget of the re-queue follower beside every first-pass build it may run with.

Reads the gfx950 code objects out of swipe_amd/csrc/*.o (clang offload bundles; no GPU needed), takes .vgpr_count of every
kernel from the notes (llvm-readelf) and checks, for every build the host lets a follower run beside (run_search /
run_search2: bound builds and exact builds of at most 32 rows, never more than FOLLOW_MAX_ROWS = 48 rows per lane), whether
ONE BLOCK of the producer and one follower wave fit a SIMD's 512 registers (allocation granule 8):
    waves per SIMD of one block x ceil8(producer) + ceil8(follower of a query of G x K rows) <= 512
256-thread blocks put one wave on each SIMD, so a contested CU just holds a block less; 512-thread blocks (long two-query
bound builds) put two, and that is where round 3's hang lived (52 rows: 2 x 224 + 72).

    python tools/kernel_registers.py            # table of the 512-thread builds + every violation among eligible builds
    python tools/kernel_registers.py --all      # every kernel's register count
"""
import glob, os, re, struct, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
FOLLOW_MAX_ROWS = 48


def code_objects(path):
    data = open(path, "rb").read()
    magic, pos = b"__CLANG_OFFLOAD_BUNDLE__", 0
    while True:
        i = data.find(magic, pos)
        if i < 0:
            return
        n = struct.unpack_from("<Q", data, i + 24)[0]
        p = i + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, p)
            p += 24
            triple = data[p:p + tl].decode()
            p += tl
            if "gfx950" in triple and size:
                yield data[i + off:i + off + size]
        pos = i + 24


def kernels(blob):
    tmp = "/tmp/_kernel_registers.co"
    open(tmp, "wb").write(blob)
    txt = subprocess.run([READELF, "--notes", tmp], capture_output=True, text=True).stdout
    sym, tpb = None, 0
    for line in txt.splitlines():
        m = re.search(r"\.max_flat_workgroup_size:\s+(\d+)", line)
        if m:
            tpb = int(m.group(1))
        m = re.search(r"\.symbol:\s+(\S+)\.kd", line)
        if m:
            sym = m.group(1)
        m = re.search(r"\.vgpr_count:\s+(\d+)", line)
        if m and sym:
            name = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip()
            yield re.sub(r"\(.*", "", name).replace("void ", ""), int(m.group(1)), tpb
            sym = None


def main():
    table = {}
    for o in sorted(glob.glob(os.path.join(ROOT, "swipe_amd", "csrc", "*.o"))):
        for blob in code_objects(o):
            for name, v, tpb in kernels(blob):
                table[name] = (v, tpb)
    if "--all" in sys.argv:
        for name, (v, tpb) in sorted(table.items()):
            print("%4d %4d  %s" % (v, tpb, name))
        return
    follower = {int(re.search(r"<(\d+)>", n).group(1)): v for n, (v, _) in table.items() if n.startswith("swa_requeue_follow_kernel")}
    c8 = lambda v: (v + 7) // 8 * 8
    def follower_rows(qlen):
        return next((k for k in sorted(follower) if 64 * k >= qlen), max(follower))
    bad = 0
    print("follower registers by rows per lane:", follower)
    for name, (v, tpb) in sorted(table.items(), key=lambda t: (t[0].split("<")[0], [int(x) for x in re.findall(r"\d+", t[0])])):
        m = re.match(r"(swa_narrow_bound_kernel|swa_dual_bound_kernel|swa_one_bound_kernel|swa_narrow_split_kernel|swa_narrow_one_kernel|"
                     r"swa_dual_kernel|swa_dual_one_kernel)<(.*)>", name)
        if not m:
            continue
        kind, a = m.group(1), [x.strip() for x in m.group(2).split(",")]
        K = int(a[0])
        G = {"swa_narrow_bound_kernel": 2, "swa_dual_bound_kernel": 2, "swa_narrow_split_kernel": 2, "swa_dual_kernel": 3}.get(kind)
        G = int(a[G]) if G is not None else 1
        if "true" in a[4:]:
            continue                                        # pass builds: no follower
        exact = kind in ("swa_narrow_split_kernel", "swa_narrow_one_kernel", "swa_dual_kernel", "swa_dual_one_kernel")
        eligible = K <= FOLLOW_MAX_ROWS and not (exact and K > 32)
        fk = follower_rows(G * K)
        need = (tpb // 256) * c8(v) + c8(follower[fk])
        if tpb == 512:
            print("%-46s %3d registers x 2 waves + follower<%d> %3d = %3d  %s" % (name, v, fk, follower[fk], need,
                  "ok" if need <= 512 else ("NO ROOM" + ("" if eligible else " (no follower: more than %d rows)" % FOLLOW_MAX_ROWS))))
        if eligible and need > 512:
            bad += 1
            print("VIOLATION: %s" % name)
    print("%d eligible builds without room for a follower beside one of their blocks" % bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
