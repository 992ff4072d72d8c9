#!/usr/bin/env python3
"""gfx950sim, step 1: turn one clang offload bundle (what __hipRegisterFatBinary is handed) into what the
interpreter reads - the gfx950 code object's disassembly (llvm-objdump text) and a flat metadata file.

    prep.py <bundle-file> <out-dir>

writes  <out-dir>/co.elf   the gfx950 code object
        <out-dir>/co.s     llvm-objdump -d --mcpu=gfx950
        <out-dir>/co.meta  one 'K' line per kernel + one 'A' line per kernel argument:
             K <name> <code_addr> <lds_bytes> <scratch_bytes> <kernarg_bytes> <rsrc1> <rsrc2> <rsrc3> <code_properties> <preload>
             A <offset> <size> <value_kind>
        <out-dir>/ok       written last (the cache entry is complete)

Test infrastructure only: nothing in swipe_amd/ uses it. Needs /opt/rocm/lib/llvm/bin (present in the ROCm image).
"""
import os
import struct
import subprocess
import sys

LLVM = os.environ.get("HIPSIM_LLVM", "/opt/rocm/lib/llvm/bin")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def unbundle(blob: bytes) -> bytes:
    if not blob.startswith(MAGIC):
        raise SystemExit("prep.py: not an uncompressed clang offload bundle")
    (n,) = struct.unpack_from("<Q", blob, 24)
    p = 32
    for _ in range(n):
        off, size, tl = struct.unpack_from("<QQQ", blob, p)
        triple = blob[p + 24:p + 24 + tl].decode()
        p += 24 + tl
        if "gfx950" in triple:
            return blob[off:off + size]
    raise SystemExit("prep.py: the bundle holds no gfx950 code object")


def elf_sections(elf: bytes):
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", elf, 0x3A)
    secs = []
    for i in range(shnum):
        name, typ, flags, addr, off, size, link, info, align, entsize = struct.unpack_from("<IIQQQQIIQQ", elf, shoff + i * shentsize)
        secs.append(dict(name=name, type=typ, addr=addr, off=off, size=size, link=link, entsize=entsize))
    strtab = secs[shstrndx]
    for s in secs:
        e = elf.index(b"\0", strtab["off"] + s["name"])
        s["name"] = elf[strtab["off"] + s["name"]:e].decode()
    return secs


def elf_symbols(elf: bytes, secs):
    out = {}
    for s in secs:
        if s["type"] not in (2, 11):        # SYMTAB, DYNSYM
            continue
        st = secs[s["link"]]
        for i in range(s["size"] // 24):
            name, info, other, shndx, value, size = struct.unpack_from("<IBBHQQ", elf, s["off"] + i * 24)
            e = elf.index(b"\0", st["off"] + name)
            out[elf[st["off"] + name:e].decode()] = (value, size, shndx)
    return out


def read_at(elf: bytes, secs, addr: int, n: int) -> bytes:
    for s in secs:
        if s["type"] != 8 and s["addr"] <= addr and addr + n <= s["addr"] + s["size"] and s["addr"]:
            o = s["off"] + addr - s["addr"]
            return elf[o:o + n]
    raise SystemExit(f"prep.py: address {addr:#x} is in no section")


def main():
    src, out = sys.argv[1], sys.argv[2]
    os.makedirs(out, exist_ok=True)
    elf = unbundle(open(src, "rb").read())
    co = os.path.join(out, "co.elf")
    open(co, "wb").write(elf)
    with open(os.path.join(out, "co.s"), "wb") as f:
        subprocess.check_call([os.path.join(LLVM, "llvm-objdump"), "-d", "--mcpu=gfx950", co], stdout=f)
    notes = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", co]).decode()
    import yaml
    doc = notes[notes.index("---"):]
    doc = doc[:doc.index("\n...")] if "\n..." in doc else doc
    md = yaml.safe_load(doc)
    secs = elf_sections(elf)
    syms = elf_symbols(elf, secs)
    lines = []
    for k in md.get("amdhsa.kernels", []):
        name = k[".name"]
        kd_addr = syms[k[".symbol"]][0]
        kd = read_at(elf, secs, kd_addr, 64)
        lds, scratch, kernarg = struct.unpack_from("<III", kd, 0)
        entry, = struct.unpack_from("<q", kd, 16)
        rsrc3, rsrc1, rsrc2 = struct.unpack_from("<III", kd, 44)
        props, preload = struct.unpack_from("<HH", kd, 56)
        lines.append(f"K {name} {kd_addr + entry} {lds} {scratch} {kernarg} {rsrc1} {rsrc2} {rsrc3} {props} {preload}")
        for a in k.get(".args", []):
            lines.append(f"A {a['.offset']} {a['.size']} {a['.value_kind']}")
    open(os.path.join(out, "co.meta"), "w").write("\n".join(lines) + "\n")
    open(os.path.join(out, "ok"), "w").write("1\n")


if __name__ == "__main__":
    main()
