#!/usr/bin/env python3
"""Print VGPR/AGPR/scratch/LDS per kernel of a HIP object or shared library (parses the clang offload bundle)."""
import re, struct, subprocess, sys, tempfile

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def code_objects(blob):
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    pos = 0
    while True:
        pos = blob.find(magic, pos)
        if pos < 0:
            return
        n, = struct.unpack_from("<Q", blob, pos + 24)
        off = pos + 32
        for _ in range(n):
            o, sz, tl = struct.unpack_from("<QQQ", blob, off)
            triple = blob[off + 24: off + 24 + tl].decode()
            off += 24 + tl
            if "amdgcn" in triple and sz:
                yield triple, blob[pos + o: pos + o + sz]
        pos += 24


def main(path, pat="."):
    blob = open(path, "rb").read()
    for triple, co in code_objects(blob):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co); f.flush()
            txt = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
        for blk in txt.split("- .agpr_count:")[1:]:
            g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
            name = g("name")
            if re.search(pat, name):
                print("%-60s vgpr %s agpr %s sgpr %s scratch %s lds %s" % (
                    name[:60], g("vgpr_count"), blk.split()[0], g("sgpr_count"), g("private_segment_fixed_size"),
                    g("group_segment_fixed_size")))


if __name__ == "__main__":
    main(*sys.argv[1:])
