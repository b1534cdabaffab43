#!/usr/bin/env python3
"""Extract the gfx950 code object(s) of a HIP object / shared library and disassemble them (llvm-objdump -d).
Usage: tools/disasm.py <file.o|.so> <out_prefix>   ->  <out_prefix>.<k>.s"""
import subprocess, sys
sys.path.insert(0, __import__("os").path.dirname(__file__))
from kernel_resources import code_objects

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
blob = open(sys.argv[1], "rb").read()
for k, (triple, co) in enumerate(code_objects(blob)):
    path = "%s.%d.co" % (sys.argv[2], k)
    open(path, "wb").write(co)
    with open("%s.%d.s" % (sys.argv[2], k), "w") as f:
        subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", path], stdout=f)
    print(triple, len(co), "->", "%s.%d.s" % (sys.argv[2], k))
