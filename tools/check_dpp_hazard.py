#!/usr/bin/env python3
"""Static check of a gfx950 ISA listing for the hazard class round 3 ran into: a DPP read of a VGPR less than two wait states after
a VALU write of it ("VALU writes VGPR -> v_*_dpp reads that VGPR: 2 wait states").  LLVM's hazard recogniser inserts the wait
states for its own instructions but does not look inside inline asm, and the kernels fold row broadcasts into FMAs with inline asm
(kmanip_device.hpp) -- so the final listing is checked instead: every instruction that reads an operand through DPP, against the two
instructions in front of it (an `s_nop N` counts N + 1 wait states).  Labels reset the window (a branch target's predecessors are
not known here; the asm runs start with their own wait states or sit behind a run that does).

Usage: tools/check_dpp_hazard.py <listing.s> [...]     (listings from `hipcc -S --cuda-device-only`, e.g. tools/mix.sh's output)
Exit code 1 if a hazard is found."""
import re
import sys


def regs(op):
    """Set of VGPR numbers an operand names: v5, v[4:5], -v[4:5], |v3|."""
    op = op.strip().lstrip("-|").rstrip("|")
    m = re.match(r"^v\[(\d+):(\d+)\]$", op)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"^v(\d+)$", op)
    return {int(m.group(1))} if m else set()


def check(path):
    bad = 0
    ndpp = 0
    window = []                      # [(wait states this slot provides, set of VGPRs its VALU write defines, text)]
    kernel = "?"
    for n, ln in enumerate(open(path, errors="replace"), 1):
        s = ln.split(";")[0].strip()
        if not s or s.startswith((".", "//")):
            continue
        if s.endswith(":"):
            if not s.startswith((".L", "BB")) and re.match(r"^[_A-Za-z0-9.$]+:$", s):
                kernel = s[:-1]
            window = []
            continue
        parts = s.split(None, 1)
        op = parts[0]
        args = [a.strip() for a in parts[1].split(",")] if len(parts) > 1 else []
        is_dpp = ("_dpp" in op) or any(k in s for k in (" row_", " quad_perm", " wave_", " row_newbcast"))
        if is_dpp and op.startswith("v_") and len(args) >= 2:
            ndpp += 1
            src = regs(args[1].split()[0])          # the DPP operand is src0
            waits = 0
            for w, defs, text in reversed(window):
                if defs & src and waits < 2:
                    print("%s:%d [%s] DPP read %s only %d wait state(s) after: %s" % (path, n, kernel[:40], args[1].split()[0], waits, text))
                    print("        " + s)
                    bad += 1
                    break
                waits += w
                if waits >= 2:
                    break
        if op == "s_nop":
            window.append((int(args[0]) + 1 if args else 1, set(), s))
        elif op.startswith("v_") and args and not op.startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
            window.append((1, regs(args[0].split()[0]), s))
        else:
            window.append((1, set(), s))
        window = window[-4:]
    print("%s: %d DPP-operand instructions checked, %d hazard(s)" % (path, ndpp, bad))
    return bad


if __name__ == "__main__":
    sys.exit(1 if sum(check(p) for p in sys.argv[1:]) else 0)
