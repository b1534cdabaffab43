#!/usr/bin/env python3
"""Static check of a gfx950 ISA listing for the hazard class round 3 ran into: a DPP read of a VGPR less than two wait states after
a VALU write of it ("VALU writes VGPR -> v_*_dpp reads that VGPR: 2 wait states").  LLVM's hazard recogniser inserts the wait
states for its own instructions but does not look inside inline asm, and the kernels fold row broadcasts into FMAs with inline asm
(kmanip_device.hpp) -- so the final listing is checked instead: every instruction that reads an operand through DPP, against the two
instructions in front of it (an `s_nop N` counts N + 1 wait states).  The window is carried across a label the code falls through
into, reset behind an unconditional branch, and a branch to a label already seen (a loop back edge) is checked against that
label's first instructions; v_permlane*_swap counts as a write of both its operands.

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


def parse(s):
    parts = s.split(None, 1)
    op = parts[0]
    args = [a.strip() for a in parts[1].split(",")] if len(parts) > 1 else []
    return op, args


def slot(op, args, s):
    """(wait states this instruction provides, VGPRs its VALU write defines, text)"""
    if op == "s_nop":
        return (int(args[0]) + 1 if args else 1, set(), s)
    if op.startswith("v_") and args and not op.startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
        defs = regs(args[0].split()[0])
        if op.startswith("v_permlane") and "swap" in op and len(args) > 1:
            defs |= regs(args[1].split()[0])          # v_permlane16_swap / v_permlane32_swap exchange: BOTH operands are written
        return (1, defs, s)
    return (1, set(), s)


def dpp_src(op, args, s):
    is_dpp = ("_dpp" in op) or any(k in s for k in (" row_", " quad_perm", " wave_", " row_newbcast"))
    if is_dpp and op.startswith("v_") and len(args) >= 2:
        return regs(args[1].split()[0])              # the DPP operand is src0
    return None


def hazard(window, src):
    """The instruction in `window` (most recent last) whose write of a register in `src` is less than two wait states old."""
    waits = 0
    for w, defs, text in reversed(window):
        if defs & src and waits < 2:
            return waits, text
        waits += w
        if waits >= 2:
            break
    return None


def check(path):
    bad = 0
    ndpp = 0
    window = []                      # the last few instructions: [(wait states, VGPRs written, text)]
    heads = {}                       # label -> the first instructions after it (for back edges)
    open_heads = []                  # labels whose head is still being collected
    kernel = "?"
    prev_op = ""
    for n, ln in enumerate(open(path, errors="replace"), 1):
        s = ln.split(";")[0].strip()
        if not s or s.startswith(("//",)) or (s.startswith(".") and not s.endswith(":")):
            continue
        if s.endswith(":"):
            lab = s[:-1]
            if not lab.startswith((".L", "BB")) and re.match(r"^[_A-Za-z0-9.$]+$", lab):
                kernel = lab
                window = []
            elif prev_op in ("s_branch", "s_endpgm", "s_setpc_b64", "s_swappc_b64"):
                window = []          # no fall-through into this label: its predecessors are branches (checked at the branch)
            # (otherwise the window is CARRIED across the label: the fall-through path is a real predecessor)
            heads[lab] = []
            open_heads.append(lab)
            continue
        op, args = parse(s)
        for lab in list(open_heads):
            heads[lab].append((op, args, s))
            if len(heads[lab]) >= 3:
                open_heads.remove(lab)
        src = dpp_src(op, args, s)
        if src is not None:
            ndpp += 1
            h = hazard(window, src)
            if h:
                print("%s:%d [%s] DPP read %s only %d wait state(s) after: %s" % (path, n, kernel[:40], args[1].split()[0], h[0], h[1]))
                print("        " + s)
                bad += 1
        window.append(slot(op, args, s))
        window = window[-4:]
        # a branch to a label already seen (a loop back edge): the loop's last instructions precede the loop head's first ones
        if op.startswith(("s_cbranch", "s_branch")) and args and args[0] in heads and args[0] not in open_heads:
            w2 = list(window)
            for hop, hargs, hs in heads[args[0]]:
                hsrc = dpp_src(hop, hargs, hs)
                if hsrc is not None:
                    h = hazard(w2, hsrc)
                    if h:
                        print("%s:%d [%s] back edge to %s: DPP read %s only %d wait state(s) after: %s" % (path, n, kernel[:40], args[0], hargs[1].split()[0], h[0], h[1]))
                        bad += 1
                w2.append(slot(hop, hargs, hs))
        prev_op = op
    print("%s: %d DPP-operand instructions checked, %d hazard(s)" % (path, ndpp, bad))
    return bad


if __name__ == "__main__":
    sys.exit(1 if sum(check(p) for p in sys.argv[1:]) else 0)
