#!/usr/bin/env python3
"""Static instruction MIX per source function of ONE kernel in a `hipcc -S -gline-tables-only` listing: how many of a function's
instructions are f64 arithmetic, and how many are moves / selects / AGPR shuttles / lane reads / LDS / scalar.
Usage: tools/asm_mix.py <listing.s> <kernel-symbol-substring> [n hottest lines of the non-arithmetic classes]"""
import collections, os, re, sys

path, sym = sys.argv[1], sys.argv[2]
nhot = int(sys.argv[3]) if len(sys.argv) > 3 else 0


def klass(op, s):
    if op.startswith("v_accvgpr"):
        return "agpr"
    if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
        return "lane"
    if op.startswith("v_cndmask"):
        return "sel"
    if op.startswith("v_mov") or op.startswith("v_pk_mov"):
        return "dppmov" if ("dpp" in op or "row_" in s or "quad_perm" in s) else "mov"
    if op.startswith(("v_permlane", "v_swap")):
        return "perm"
    if "f64" in op:
        if op.startswith("v_cmp"):
            return "cmp64"
        return "f64"
    if op.startswith("v_cmp"):
        return "cmp"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith(("s_cbranch", "s_branch")):
        return "br"
    if op.startswith("s_"):
        return "salu"
    return "other"


files = {}
cur = None
inside = False
counts = collections.defaultdict(collections.Counter)
for ln in open(path, errors="replace"):
    s = ln.strip()
    if s.startswith(".file"):
        m = re.match(r'\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', s)
        if m:
            files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
        continue
    if re.match(r"^[_A-Za-z0-9.$]+:", s) and not s.startswith("."):
        inside = sym in s
        continue
    if not inside:
        continue
    if s.startswith(".loc"):
        p = s.split()
        cur = (files.get(int(p[1]), p[1]), int(p[2]))
        continue
    if not s or s.startswith((".", ";", "//")) or s.endswith(":"):
        continue
    op = s.split()[0]
    if cur:
        counts[cur][klass(op, s)] += 1


def func_ranges(src):
    starts = []
    prev = ""
    for i, t in enumerate(open(src, errors="replace").read().split("\n"), 1):
        if not t.startswith((" ", "\t", "#", "/")):
            m = re.search(r"\b(?:void|real|int|bool|double|float|uint32_t|BSrc<G>)\s+([A-Za-z_0-9]+)\s*\(", t)
            if m and ("__device__" in t or "__global__" in t or "__device__" in prev or "__global__" in prev):
                starts.append((i, m.group(1)))
        prev = t
    return starts


here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gym_kmanip_amd", "csrc")
cols = ["f64", "mov", "dppmov", "sel", "agpr", "lane", "perm", "cmp64", "cmp", "valu", "lds", "vmem", "salu", "br", "nop", "wait"]
tot = collections.Counter()
agg = collections.defaultdict(collections.Counter)
for fname in ("kmanip_dyn.hip", "kmanip_ik_coop.hpp", "kmanip_device.hpp"):
    st = func_ranges(os.path.join(here, fname))
    for (f, l), c in counts.items():
        if f != fname:
            continue
        name = "?"
        for s0, n in st:
            if s0 <= l:
                name = n
            else:
                break
        agg[name].update(c)
        tot.update(c)
for (f, l), c in counts.items():
    if f not in ("kmanip_dyn.hip", "kmanip_ik_coop.hpp", "kmanip_device.hpp"):
        agg["<" + f + ">"].update(c)
        tot.update(c)
print("%-28s %6s | " % ("function", "total") + " ".join("%6s" % c for c in cols))
for name, c in sorted(agg.items(), key=lambda kv: -sum(kv[1].values())):
    print("%-28s %6d | " % (name[:28], sum(c.values())) + " ".join("%6d" % c[k] for k in cols))
print("%-28s %6d | " % ("TOTAL", sum(tot.values())) + " ".join("%6d" % tot[k] for k in cols))
if nhot:
    print("== hottest lines by non-arithmetic VALU (mov+dppmov+sel+agpr+lane)")
    key = lambda c: c["mov"] + c["dppmov"] + c["sel"] + c["agpr"] + c["lane"]
    for (f, l), c in sorted(counts.items(), key=lambda kv: -key(kv[1]))[:nhot]:
        print("  %s:%d  %s" % (f, l, dict(c.most_common(6))))
