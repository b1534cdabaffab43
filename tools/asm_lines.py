#!/usr/bin/env python3
"""Static instruction counts per source line / function of ONE kernel in a `hipcc -S -gline-tables-only` listing.
Usage: tools/asm_lines.py <listing.s> <kernel-symbol-substring> [lines]
With the .loc directives of the listing, every instruction is attributed to the source line the compiler says it came
from (inlined frames: the innermost line); the report groups lines into the functions of kmanip_dyn.hip / the headers."""
import collections, os, re, sys

path, sym = sys.argv[1], sys.argv[2]
files = {}
cur = None
inside = False
counts = collections.Counter()
kinds = collections.defaultdict(collections.Counter)
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
        counts[cur] += 1
        kinds[cur][op] += 1
tot = sum(counts.values())
print("total instructions attributed:", tot)
byfile = collections.Counter()
for (f, l), c in counts.items():
    byfile[f] += c
print("by file:", byfile.most_common(8))


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
for fname in ("kmanip_dyn.hip", "kmanip_ik_coop.hpp", "kmanip_device.hpp"):
    st = func_ranges(os.path.join(here, fname))
    agg = collections.Counter()
    for (f, l), c in counts.items():
        if f != fname:
            continue
        name = "?"
        for s0, n in st:
            if s0 <= l:
                name = n
            else:
                break
        agg[name] += c
    print("==", fname)
    for n, c in agg.most_common(45):
        print("  %-32s %7d  %5.1f%%" % (n, c, 100.0 * c / tot))
if len(sys.argv) > 3:
    print("== hottest lines")
    for (f, l), c in counts.most_common(int(sys.argv[3])):
        print("  %s:%d  %d  %s" % (f, l, c, dict(kinds[(f, l)].most_common(4))))
