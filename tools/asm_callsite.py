#!/usr/bin/env python3
"""Static instruction counts of the code inlined (transitively) through ONE call site, grouped by innermost source line.
Usage: tools/asm_callsite.py <listing.s> <kernel-symbol-substring> <file:line of the call site> [n]
The listing must come from `hipcc -S -gline-tables-only` (its .loc comments carry the inlined-at chain)."""
import collections, re, sys
path, sym, site = sys.argv[1], sys.argv[2], sys.argv[3]
n = int(sys.argv[4]) if len(sys.argv) > 4 else 40
inside = False; cur = None; on = False
cnt = collections.Counter(); kinds = collections.defaultdict(collections.Counter)
for ln in open(path, errors="replace"):
    s = ln.strip()
    if re.match(r"^[_A-Za-z0-9.$]+:", s) and not s.startswith("."):
        inside = sym in s; continue
    if not inside: continue
    if s.startswith(".loc"):
        m = re.search(r";\s*(\S+?):(\d+):\d+(.*)", s)
        if m:
            cur = (m.group(1).split("/")[-1], int(m.group(2)))
            on = (site + ":") in m.group(3) or ("%s:%d" % cur) == site
        continue
    if not s or s.startswith((".", ";", "//")) or s.endswith(":"): continue
    if on and cur:
        cnt[cur] += 1; kinds[cur][s.split()[0]] += 1
print("instructions inlined through %s: %d" % (site, sum(cnt.values())))
for k, c in cnt.most_common(n):
    print("  %s:%d  %d  %s" % (k[0], k[1], c, dict(kinds[k].most_common(5))))
