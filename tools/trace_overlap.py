#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace CSV: how much of every k_render_rgb dispatch ran WHILE a k_step dispatch was running.
   python tools/trace_overlap.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
name = lambda r: r.get("Kernel_Name") or r.get("Name") or ""
t0 = lambda r: int(r.get("Start_Timestamp") or r.get("Start"))
t1 = lambda r: int(r.get("End_Timestamp") or r.get("End"))
steps = sorted((t0(r), t1(r)) for r in rows if "k_step" in name(r))
rend = sorted((t0(r), t1(r)) for r in rows if "k_render_rgb" in name(r))
def overlap(a, b, ivs):
    return sum(max(0, min(b, y) - max(a, x)) for x, y in ivs if x < b and y > a)
ov = [(b - a, overlap(a, b, steps)) for a, b in rend]
# the run has two halves: renders behind the steps first, then in sequence -- split at the largest gap in overlap behaviour
half = len(ov) // 2
for tag, part in (("first half of the render dispatches (RenderBehind)", ov[:half]), ("second half (cameras in sequence)", ov[half:])):
    tot = sum(d for d, _ in part); o = sum(x for _, x in part)
    print("%-52s %4d dispatches, mean %.1f us each, %.1f %% of their time overlapped a running k_step" % (tag, len(part), tot / max(len(part), 1) / 1e3, 100.0 * o / max(tot, 1)))
span = lambda ivs: (ivs[-1][1] - ivs[0][0]) / 1e3 if ivs else 0.0
ks = [(a, b) for a, b in steps]
print("k_step dispatches: %d, mean %.1f us" % (len(ks), sum(b - a for a, b in ks) / max(len(ks), 1) / 1e3))
