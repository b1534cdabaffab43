#!/usr/bin/env python3
"""Diagnostic (GPU box): which launches of the bench workload are slow, and why.  Replays bench.py's desynchronised workload,
times every kmanip_step with events and prints, for the slow ones, the IK evaluation counts (kmanip_get_diag) of the batch.
   python tests/tools/slow_launches.py [launches] [ik_max_nfev]     (ik_max_nfev: the opt-in cap of KModelDesc, 0 = the reference's 100 n)"""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import torch
import bench

args = bench.parse_args(["--no-variants", "--no-cpu-baseline"])
cap = int(sys.argv[2]) if len(sys.argv) > 2 else 0
w = bench.Workload(torch, "KManipSoloArm", 4096, 0, 0, 0, "newton", 100, ik_max_nfev=cap)
print("KManipSoloArm @ 4096 envs, ik_max_nfev = %d%s" % (cap, " (the reference's default, 100 n)" if cap == 0 else " (opt-in cap)"))
env = w.env
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
w.lay_out(steps)                      # bench.py's action stream (Philox keyed (seed; env id, episode, step))
rows = []
slow_envs = {}
for k in range(steps):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); w.step(); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    mask, nfev, st = env.get_diag()
    rows.append((ms, int(nfev[:, 0].max()), int(((mask & 0xFFF00) != 0).sum())))
    if ms > 2.0 or k < 3:
        j = int(nfev[:, 0].argmax())
        print("step %3d  %.3f ms  max nfev %d (env %d, status %d)  nfev>100: %d envs  contact mask of that env %s" % (
            k, ms, nfev[j, 0], j, st[j, 0], int((nfev[:, 0] > 100).sum()), hex(int(mask[j]))))
        if ms > 2.0:
            slow_envs.setdefault(j, []).append(k)

r = np.array(rows, dtype=np.float64)
if os.environ.get("KM_LAUNCH_DUMP"):       # one launch time per line (ms), for tools/scaling_model.py
    np.savetxt(os.environ["KM_LAUNCH_DUMP"], r[:, 0], fmt="%.4f", header="k_step launch times (ms), KManipSoloArm @ 4096 envs, bench.py's workload, ik_max_nfev %d" % cap)
print("launch ms: mean %.3f  p50 %.3f  p90 %.3f  p99 %.3f  max %.3f" % (r[:, 0].mean(), *np.percentile(r[:, 0], [50, 90, 99]), r[:, 0].max()))
print("  (end-of-step state) max IK nfev per launch: p50 %d p90 %d max %d; envs with a sphere-cube contact: mean %.1f" % (
    *np.percentile(r[:, 1], [50, 90]), r[:, 1].max(), r[:, 2].mean()))
for lo, hi in [(0, 100), (100, 200), (200, 400), (400, 10000)]:
    sel = (r[:, 1] >= lo) & (r[:, 1] < hi)
    if sel.any():
        print("  launches with max nfev in [%d, %d): %4d  mean %.3f ms" % (lo, hi, sel.sum(), r[sel, 0].mean()))
print("  slow launches (> 2 ms) by the env with the largest nfev: %s" % {e: ks for e, ks in sorted(slow_envs.items())})
