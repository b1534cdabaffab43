"""Diagnostic (GPU box): what would a launch of K control steps (no grid-wide barrier between the steps: every wave runs its own
envs K steps on) last?  Per-wave k_step cycles of consecutive launches of the bench workload: the per-step launches last
sum_k max_w t[k, w]; a K-step launch would last max_w sum_k t[k, w] per chunk.
   KMANIP_WAVE_CLOCKS=1 python tests/tools/wave_chain_sums.py [launches]"""
import ctypes as C, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
os.environ["KMANIP_WAVE_CLOCKS"] = "1"
import numpy as np, torch
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 256
w = bench.Workload(torch, "KManipSoloArm", 4096, 0, 0, 0, "newton", 100)
env, n, epb = w.env, 4096, 4
w.lay_out(steps)
T = np.zeros((steps, n // epb))
for k in range(steps):
    w.step()
    S = env.L.kmanip_dbg_wave_slots(env.h)       # entries of clk / slot_env (include/kmanip_debug.h): more than n with the heavy-first dispatch
    assert S == n, "this tool reads the plain grid (unset KMANIP_HEAVY_DISPATCH / KMANIP_HEAVY_EPB)"
    clk = np.zeros(S, dtype=np.uint64); slot = np.zeros(S, dtype=np.int32); work = np.zeros(n, dtype=np.int32)
    env.L.kmanip_dbg_wave_clocks(env.h, clk.ctypes.data_as(C.POINTER(C.c_ulonglong)), slot.ctypes.data_as(C.POINTER(C.c_int32)), work.ctypes.data_as(C.POINTER(C.c_int32)))
    T[k] = (clk & np.uint64(0xFFFFFFFFFF)).reshape(-1, epb)[:, 0].astype(np.float64)      # (wave slots are fixed at 4096 envs: no cost sort)
per_step = T.max(1).sum()
print("KManipSoloArm @ 4096 envs, %d launches: wave cycles mean %.0f, mean of the launches' max %.0f (%.2f x the mean)" % (steps, T.mean(), T.max(1).mean(), T.max(1).mean() / T.mean()))
for K in (1, 2, 4, 8, 16, 32, 64, steps):
    m = sum(T[i:i + K].sum(0).max() for i in range(0, steps, K))
    print("  K = %3d steps per launch: %.3f of the per-step launches' time  (%.0f cycles per step)" % (K, m / per_step, m / steps))
# how persistent is a heavy wave?
r = np.corrcoef(T[:-1].ravel(), T[1:].ravel())[0, 1]
print("  correlation of a wave's time with its own time one step later: %.3f" % r)
