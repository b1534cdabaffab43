"""Diagnostic (GPU box): per-wave k_step durations of one launch against the per-env cost predictors of k_sort_envs.
   KMANIP_WAVE_CLOCKS=1 python tests/tools/wave_times.py KManipDualArm 8192"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
os.environ["KMANIP_WAVE_CLOCKS"] = "1"
import numpy as np, torch
from gym_kmanip_amd import env_hip
env_id = sys.argv[1] if len(sys.argv) > 1 else "KManipDualArm"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
e = env_hip.make(env_id, num_envs=n, seed=0)
e.k_reset(); e.set_state(step=(np.arange(n) % 64).astype(np.int32))
for k in range(80):
    e.step_flat(e.sample_action())
epb = 4 if e.cm.nlink == 10 else 2
sorted_on = os.environ.get("KMANIP_COST_SORT", "1" if n > (4096 if e.cm.nlink == 10 else 2048) else "0") == "1"
rows = []
starts = []
for k in range(6):
    pre_work = np.zeros(n, dtype=np.int32)
    e.L.kmanip_dbg_wave_clocks(e.h, None, None, pre_work.ctypes.data_as(C.POINTER(C.c_int32)))
    pre_nf = e.get_diag()[1].max(1)
    e.step_flat(e.sample_action())
    S = e.L.kmanip_dbg_wave_slots(e.h)           # entries of clk / slot_env (include/kmanip_debug.h): more than n with the heavy-first dispatch
    assert S == n, "this tool reads the plain grid (unset KMANIP_HEAVY_DISPATCH / KMANIP_HEAVY_EPB; tests/tools/wave_times_dispatch.py handles those)"
    clk = np.zeros(S, dtype=np.uint64); slot = np.zeros(S, dtype=np.int32); work = np.zeros(n, dtype=np.int32)
    e.L.kmanip_dbg_wave_clocks(e.h, clk.ctypes.data_as(C.POINTER(C.c_ulonglong)), slot.ctypes.data_as(C.POINTER(C.c_int32)), work.ctypes.data_as(C.POINTER(C.c_int32)))
    # (slot is the launch's own slot -> env map: the sorted order, the SPREAD deal of a single-arm handle, or the identity;
    #  wave_clk is indexed by slot = wave index in slot space * EPB + lane group)
    nf = e.get_diag()[1].max(1)
    start = ((clk >> np.uint64(40)) & np.uint64(0xFFFFFF)).reshape(-1, epb)[:, 0].astype(np.int64)   # 100 MHz ticks, 24 bits
    clk = clk & np.uint64(0xFFFFFFFFFF)
    w = clk.reshape(-1, epb)[:, 0].astype(np.float64)                           # one entry per wave (its first slot)
    st0 = (start - start.min()) % (1 << 24)
    end_real = st0.max()                                                        # start of the last-dispatched wave, 10 ns units
    starts.append((st0, w))
    envs = slot.reshape(-1, epb)
    wk = work[envs] & 0x3FFFFFFF; nfw = nf[envs]; pw = (pre_work[envs] & 0x3FFFFFFF) + 100000 * (pre_work[envs] >> 30); pn = pre_nf[envs]
    rows.append((w, wk, nfw, pw, pn))
w = np.concatenate([r[0] for r in rows]); wk = np.concatenate([r[1] for r in rows]); nfw = np.concatenate([r[2] for r in rows])
pw = np.concatenate([r[3] for r in rows]); pn = np.concatenate([r[4] for r in rows])
print("%s n=%d sorted=%s: wave ticks mean %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f (x%d waves/slots %.1f rounds)" % (
    env_id, n, sorted_on, w.mean(), np.median(w), np.percentile(w, 90), np.percentile(w, 99), w.max(), len(rows[0][0]), len(rows[0][0]) / 1024))
# least-squares fit: wave ticks ~ c0 + c1 * max(work) + c2 * max(nfev) (this step's own counters: how well COULD they predict)
A = np.stack([np.ones_like(w), wk.max(1), nfw.max(1)], 1)
c, res, *_ = np.linalg.lstsq(A, w, rcond=None)
print("fit on this step's counters: ticks = %.0f + %.1f * max work + %.1f * max nfev;  R^2 %.3f" % (c[0], c[1], c[2], 1 - ((w - A @ c) ** 2).sum() / ((w - w.mean()) ** 2).sum()))
A2 = np.stack([np.ones_like(w), pw.max(1), pn.max(1)], 1)
c2, *_ = np.linalg.lstsq(A2, w, rcond=None)
print("fit on the PREVIOUS step's counters (what the sort sees): R^2 %.3f  coefficients %s" % (1 - ((w - A2 @ c2) ** 2).sum() / ((w - w.mean()) ** 2).sum(), np.round(c2, 1)))
print("corr(work, previous work) %.3f   corr(nfev, previous nfev) %.3f" % (np.corrcoef(wk.ravel(), pw.ravel())[0, 1], np.corrcoef(nfw.ravel(), pn.ravel())[0, 1]))
order = np.argsort(w)[::-1][:16]
print("slowest waves: ticks | work of its envs | nfev | previous work | previous nfev | predicted-cost rank of the wave (0 = first dispatched)")
for i in order:
    print("  %8.0f | %s | %s | %s | %s | slot-block %d" % (w[i], wk[i], nfw[i], pw[i], pn[i], i % len(rows[0][0])))

st0, ww = starts[-1]
o = np.argsort(st0)
print("last launch: wave starts (us after the first): p25 %.0f p50 %.0f p75 %.0f last %.0f;  waves started after 75 %% of the span: mean ticks %.0f" % (
    np.percentile(st0, 25) / 100, np.percentile(st0, 50) / 100, np.percentile(st0, 75) / 100, st0.max() / 100, ww[st0 > 0.75 * st0.max()].mean()))
# effective core clock: a wave's cycles / its real duration is not recorded; estimate from the first round (all start at ~0, the
# second-round waves start when a first-round wave ends): k-th start time vs k-th smallest first-round duration
first = np.sort(ww[st0 < 2000])          # waves that started within 20 us
later = np.sort(st0[st0 >= 2000])
k = min(len(first), len(later), 800)
if k > 100:
    f = np.polyfit(first[:k], later[:k] * 10.0, 1)   # ns per cycle
    print("core clock estimate from the first-round ends: %.2f GHz (%d waves)" % (1.0 / f[0], k))
# ---- how much of the launch is scheduling?  Greedy list scheduling of the LAST launch's measured wave durations on 1024 slots
import heapq
def makespan(durs, m=1024):
    h = [0.0] * m
    heapq.heapify(h)
    for d in durs:
        heapq.heappush(h, heapq.heappop(h) + d)
    return max(h)
d_last = starts[-1][1]
if len(d_last) > 1024:
    rng = np.random.default_rng(0)
    print("list scheduling of the last launch's waves on 1024 slots (M cycles): dispatch order as run %.2f | true longest-first %.2f | random %.2f | total / 1024 = %.2f | longest wave %.2f" % (
        makespan(d_last) / 1e6, makespan(np.sort(d_last)[::-1]) / 1e6, makespan(rng.permutation(d_last)) / 1e6, d_last.sum() / 1024 / 1e6, d_last.max() / 1e6))
