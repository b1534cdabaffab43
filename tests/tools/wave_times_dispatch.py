"""Diagnostic (GPU box): per-wave k_step cycles of single-arm launches under the heavy-first dispatch.
   python tests/tools/wave_times_dispatch.py [n_envs] [launches]        (KMANIP_HEAVY_DISPATCH=0 for the classic mapping)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
os.environ["KMANIP_WAVE_CLOCKS"] = "1"
import numpy as np, torch
from gym_kmanip_amd import env_hip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
L = int(sys.argv[2]) if len(sys.argv) > 2 else 12
e = env_hip.make("KManipSoloArm", num_envs=n, seed=0)
e.k_reset(); e.set_state(step=(np.arange(n) % 64).astype(np.int32))
for k in range(80):
    e.step_flat(e.sample_action())
S = e.L.kmanip_dbg_wave_slots(e.h)
epb = 4 if n >= 4096 else 2
disp = S > n
allw, heavy_w, light_w, nheavy, late, pairs, ntab = [], [], [], [], [], [], []
coupled_prev = None
for k in range(L):
    e.step_flat(e.sample_action())
    clk = np.zeros(S, dtype=np.uint64); slot = np.full(S, -1, dtype=np.int32)
    e.L.kmanip_dbg_wave_clocks(e.h, clk.ctypes.data_as(C.POINTER(C.c_ulonglong)), slot.ctypes.data_as(C.POINTER(C.c_int32)), None)   # (the launch's own slot -> env map)
    dg = e.get_diag()
    nf = dg[1].max(1); coupled_now = (dg[0] & 0xFFF00) != 0          # (end-of-step state: a sphere on the cube)
    table_now = (dg[0].astype(np.uint32) & np.uint32(0xFFF00000)) != 0                       # (a sphere on the table)
    was = coupled_prev if k else np.zeros(n, dtype=bool)
    coupled_prev = coupled_now.copy()
    start = ((clk >> np.uint64(40)) & np.uint64(0xFFFFFF)).astype(np.int64)
    ticks = (clk & np.uint64(0xFFFFFFFFFF)).astype(np.float64)
    sl = slot[:(S // epb) * epb].reshape(-1, epb); tk = ticks[:(S // epb) * epb].reshape(-1, epb); st = start[:(S // epb) * epb].reshape(-1, epb)
    used = (sl >= 0).any(1)
    cnt = (sl >= 0).sum(1)
    first = np.argmax(sl >= 0, 1)
    w = tk[np.arange(len(tk)), first][used]; s0 = st[np.arange(len(st)), first][used]; c = cnt[used]
    s0 = (s0 - s0.min()) % (1 << 24)
    allw.append(w)
    if disp:
        hv = c < epb
        hv[-1] = False if c[-1] < epb and not (c[:-1] < epb).all() else hv[-1]      # the last light wave may be partly filled
        heavy_w.append(w[hv]); light_w.append(w[~hv]); nheavy.append(int(hv.sum()))
    late.append((s0 > 2000).sum())        # waves that started more than 20 us after the first
    mx = np.argmax(w)
    top = np.argsort(w)[::-1][:4]
    envs_of = sl[used]
    tops = "; ".join("%.0f nfev %s cpl %s was %s tab %s" % (w[i], nf[envs_of[i][envs_of[i] >= 0]], coupled_now[envs_of[i][envs_of[i] >= 0]].astype(int), was[envs_of[i][envs_of[i] >= 0]].astype(int), table_now[envs_of[i][envs_of[i] >= 0]].astype(int)) for i in top)
    ntab.append(float(table_now.mean()))
    ncw = coupled_now[np.where(envs_of >= 0, envs_of, 0)].sum(1)
    pairs.append(int((ncw >= 2).sum()))
    print("   top 4 waves: " + tops)
    print("launch %2d: waves %4d  max %.0f (%s, %d env(s), nfev %s, started +%d us)  mean %.0f  p99 %.0f   late-started waves %d%s" % (
        k, len(w), w[mx], "heavy" if disp and c[mx] < epb else "light", c[mx], nf[sl[used][mx][sl[used][mx] >= 0]], s0[mx] // 100, w.mean(), np.percentile(w, 99), late[-1],
        "   heavy waves %d: mean %.0f max %.0f | light: mean %.0f p99 %.0f max %.0f" % (nheavy[-1], heavy_w[-1].mean() if nheavy[-1] else 0, heavy_w[-1].max() if nheavy[-1] else 0,
                                                                                  light_w[-1].mean(), np.percentile(light_w[-1], 99), light_w[-1].max()) if disp else ""))
w = np.concatenate(allw)
print("all launches: wave ticks mean %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f; mean of the launches' max %.0f" % (
    w.mean(), np.median(w), np.percentile(w, 90), np.percentile(w, 99), w.max(), np.mean([a.max() for a in allw])))
print("waves that ended the step holding two or more envs with a sphere on the cube: mean %.2f per launch; envs with a sphere on the TABLE at the end of a step: %.1f %%" % (np.mean(pairs), 100 * np.mean(ntab)))
if disp:
    print("heavy envs per launch: mean %.1f (cap %d); heavy-wave ticks mean %.0f p90 %.0f max %.0f; light-wave ticks mean %.0f p99 %.0f max %.0f" % (
        np.mean(nheavy), (S // 4 - n - 4), np.concatenate(heavy_w).mean(), np.percentile(np.concatenate(heavy_w), 90), np.concatenate(heavy_w).max(),
        np.concatenate(light_w).mean(), np.percentile(np.concatenate(light_w), 99), np.concatenate(light_w).max()))
