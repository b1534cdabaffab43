"""Diagnostic (GPU box): is an env's IK evaluation count predictable from its previous step's?  python tests/tools/ik_persistence.py [env_id] [n] [steps]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from gym_kmanip_amd import env_hip
env_id = sys.argv[1] if len(sys.argv) > 1 else "KManipSoloArm"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
e = env_hip.make(env_id, num_envs=n, seed=0)
e.k_reset(); e.set_state(step=(np.arange(n) % 64).astype(np.int32))
for k in range(70):
    e.step_flat(e.sample_action())
NF = []
for k in range(steps):
    e.step_flat(e.sample_action())
    NF.append(e.get_diag()[1].max(1).copy())
NF = np.array(NF, dtype=np.float64)
a, b = NF[:-1].ravel(), NF[1:].ravel()
ok = (a > 0) & (b > 0)                      # (0 = the env was reset in that step)
a, b = a[ok], b[ok]
print("%s: IK evaluations per step: mean %.1f p50 %.0f p90 %.0f p99 %.0f max %.0f" % (env_id, a.mean(), *np.percentile(a, [50, 90, 99]), a.max()))
print("  lag-1 correlation of the count %.3f; of log(count) %.3f" % (np.corrcoef(a, b)[0, 1], np.corrcoef(np.log(a), np.log(b))[0, 1]))
for thr in (24, 28, 32, 40):
    p = (b >= thr).mean(); pc = (b[a >= thr] >= thr).mean() if (a >= thr).any() else float("nan")
    pc2 = (b[a >= 20] >= thr).mean()
    print("  P(count >= %d) = %.4f;  given the previous step's count >= %d: %.4f (x%.1f);  given previous >= 20 (%.1f %% of envs): %.4f" % (thr, p, thr, pc, pc / p, 100 * (a >= 20).mean(), pc2))
# ---- does a joint sitting at (or near) a limit predict a long IK in the NEXT step?
d = e.cm.desc
nl = e.cm.nlink
lo = np.array([d.jnt_range[i][0] for i in range(nl)]); hi = np.array([d.jnt_range[i][1] for i in range(nl)])
span = np.where(hi > lo, hi - lo, 1.0)
F, Y = [], []
prev = None
for k in range(120):
    q = e.get_state()[0][:, :nl]
    margin = np.minimum(q - lo, hi - q) / span
    nq = d.arm_nq[0]
    feat = margin[:, :nq].min(1)                      # the arm's IK joints
    e.step_flat(e.sample_action())
    nf = e.get_diag()[1].max(1)
    F.append(feat); Y.append(nf)
F = np.concatenate(F); Y = np.concatenate(Y).astype(np.float64)
ok = Y > 0
F, Y = F[ok], Y[ok]
for thr in (0.0005, 0.002, 0.01, 0.05):
    sel = F < thr
    print("  joints within %.2f %% of a limit before the step: %.1f %% of envs; their P(count >= 24) = %.4f, mean count %.1f  (all: %.4f, %.1f)" % (
        100 * thr, 100 * sel.mean(), (Y[sel] >= 24).mean() if sel.any() else float("nan"), Y[sel].mean() if sel.any() else float("nan"), (Y >= 24).mean(), Y.mean()))
print("  of the envs with count >= 24, the share that had a joint within 1 %% of a limit: %.2f" % ((F[Y >= 24] < 0.01).mean()))
