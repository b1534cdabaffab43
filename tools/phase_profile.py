"""Diagnostic: per-phase cycle shares of k_step from the -DKM_PROFILE build (make -C gym_kmanip_amd/csrc prof).
Usage: KMANIP_LIB=gym_kmanip_amd/libkmanip_hip_prof.so python tools/phase_profile.py [solver]"""
import ctypes as C, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
os.environ.setdefault("KMANIP_LIB", os.path.join(ROOT, "gym_kmanip_amd", "libkmanip_hip_prof.so"))
import numpy as np, torch
from gym_kmanip_amd import env_hip
NPH = 48
names = ["fk", "bias bodies", "collide", "composite+mass+bias_proj", "invert_mass", "build_constraints", "solve (PGS)", "newton: a_s",
         "newton: start evals"] + ["newton %s: %s" % (sb, ph) for sb in ("ALL", "ARM", "CUBE") for ph in ("H build", "chol", "tri-solve", "ls setup", "ls loop", "eval")] + [
         "integrate", "post-solve (sibling wait)", "load state", "before_step (decode + IK)", "tail (reward/obs/store)", "auto-reset",
         "IK: residual + Jacobian", "IK: normal matrix", "IK: trust-region solve", "IK: select_step", "IK: ratio / radius / tests",
         "newton: start eval at qacc_smooth (when it beats the warm start)", "invert_mass: row loads (two-arm block path)"]
names += ["-"] * (40 - len(names))
names += ["newton: waiting for wave-mates in a loop this env does not run (divergence)", "COUNT coupled-env iterations on the partial refactorisation",
          "COUNT coupled-env iterations on the full factorisation", "COUNT wave-mates' non-plain iterations in the joint loop: partial + 65536 * full",
          "COUNT coupled-env iterations run ALONE (wave-mates out of the joint loop)", "COUNT coupled-env iterations beside wave-mates",
          "CYCLES of the coupled env's iterations run alone", "CYCLES of the coupled env's iterations beside wave-mates"]
solver = sys.argv[1] if len(sys.argv) > 1 else "newton"
env_id = sys.argv[2] if len(sys.argv) > 2 else "KManipSoloArm"          # 20-link ids: totals only (no per-workgroup view)
n = 4096
if os.environ.get("KM_PHASE_INFINITE_TABLE"):        # A/B: the round-2 infinite table plane (what does the rectangle change in this scenario?)
    from gym_kmanip_amd.model import compile_model, ENV_SPECS
    _cm = compile_model(ENV_SPECS[env_id], solver=solver)
    _cm.desc.table_rect[0] = -np.inf; _cm.desc.table_rect[1] = np.inf; _cm.desc.table_rect[2] = -np.inf; _cm.desc.table_rect[3] = np.inf
    env = env_hip.KManipEnvHip(_cm, num_envs=n, seed=0)
else:
    env = env_hip.make(env_id, num_envs=n, seed=0, solver=solver)
import numpy as _np
env.k_reset(); env.set_state(step=(_np.arange(n) % 64).astype(_np.int32))   # desynchronised episode phases (as bench.py)
L = env.L
gen = torch.Generator(device="cuda"); gen.manual_seed(0)
acts = [(torch.rand((n, env.cm.act_dim), generator=gen, device="cuda") * 2 - 1) for _ in range(8)]
for k in range(80): env.step_flat(acts[k % 8])
torch.cuda.synchronize()
buf = (C.c_ulonglong * NPH)()
prof = L.kmanip_dbg_prof if env.cm.nlink == 10 else L.kmanip_dbg_prof20
epb = 4 if env.cm.nlink == 10 else 2
prof(buf, 1)
steps = 32
for k in range(steps): env.step_flat(acts[k % 8])
torch.cuda.synchronize()
prof(buf, 0)
v = np.array(list(buf), dtype=np.float64)
nblocks = n // epb
per = v / nblocks / steps            # cycles (100 MHz memtime ticks?) per block per control step
tot = per[:41].sum()       # (slots 41.. are event counters and the iteration stopwatch, not phases)
print("solver", solver, "total ticks per block per step %.0f" % tot)
for nm, x in zip(names, per):
    if nm.startswith("COUNT") or nm.startswith("CYCLES"):
        print("  %-26s %10.4f" % (nm, x))
    else:
        print("  %-26s %10.0f  %5.1f%%" % (nm, x, 100 * x / tot))

# ---- per-workgroup, per-lane-group view of the LAST launch: the kernel ends when its slowest wave does, and a wave is as
# slow as its slowest env in every phase.  Slot 13 ("integrate") of a group also holds its wait for the sibling envs.
nb = n // 4
blk = (C.c_ulonglong * (NPH * 4 * nb))()
if env.cm.nlink == 10 and hasattr(L, "kmanip_dbg_prof_blocks") and L.kmanip_dbg_prof_blocks(blk, nb) == 0:
    B = np.array(list(blk), dtype=np.float64).reshape(nb, 4, NPH)
    if os.environ.get("KM_PHASE_DUMP"):
        np.save(os.environ["KM_PHASE_DUMP"], B)
    tot_b = B[:, 0, :41].sum(1)                       # wave lifetime as seen by lane group 0
    order = np.argsort(tot_b)
    print("last launch: wave totals  mean %.0f  p50 %.0f  p90 %.0f  p99 %.0f  max %.0f" % (
        tot_b.mean(), np.median(tot_b), np.percentile(tot_b, 90), np.percentile(tot_b, 99), tot_b.max()))
    newton = list(range(9, 27))
    work = B.copy(); work[:, :, 28] = 0; work[:, :, 31] = 0; work[:, :, 40:] = 0     # drop the wait / tail slots: a group's OWN work
    own = work.sum(2)                                            # [nb, 4]
    slow_g = own.argmax(1)
    slow = order[-10:]
    print("  phase                      mean group   busiest group of the 10 slowest waves")
    for i, nm in enumerate(names):
        print("  %-26s %10.0f %14.0f" % (nm, B[:, :, i].mean(), np.mean([B[wv, slow_g[wv], i] for wv in slow])))
    print("  own work of that group: %.0f of the wave's %.0f cycles; Newton phases %.0f" % (
        np.mean([own[wv, slow_g[wv]] for wv in slow]), tot_b[slow].mean(), np.mean([B[wv, slow_g[wv], newton].sum() for wv in slow])))


# ---- the slowest waves one by one: which phase classes make them slow (summed over the wave's 4 groups where it is the max)
if "B" in dir():
    ik = [30, 33, 34, 35, 36, 37]; allp = list(range(9, 15)); armp = list(range(15, 21)); cubep = list(range(21, 27))
    fixed = [0, 1, 2, 3, 4, 5, 7, 8, 27, 38, 39]
    print("slowest waves: total | per-group max of: IK, ALL loop, ARM loop, CUBE loop, fixed per-sub-step work, waiting for the mates' loops")
    for wv in order[-16:][::-1]:
        g = B[wv]
        print("  wave %4d  %8.0f | IK %7.0f  ALL %7.0f  ARM %7.0f  CUBE %7.0f  fixed %7.0f  wait %7.0f" % (
            wv, tot_b[wv], g[:, ik].sum(1).max(), g[:, allp].sum(1).max(), g[:, armp].sum(1).min(), g[:, cubep].sum(1).max(), g[:, fixed].sum(1).max(), g[:, 40].max()))
    print("ALL-loop slots of the ALL-heaviest group of the 6 waves with the longest ALL loops: H build | chol | tri-solve | ls setup | ls loop | eval")
    allsum = B[:, :, allp].sum(2).max(1)
    for wv in np.argsort(allsum)[-6:][::-1]:
        g = B[wv]; gi = g[:, allp].sum(1).argmax()
        print("  wave %4d group %d  " % (wv, gi) + "  ".join("%8.0f" % g[gi, i] for i in allp) + "   (wave total %.0f)" % tot_b[wv])
    print("IK slots of the IK-heaviest group of the 4 slowest waves: before_step | res+jac | normal matrix | TR solve | select_step | ratio/tests")
    for wv in order[-4:][::-1]:
        g = B[wv]; gi = g[:, ik].sum(1).argmax()
        print("  wave %4d group %d  " % (wv, gi) + "  ".join("%8.0f" % g[gi, i] for i in ik))
