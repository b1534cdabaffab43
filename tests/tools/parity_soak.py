#!/usr/bin/env python3
"""Diagnostic (GPU box): wide GPU-vs-oracle parity run.  Thousands of envs, several episodes, random actions; the oracle is
re-synchronised to the device state after every control step, so each step is an independent one-step parity sample
(free-running comparisons measure chaos, not the kernel).  Prints, per env id: the worst one-step deviation of qpos / qvel /
reward over the envs whose float32 ctrl came out identical, the number of contact-mask, done-byte and float32-ctrl mismatches, and which contact bits were seen.
Usage: python tests/tools/parity_soak.py [num_envs] [steps]"""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import torch
from gym_kmanip_amd import env_hip
from gym_kmanip_amd.model import compile_model
from oracle.oracle import Oracle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 130
only = sys.argv[3] if len(sys.argv) > 3 else None           # optional: one env id
for env in ["KManipSoloArm", "KManipDualArm", "KManipTorso"]:
    if only and env != only:
        continue
    cm = compile_model(env, auto_reset=True)
    dev = env_hip.KManipEnvHip(cm, num_envs=n, seed=11, env_id_offset=3); orc = Oracle(cm, n, seed=11, env_id_offset=3)
    dev.k_reset(); orc.reset()
    stagger = (np.arange(n) % 64).astype(np.int32)
    dev.set_state(step=stagger); orc.set_state(*dev.get_state())
    rng = None
    worst = dict(q=0.0, v=0.0, r=0.0)
    n_mask = n_done = n_ctrl = n_nfev = 0
    n_q10 = n_q9 = 0                              # one-step samples (float32 ctrl agreeing) with |dq| above 1e-10 / 1e-9: how FAT is the tail behind `worst`?
    seen = 0
    t0 = time.time()
    for k in range(steps):
        a = dev.sample_action()                   # bench.py's counter-based action stream
        act = a.cpu().numpy()
        assert np.array_equal(act, orc.sample_action())
        pre = orc.get_state()
        dev.step_flat(a)
        oo, ro, do = orc.step(act, nthreads=16)
        sg, so = dev.get_state(), orc.get_state()
        ok = ~(sg[2] != so[2]).any(axis=1)       # a float32 rounding flip of ctrl (1 ulp) legitimately moves that env's step by ~1e-5
        dvk = np.where(ok, np.abs(sg[1] - so[1]).max(axis=1), 0.0)
        if dvk.max() > worst["v"] and os.environ.get("KM_SOAK_DUMP"):
            e = int(dvk.argmax())                 # the worst one-step sample so far: its inputs and both outputs, for offline analysis
            np.savez(os.path.join(os.environ["KM_SOAK_DUMP"], "worst_%s.npz" % env), step=k, env=e, act=act[e], **{"pre%d" % i: pre[i][e] for i in range(5)},
                     **{"dev%d" % i: sg[i][e] for i in range(5)}, **{"orc%d" % i: so[i][e] for i in range(5)})
        worst["q"] = max(worst["q"], float(np.abs(sg[0] - so[0])[ok].max())); worst["v"] = max(worst["v"], float(np.abs(sg[1] - so[1])[ok].max()))
        worst["r"] = max(worst["r"], float(np.abs(dev.reward.cpu().numpy() - ro)[ok].max()))
        dqk = np.where(ok, np.abs(sg[0] - so[0]).max(axis=1), 0.0)
        n_q10 += int((dqk > 1e-10).sum()); n_q9 += int((dqk > 1e-9).sum())
        mg, nfg, stg = dev.get_diag(); mo, nfo, sto = orc.get_diag()
        n_mask += int((mg != mo).sum()); n_done += int((dev.done.cpu().numpy() != do).sum()); n_ctrl += int((sg[2] != so[2]).any(axis=1).sum())
        n_nfev += int((np.abs(nfg - nfo) > 1).sum())
        seen |= int(np.bitwise_or.reduce(mg))
        orc.set_state(*sg)                       # one-step samples
        orc.set_episode(dev.get_episode()) if hasattr(orc, "set_episode") else None
    print("%-14s %d envs x %d steps (%.0f s): worst one-step |dq| %.2e |dv| %.2e |dr| %.2e (envs whose float32 ctrl agrees); mismatches: mask %d done %d ctrl(f32) %d nfev(>1) %d; samples with |dq| > 1e-10: %d, > 1e-9: %d of %d; bits seen %#x" % (
        env, n, steps, time.time() - t0, worst["q"], worst["v"], worst["r"], n_mask, n_done, n_ctrl, n_nfev, n_q10, n_q9, n * steps, seen), flush=True)
    dev.k_close()
