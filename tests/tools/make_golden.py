#!/usr/bin/env python3
"""Generate tests/golden/*.npz (small fixtures = inputs + expected outputs).

Sources of truth, per file:
  ik_scipy_<env>.npz   REAL scipy.optimize.least_squares + scipy Rotation on the NumPy restatement
                       of ik_mujoco.py (oracle/ik_scipy.py)  -> pins the C oracle and the HIP IK.
  euler_goal.npz       scipy Rotation euler decode (env_sim.py:66-69).
  fk_<env>.npz         NumPy FK from the asset JSON (home pose anchors == SURVEY.md A.4).
  traj_<env>.npz       C oracle 64-step trajectories with a seeded action stream -> pins the HIP
                       step (oracle-vs-oracle regression on CPU).
  philox_kat.npz       published Random123 philox4x32-10 known-answer vectors.
Run in the build container: python tests/tools/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)

from gym_kmanip_amd.model import compile_model  # noqa: E402
from oracle import ik_scipy as S  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
ARMS = {"KManipSoloArm": [(0, "eer_site_pos")], "KManipDualArm": [(0, "eer_site_pos"), (1, "eel_site_pos")],
        "KManipTorso": [(0, "eer_site_pos"), (1, "eel_site_pos")]}


def ranges(cm):
    return np.array([l["joint"]["range"] for l in cm.asset["links"]], dtype=float)


def home(cm):
    return np.array([cm.desc.q_home[i] for i in range(cm.nlink)])


def gen_ik(env, n_cases=48, seed=0):
    cm = compile_model(env)
    rng = np.random.default_rng(seed)
    arm = S.NumpyArm(cm.asset)
    rg, hm = ranges(cm), home(cm)
    rec = dict(arm=[], qpos=[], action=[], goal_pos=[], goal_quat=[], q_out=[], qpos_after=[], nfev=[], status=[],
               res0=[], jac0=[])
    for t in range(n_cases):
        ai, site = ARMS[env][t % len(ARMS[env])]
        n = cm.desc.arm_nq[ai]
        mask = np.array(list(cm.desc.arm_q_id[ai])[:n])
        qpos = np.zeros(cm.nq)
        qpos[:cm.nlink] = np.clip(hm + rng.normal(0, 0.3, cm.nlink) * (t > 1), rg[:, 0] + 1e-3, rg[:, 1] - 1e-3)
        if t % 8 == 3:
            qpos[mask[1]] = rg[mask[1], 0]          # exactly on a bound
        if t % 8 == 5:
            qpos[mask[2]] = rg[mask[2], 1] + 1e-3   # infeasible start -> "IK failed" branch
        qpos[cm.nlink:cm.nlink + 3] = [0.2, 0.5, 0.65]
        qpos[cm.nlink + 3] = 1
        xp, xq, ax = arm.fk(qpos)
        p, mat = arm.site(site, xp, xq)
        a = rng.uniform(-1, 1, 6).astype(np.float32)
        gp = p + a[:3].astype(float) * 0.01
        gq = S.euler_goal(mat, a[3:].astype(float) * 0.1)
        ph = S.FakePhysics(arm, qpos, rg)
        r0 = S.ik_res(qpos[mask].copy(), physics=S.FakePhysics(arm, qpos, rg), goal_pos=gp, goal_orn=gq, q_mask=mask,
                      q_pos_home=hm[mask], q_pos_prev=qpos[mask], ee_site=site)
        j0 = S.ik_jac(qpos[mask].copy(), physics=S.FakePhysics(arm, qpos, rg), goal_orn=gq, q_mask=mask, ee_site=site)
        q, res = S.ik(ph, gp, gq, mask, hm, qpos.copy(), site)
        pad = lambda v, k: np.pad(np.asarray(v, dtype=float).ravel(), (0, k - np.size(v)))
        rec["arm"].append(ai); rec["qpos"].append(qpos); rec["action"].append(a)
        rec["goal_pos"].append(gp); rec["goal_quat"].append(gq)
        rec["q_out"].append(pad(q, 7)); rec["qpos_after"].append(ph.qpos.copy())
        rec["nfev"].append(res.nfev if res is not None else 0)
        rec["status"].append(res.status if res is not None else -2)
        rec["res0"].append(pad(r0, 20)); rec["jac0"].append(pad(j0, 140))
    np.savez_compressed(os.path.join(OUT, "ik_scipy_%s.npz" % env), **{k: np.array(v) for k, v in rec.items()})


def gen_euler(seed=1):
    rng = np.random.default_rng(seed)
    from scipy.spatial.transform import Rotation as R
    mats = R.random(64, random_state=2).as_matrix()
    deltas = rng.uniform(-0.1, 0.1, (64, 3))
    out = np.array([S.euler_goal(m, d) for m, d in zip(mats, deltas)])
    np.savez_compressed(os.path.join(OUT, "euler_goal.npz"), mat=mats, delta=deltas, quat=out)


def gen_fk(env, seed=3):
    cm = compile_model(env)
    rng = np.random.default_rng(seed)
    arm = S.NumpyArm(cm.asset)
    rg, hm = ranges(cm), home(cm)
    Q, XP, XQ, SP, SM = [], [], [], [], []
    for t in range(16):
        q = hm.copy() if t == 0 else rng.uniform(rg[:, 0], rg[:, 1])
        xp, xq, ax = arm.fk(q)
        sp = np.zeros((2, 3)); sm = np.zeros((2, 3, 3))
        for ai, site in ARMS[env]:
            sp[ai], sm[ai] = arm.site(site, xp, xq)
        Q.append(q); XP.append(xp); XQ.append(xq); SP.append(sp); SM.append(sm)
    np.savez_compressed(os.path.join(OUT, "fk_%s.npz" % env), q=np.array(Q), xpos=np.array(XP), xquat=np.array(XQ),
                        site_pos=np.array(SP), site_mat=np.array(SM))


def gen_traj(env, n_envs=4, seed=7, steps=66):
    """steps > 64 so that the auto-reset boundary is inside the fixture."""
    cm = compile_model(env, auto_reset=True)
    o = Oracle(cm, n_envs, seed=seed, env_id_offset=100)
    rng = np.random.default_rng(seed)
    obs0 = o.reset()
    rec = dict(obs0=obs0, act=[], obs=[], rew=[], done=[], qpos=[], qvel=[], ctrl=[], warm=[], mask=[], nfev=[],
               status=[])
    s0 = o.get_state()
    rec["qpos0"], rec["qvel0"], rec["ctrl0"], rec["warm0"] = s0[0], s0[1], s0[2], s0[3]
    for k in range(steps):
        act = rng.uniform(-1, 1, (n_envs, cm.act_dim)).astype(np.float32)
        if k < 4:
            act[0] = 0
        obs, rew, done = o.step(act)
        qp, qv, ct, wm, st = o.get_state()
        m, nf, stt = o.get_diag()
        for key, v in zip(["act", "obs", "rew", "done", "qpos", "qvel", "ctrl", "warm", "mask", "nfev", "status"],
                          [act, obs, rew, done, qp, qv, ct, wm, m, nf, stt]):
            rec[key].append(v)
    np.savez_compressed(os.path.join(OUT, "traj_%s.npz" % env), seed=seed, env_id_offset=100,
                        **{k: np.array(v) for k, v in rec.items()})


def gen_philox():
    # Random123 kat_vectors: philox4x32 10 rounds (counter[4], key[2]) -> output[4]
    kat = np.array([
        [0, 0, 0, 0, 0, 0, 0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8],
        [0xffffffff] * 6 + [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd],
        [0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344, 0xa4093822, 0x299f31d0,
         0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1],
    ], dtype=np.uint32)
    np.savez_compressed(os.path.join(OUT, "philox_kat.npz"), kat=kat)


def main():
    os.makedirs(OUT, exist_ok=True)
    gen_philox()
    gen_euler()
    for env in ARMS:
        gen_fk(env)
        gen_ik(env)
        gen_traj(env)
        print("golden:", env)


if __name__ == "__main__":
    main()
