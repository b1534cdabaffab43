"""Oracle pinning: the C restatement of ik_mujoco.py + SciPy TRF against (a) fixtures produced by the
real scipy.optimize.least_squares (tests/golden/ik_scipy_*.npz, tests/tools/make_golden.py) and (b) a live
SciPy run on fresh seeds.  CPU only."""
import os

import numpy as np
import pytest

from conftest import ENVS3, GOLDEN
from gym_kmanip_amd.model import compile_model
from oracle import ik_scipy as S
from oracle.oracle import Oracle

IK_TOL = 1e-6   # rad; nfev may differ by one evaluation at a termination knife-edge


@pytest.mark.parametrize("env", ENVS3)
def test_fk_golden(env):
    g = np.load(os.path.join(GOLDEN, "fk_%s.npz" % env))
    cm = compile_model(env)
    o = Oracle(cm, 1)
    for q, xp, xq, sp, sm in zip(g["q"], g["xpos"], g["xquat"], g["site_pos"], g["site_mat"]):
        qpos = np.zeros(cm.nq); qpos[:cm.nlink] = q; qpos[cm.nlink + 3] = 1
        xpos, xquat, spos, smat = o.fk(qpos)
        assert np.abs(xpos - xp).max() < 1e-12
        assert np.abs(xquat - xq).max() < 1e-12
        for a in range(2):
            if cm.desc.arm_present[a]:
                assert np.abs(spos[a] - sp[a]).max() < 1e-12 and np.abs(smat[a] - sm[a]).max() < 1e-12


def test_home_pose_anchors():
    """SURVEY.md A.4 anchors (FK of the reference XML at the home pose)."""
    exp = {"KManipSoloArm": ([0.2577, 0.4994, 0.6264], None),
           "KManipDualArm": ([0.2579, 0.4990, 0.6265], [-0.1724, 0.5777, 0.6578]),
           "KManipTorso": ([0.1835, 0.4186, 0.5554], [-0.1773, 0.4093, 0.5258])}
    for env, (r, l) in exp.items():
        cm = compile_model(env)
        qpos = np.zeros(cm.nq); qpos[:cm.nlink] = cm.spec.q_pos_home; qpos[cm.nlink + 3] = 1
        _, _, sp, sm = Oracle(cm, 1).fk(qpos)
        assert np.abs(sp[0] - r).max() < 1e-4
        if l is not None:
            assert np.abs(sp[1] - l).max() < 1e-4
    q = S.mju_mat2quat(sm[0])  # last = torso; solo quaternion anchor checked below
    cm = compile_model("KManipSoloArm")
    qpos = np.zeros(cm.nq); qpos[:cm.nlink] = cm.spec.q_pos_home; qpos[cm.nlink + 3] = 1
    _, _, sp, sm = Oracle(cm, 1).fk(qpos)
    assert np.abs(S.mju_mat2quat(sm[0]) - [0.9293, -0.1011, -0.0839, -0.3451]).max() < 1e-4


def test_euler_goal_vs_scipy_rotation():
    g = np.load(os.path.join(GOLDEN, "euler_goal.npz"))
    o = Oracle(compile_model("KManipSoloArm"), 1)
    for m, d, q in zip(g["mat"], g["delta"], g["quat"]):
        qc = o.euler_goal(m, d)
        assert min(np.abs(qc - q).max(), np.abs(qc + q).max()) < 1e-12


@pytest.mark.parametrize("env", ENVS3)
def test_ik_golden_scipy(env):
    g = np.load(os.path.join(GOLDEN, "ik_scipy_%s.npz" % env))
    cm = compile_model(env)
    o = Oracle(cm, 1)
    nfev_mismatch = 0
    for i in range(len(g["arm"])):
        arm = int(g["arm"][i]); n = cm.desc.arm_nq[arm]
        mask = np.array(list(cm.desc.arm_q_id[arm])[:n])
        qpos = g["qpos"][i]
        f = o.ik_res(arm, qpos, qpos[mask], qpos, g["goal_pos"][i], g["goal_quat"][i])
        J = o.ik_jac(arm, qpos, qpos[mask], qpos, g["goal_pos"][i], g["goal_quat"][i])
        assert np.abs(f - g["res0"][i][:6 + 2 * n]).max() < 1e-13
        assert np.abs(J.ravel() - g["jac0"][i][:(6 + 2 * n) * n]).max() < 1e-13
        q, qp_after, nfev, st = o.ik(arm, qpos, g["goal_pos"][i], g["goal_quat"][i])
        assert np.abs(q - g["q_out"][i][:n]).max() < IK_TOL
        assert np.abs(qp_after - g["qpos_after"][i]).max() < IK_TOL
        nfev_mismatch += int(nfev != g["nfev"][i] or st != g["status"][i])
        if g["status"][i] == -2:   # "IK failed" branch: x0 outside bounds, nothing evaluated
            assert nfev == 0 and st == -2 and np.array_equal(qp_after, qpos)
    assert nfev_mismatch <= 2, nfev_mismatch


@pytest.mark.parametrize("env", ["KManipSoloArm", "KManipTorso"])
def test_ik_live_scipy(env):
    """Fresh seeds against SciPy itself (not a fixture)."""
    cm = compile_model(env)
    o = Oracle(cm, 1)
    arm = S.NumpyArm(cm.asset)
    rg = np.array([l["joint"]["range"] for l in cm.asset["links"]], dtype=float)
    hm = np.array([cm.desc.q_home[i] for i in range(cm.nlink)])
    rng = np.random.default_rng(1234)
    n = cm.desc.arm_nq[0]; mask = np.array(list(cm.desc.arm_q_id[0])[:n])
    for t in range(12):
        qpos = np.zeros(cm.nq)
        qpos[:cm.nlink] = np.clip(hm + rng.normal(0, 0.4, cm.nlink), rg[:, 0] + 1e-3, rg[:, 1] - 1e-3)
        qpos[cm.nlink + 3] = 1
        xp, xq, _ = arm.fk(qpos)
        p, mat = arm.site("eer_site_pos", xp, xq)
        a = rng.uniform(-1, 1, 6)
        gp = p + a[:3] * 0.01; gq = S.euler_goal(mat, a[3:] * 0.1)
        ph = S.FakePhysics(arm, qpos, rg)
        qs, res = S.ik(ph, gp, gq, mask, hm, qpos.copy(), "eer_site_pos")
        qc, qp_after, nfev, st = o.ik(0, qpos, gp, gq)
        assert np.abs(qs - qc).max() < IK_TOL
        assert np.abs(ph.qpos - qp_after).max() < IK_TOL
