"""The pin for MuJoCo's mj_step under the path: the oracle (CPU) and the HIP engine (-m gpu) against
tests/golden/mj_<asset>.npz, files that ONLY a box with the real `mujoco` package can make:

    python tests/tools/make_golden_mujoco.py

No such box exists on this pool, so the two pin tests skip -- with that command in the reason -- until the files exist; the
generator's plumbing and the checker itself are exercised below against an oracle-backed stand-in (engine "fake": refused as a pin).

Tolerances when the files are real (float64 both sides; what differs is stated):
  qM, qfrc_bias         1e-9 relative to the largest entry: same rigid-body algorithm, different operation order
  efc_J / efc_R / efc_aref   rows matched as a set (contact order is the engine's business), 1e-7 / 1e-6 relative
  qacc_smooth           1e-8 relative;  qacc: 1e-5 relative -- two Newton solvers stopped by tolerance 1e-8 on different iterates
  control step          qpos 1e-6, qvel 1e-4 (ten sub-steps of those qacc differences), contact masks and ctrl exact
"""
import glob
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN
from gym_kmanip_amd.model import compile_model

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
import mujoco_pin as MP  # noqa: E402

HOWTO = "tests/golden/mj_*.npz absent: MuJoCo's mj_step stays UNPINNED until `python tests/tools/make_golden_mujoco.py` runs on a box with the `mujoco` package"
REAL = {"M": 1e-9, "rows": 1e-7, "aref": 1e-6, "smooth": 1e-8, "qacc": 1e-5, "qpos": 1e-6, "qvel": 1e-4}
SELF = {"M": 1e-13, "rows": 1e-12, "aref": 1e-12, "smooth": 1e-12, "qacc": 1e-12, "qpos": 1e-13, "qvel": 1e-12}


def fixtures():
    return sorted(glob.glob(os.path.join(GOLDEN, "mj_*.npz")))


def load(path, allow_fake=False):
    f = np.load(path)
    meta = MP.read_meta(f)
    if meta["engine"] != "mujoco" and not allow_fake:
        pytest.fail("%s was made with engine %r: only a file made by the real mujoco package is a pin" % (path, meta["engine"]))
    return f, meta


def rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return float(np.abs(a - b).max() / max(1e-300, np.abs(b).max())) if a.size else 0.0


def check_oracle(f, meta, tol):
    """Every state of a fixture against the oracle; returns the worst deviations (for the log)."""
    from oracle.oracle import Oracle
    cm = compile_model(MP.qpos_spec(meta["asset"]), auto_reset=False)
    assert list(cm.spec.act_list) == meta["act_list"]
    o = Oracle(cm, 1)
    worst = dict.fromkeys(("M", "bias", "smooth", "qacc", "R", "aref", "qpos", "qvel"), 0.0)
    for i in range(meta["nstate"]):
        qpos, qvel, ctrl = f["qpos"][i], f["qvel"][i], f["ctrl"][i]
        r = o.dynamics(qpos, qvel, ctrl)
        worst["M"] = max(worst["M"], rel(r["M"], f["qM"][i])); worst["bias"] = max(worst["bias"], rel(r["bias"], f["qfrc_bias"][i]))
        worst["smooth"] = max(worst["smooth"], rel(r["qacc_smooth"], f["qacc_smooth"][i])); worst["qacc"] = max(worst["qacc"], rel(r["qacc"], f["qacc"][i]))
        ne = int(f["nefc"][i])
        assert r["nefc"] == ne, (i, r["nefc"], ne)
        perm = MP.match_rows(f["efc_J"][i][:ne], r["J"], tol["rows"])
        assert perm is not None, "state %d: the oracle's constraint rows are not MuJoCo's (as a set)" % i
        worst["R"] = max(worst["R"], rel(r["R"][perm], f["efc_R"][i][:ne])); worst["aref"] = max(worst["aref"], rel(r["aref"][perm], f["efc_aref"][i][:ne]))
        t, _ = o.constraint_rows(qpos, qvel)
        assert np.array_equal(t[perm] == 0, f["efc_type"][i][:ne] == MP.MJ_CNSTR_FRICTION_DOF), i      # friction-loss rows are the two-sided ones
        assert o.contact_mask(qpos)[0] == int(f["mask"][i]), (i, hex(o.contact_mask(qpos)[0]), hex(int(f["mask"][i])))
        # the control step: Physics.step(10) in the legacy order from the ctrl before_step set
        q, v, w, bad, mask, _, _ = o.physics_step(qpos, qvel, f["ctrl_set"][i], f["warm"][i], qpos, meta["n_sub"])
        assert not bad and mask == int(f["post_mask"][i]), (i, hex(mask), hex(int(f["post_mask"][i])))
        worst["qpos"] = max(worst["qpos"], float(np.abs(q - f["post_qpos"][i]).max())); worst["qvel"] = max(worst["qvel"], float(np.abs(v - f["post_qvel"][i]).max()))
    assert worst["M"] < tol["M"] and worst["bias"] < tol["M"], worst
    assert worst["smooth"] < tol["smooth"] and worst["qacc"] < tol["qacc"], worst
    assert worst["R"] < tol["rows"] and worst["aref"] < tol["aref"], worst
    assert worst["qpos"] < tol["qpos"] and worst["qvel"] < tol["qvel"], worst
    return worst


def check_hip(f, meta, tol):
    """The control steps of a fixture through the C ABI: one env per state, set_state -> kmanip_step -> get_state."""
    import torch
    from gym_kmanip_amd import env_hip
    cm = compile_model(MP.qpos_spec(meta["asset"]), auto_reset=False)
    S = meta["nstate"]
    dev = env_hip.KManipEnvHip(cm, num_envs=S, device=0, seed=0)
    dev.k_reset()
    dev.set_state(qpos=f["qpos"], qvel=f["qvel"], ctrl=f["ctrl"], warm=f["warm"], step=np.zeros(S, dtype=np.int32))
    dev.step_flat(torch.from_numpy(np.ascontiguousarray(f["action"], dtype=np.float32)).cuda())
    st = dev.get_state()
    assert np.array_equal(st[2], f["ctrl_set"])                                   # before_step's ctrl (float32-quantised): exact
    assert np.array_equal(dev.get_diag()[0], f["post_mask"].astype(np.uint32))    # contact masks of the trailing mj_step1: exact
    dq, dv = float(np.abs(st[0] - f["post_qpos"]).max()), float(np.abs(st[1] - f["post_qvel"]).max())
    dev.k_close()
    assert dq < tol["qpos"] and dv < tol["qvel"], (dq, dv)
    return dq, dv


@pytest.mark.parametrize("asset", MP.ASSETS)
def test_oracle_vs_mujoco(asset):
    path = os.path.join(GOLDEN, "mj_%s.npz" % asset)
    if not os.path.exists(path):
        pytest.skip(HOWTO)
    f, meta = load(path)
    print(check_oracle(f, meta, REAL))


@pytest.mark.gpu
@pytest.mark.parametrize("asset", MP.ASSETS)
def test_hip_vs_mujoco(asset):
    path = os.path.join(GOLDEN, "mj_%s.npz" % asset)
    if not os.path.exists(path):
        pytest.skip(HOWTO)
    f, meta = load(path)
    print(check_hip(f, meta, REAL))


def test_no_fake_fixture_is_committed_as_a_pin():
    for p in fixtures():
        assert MP.read_meta(np.load(p))["engine"] == "mujoco", p


# ------------------------------------------------------------------------------------------------------------------
# plumbing: the generator and the checker above, end to end, with an oracle-backed stand-in for the mujoco module
@pytest.fixture(scope="module")
def fake_fixture(tmp_path_factory):
    import fake_mujoco
    import make_golden_mujoco as G
    out = G.gen(fake_mujoco, "solo_arm", n_states=9, seed=3, engine="fake")
    path = os.path.join(str(tmp_path_factory.mktemp("mj")), "mj_solo_arm.npz")
    np.savez_compressed(path, **out)
    return path


def test_generator_plumbing_with_a_fake_mujoco(fake_fixture):
    f, meta = load(fake_fixture, allow_fake=True)
    cm = compile_model(MP.qpos_spec("solo_arm"))
    S = meta["nstate"]
    assert meta["engine"] == "fake" and S == 9 and meta["act_list"] == ["q_pos_r", "grip_r"] and meta["n_sub"] == 10
    assert f["qpos"].shape == (S, cm.nq) and f["qM"].shape == (S, cm.nv, cm.nv) and f["efc_J"].shape[0] == S and f["efc_J"].shape[2] == cm.nv
    assert f["action"].dtype == np.float32 and f["action"].shape == (S, cm.act_dim) and f["mask"].dtype == np.uint32
    assert (f["nefc"] > 0).all() and (f["mask"] & 0xFF).any()              # friction-loss rows everywhere; a cube on the table somewhere
    assert not np.array_equal(f["qpos"], f["post_qpos"])
    # the checker on an oracle-made file: the generator's sequence `mj_step1; ctrl; mj_step2; 9 x mj_step; mj_step1` is the oracle's
    # Physics.step(10), so everything agrees to roundoff -- the step ORDER, the row matching and the mask map are what is tested
    worst = check_oracle(f, meta, SELF)
    assert worst["qpos"] < 1e-13
    with pytest.raises(pytest.fail.Exception, match="only a file made by the real mujoco"):
        load(fake_fixture)


def test_contact_list_to_mask_map():
    names = ["finger_a", "finger_b", "palm", "wrist", "forearm", "elbow"]
    I = np.eye(3)
    m = MP.contacts_to_mask([("table", "cube", [0.1 - 0.02, 0.6 + 0.02, 0.5]), ("cube", "table", [0.1 + 0.02, 0.6 - 0.02, 0.5]),
                             ("finger_b", "cube", [0, 0, 0]), ("forearm__capsule", "cube", [0, 0, 0]), ("table", "elbow__seg", [0, 0, 0])],
                            cube_pos=[0.1, 0.6, 0.52], cube_mat=I, sphere_names=names)
    assert m == (1 << 0b010) | (1 << 0b001) | (1 << (8 + 1)) | (1 << (8 + 4)) | (1 << (20 + 5))
    with pytest.raises(ValueError):
        MP.contacts_to_mask([("finger_a", "palm", [0, 0, 0])], [0, 0, 0], I, names)
    J = np.array([[1.0, 0, 0], [0, 2.0, 0], [0, 0, 3.0]])
    assert np.array_equal(MP.match_rows(J, J[[2, 0, 1]]), [1, 2, 0]) and MP.match_rows(J, J[:2]) is None
    assert MP.match_rows(J, J + np.array([[0, 0, 0], [0, 0, 0], [0, 1e-3, 0]])) is None


@pytest.mark.gpu
def test_hip_checker_on_an_oracle_made_file(fake_fixture):
    """The -m gpu leg's checker on the fake file: HIP vs the ORACLE's control steps (an oracle-vs-HIP test, not a pin) -- so that
    the first real mj_*.npz meets a checker that has run."""
    f, meta = load(fake_fixture, allow_fake=True)
    dq, dv = check_hip(f, meta, dict(qpos=1e-9, qvel=1e-7))
    print(dq, dv)
