"""The C oracle, the NumPy IK restatement and the host-side model constants against fixtures made by running the REFERENCE'S
OWN PYTHON (tests/golden/ref_*.npz, ref_spaces.json; generator: tests/tools/make_golden_ref.py over tests/tools/refrun.py).

These pin what the reference itself wrote -- grip / EE-delta / joint-delta decode with its NumPy float32/float64 promotions,
the ik()/ik_res/ik_jac call into the real scipy least_squares, the clips, get_observation, get_reward (incl. the touch / lift
branch with rigged geom names), initialize_episode, the k_reset / k_step tuple plumbing of KManipEnv, the kwargs and spaces of
the eight registered ids -- to the reference (b).  MuJoCo's mj_step under the fixtures is the
oracle's own restatement (nothing external exists for it), so agreement of the physics columns here is a consistency check
of the harness, not a pin.

Tolerances: the oracle restates SciPy's TRF with a Jacobi SVD, so its IK differs from the real least_squares at the 1e-9 rad
level: 1e-6 rad on qpos (the bar of every IK comparison in this suite), 1e-5 on qvel, float32-quantised ctrl bit-exact unless a
float64 difference straddles a float32 rounding boundary (1 ulp, counted); masks, nfev / status, done: exact."""
import json
import os

import numpy as np
import pytest

from conftest import ENVS3, GOLDEN
from gym_kmanip_amd import model as M
from gym_kmanip_amd.model import compile_model

FAMILY = {"KManipSoloArm": "solo", "KManipDualArm": "dual", "KManipTorso": "torso"}
RUN_IDS = ["KManipSoloArm", "KManipSoloArmQPos", "KManipSoloArmVision", "KManipDualArm", "KManipDualArmQPos",
           "KManipDualArmVision", "KManipTorso", "KManipTorsoVision"]


def _ref(name):
    return np.load(os.path.join(GOLDEN, name))


def obs_columns(cm):
    """Columns of the engine's flat state observation that an id's obs_list holds (the *Vision ids drop the cube keys)."""
    keys = [k for k in ("q_pos", "q_vel", "cube_pos", "cube_orn") if k in cm.spec.obs_list]
    return np.concatenate([np.arange(cm.obs_dim)[cm.obs_slices[k]] for k in keys])


@pytest.mark.parametrize("env", ENVS3)
def test_numpy_ik_restatement_is_bitwise_the_reference(env):
    """ik_scipy_<env>.npz (made by oracle/ik_scipy.py, the fixtures the C oracle and the HIP IK have been tested against since
    round 1) equals, bit for bit, what the reference's own ik / ik_res / ik_jac return on the same cases."""
    a, b = _ref("ref_ik_%s.npz" % FAMILY[env]), _ref("ik_scipy_%s.npz" % env)
    assert sorted(a.files) == sorted(b.files)
    for k in a.files:
        assert np.array_equal(a[k], b[k]), k
    assert set(a["status"].tolist()) >= {-2, 1, 2}          # the "IK failed" branch and two termination kinds are in the set


@pytest.mark.parametrize("env", ENVS3)
def test_oracle_ik_vs_reference(env):
    from oracle.oracle import Oracle
    g = _ref("ref_ik_%s.npz" % FAMILY[env])
    cm = compile_model(env)
    o = Oracle(cm, 1)
    nf_mismatch = 0
    for i in range(len(g["arm"])):
        arm = int(g["arm"][i]); n = cm.desc.arm_nq[arm]
        q, qp, nfev, st = o.ik(arm, g["qpos"][i], g["goal_pos"][i], g["goal_quat"][i])
        assert np.abs(q - g["q_out"][i][:n]).max() < 1e-6 and np.abs(qp - g["qpos_after"][i]).max() < 1e-6
        assert st == g["status"][i] or g["status"][i] > 0 and st > 0
        nf_mismatch += int(nfev != g["nfev"][i])
        mask = [cm.desc.arm_q_id[arm][k] for k in range(n)]
        x0 = g["qpos"][i][mask]
        m = 6 + 2 * n
        assert np.abs(o.ik_res(arm, g["qpos"][i], x0, g["qpos"][i], g["goal_pos"][i], g["goal_quat"][i]) - g["res0"][i][:m]).max() < 1e-12
        assert np.abs(o.ik_jac(arm, g["qpos"][i], x0, g["qpos"][i], g["goal_pos"][i], g["goal_quat"][i]).ravel() - g["jac0"][i][:m * n]).max() < 1e-12
    assert nf_mismatch <= 2, nf_mismatch


@pytest.mark.parametrize("env_id", RUN_IDS)
def test_oracle_step_vs_reference_run(env_id):
    """Every step of the reference's KManipEnv rollout as a one-step problem: state before -> ko_step -> what the reference's
    before_step left (ctrl, teleported qpos, nfev / status) and what its k_step returned (obs, reward, terminated, sim_time)."""
    from oracle.oracle import Oracle
    r = _ref("ref_run_%s.npz" % env_id)
    cm = compile_model(env_id, auto_reset=False)
    T = len(r["action"])
    o = Oracle(cm, T)
    o.set_state(r["pre_qpos"], r["pre_qvel"], r["pre_ctrl"], r["pre_warm"], r["pre_step"])
    obs, rew, done = o.step(r["action"])
    q, v, c, w, s = o.get_state()
    mask, nfev, st = o.get_diag()
    flips = c != r["ctrl_set"]
    if flips.any():
        ulp = np.spacing(np.abs(r["ctrl_set"][flips]).astype(np.float32)).astype(np.float64)
        assert (np.abs(c[flips] - r["ctrl_set"][flips]) <= ulp).all() and flips.sum() <= 2
    assert r["ctrl_set"].dtype == np.float64 and np.array_equal(r["ctrl_set"], r["ctrl_set"].astype(np.float32))   # float32 values
    assert np.abs(q - r["post_qpos"]).max() < 1e-6 and np.abs(v - r["post_qvel"]).max() < 1e-5
    cols = obs_columns(cm)
    assert np.abs(obs[:, cols] - r["obs"]).max() < 1e-6 and np.abs(rew - r["reward"]).max() < 1e-6
    assert np.array_equal(mask, r["contact_mask"])
    assert np.array_equal(st, r["ik_status"]) and np.abs(nfev - r["ik_nfev"]).max() <= 1
    # k_step's tuple: terminated is always False (no termination in the task), TimeLimit is gymnasium's wrapper
    assert not r["terminated"].any() and np.array_equal(done & 1, (r["pre_step"] + 1 >= cm.desc.max_episode_steps).astype(np.uint8))
    assert np.allclose(r["sim_time"], (r["pre_step"] + 1) * M.CONTROL_TIMESTEP, rtol=0, atol=1e-12)
    assert np.array_equal(r["info_step"], r["pre_step"] + 1)
    assert np.array_equal(r["is_success"], r["reward"] > M.REWARD_SUCCESS_THRESHOLD)


@pytest.mark.parametrize("env_id", ["KManipSoloArm", "KManipDualArm", "KManipTorso"])
def test_oracle_teleport_vs_reference_run(env_id):
    """The IK leaves qpos[arm] at its LAST EVALUATED point (ik_mujoco.py:34,67) -- the reference's data.qpos when physics.step
    is entered -- and the mocap goal pose is the decoded target (env_sim.py:62-69): through the oracle's standalone IK."""
    from oracle.oracle import Oracle
    r = _ref("ref_run_%s.npz" % env_id)
    cm = compile_model(env_id)
    o = Oracle(cm, 1)
    for t in range(0, len(r["action"]), 7):
        qp = r["pre_qpos"][t].copy()
        for arm in range(2):
            if not cm.desc.arm_present[arm]:
                continue
            # mocap ids: right = 0, left = 1 (__init__.py:139-140)
            q, qp, nfev, st = o.ik(arm, qp, r["mocap_pos"][t][arm], r["mocap_quat"][t][arm])
        assert np.abs(qp - r["qpos_teleport"][t]).max() < 1e-6, t


@pytest.mark.parametrize("env_id", RUN_IDS)
def test_reset_vs_reference_run(env_id):
    """KManipTask.initialize_episode (env_sim.py:23-36) + dm_control's reset: home pose in qpos AND ctrl, zero velocity, cube
    inside CUBE_SPAWN_RANGE at its qpos0 orientation; the first observation is get_observation of that state."""
    from oracle.oracle import Oracle
    r = _ref("ref_run_%s.npz" % env_id)
    cm = compile_model(env_id)
    nl = cm.nlink
    home = np.array([cm.desc.q_home[i] for i in range(nl)])
    o = Oracle(cm, 1)
    o.reset()
    q0 = o.get_state()[0][0]
    for e in range(len(r["reset_qpos"])):
        q = r["reset_qpos"][e]
        assert np.array_equal(q[:nl], home) and np.array_equal(r["reset_ctrl"][e], home) and not r["reset_qvel"][e].any()
        assert np.array_equal(q[nl + 3:], q0[nl + 3:])
        assert (q[nl:nl + 3] >= M.CUBE_SPAWN_RANGE[:, 0]).all() and (q[nl:nl + 3] <= M.CUBE_SPAWN_RANGE[:, 1]).all()
        obs, _ = o.observe(q, r["reset_qvel"][e])
        assert np.abs(obs[obs_columns(cm)] - r["reset_obs"][e]).max() < 1e-12
        assert np.abs(o.after_reset(q, r["reset_qvel"][e], r["reset_ctrl"][e]) - r["reset_warm"][e]).max() < 1e-9
        assert r["reset_sim_time"][e] == 0.0
    assert (q0[nl:nl + 3] >= M.CUBE_SPAWN_RANGE[:, 0]).all() and (q0[nl:nl + 3] <= M.CUBE_SPAWN_RANGE[:, 1]).all()


@pytest.mark.parametrize("env", ENVS3)
def test_oracle_obs_reward_vs_reference(env):
    from oracle.oracle import Oracle
    g = _ref("ref_obs_%s.npz" % FAMILY[env])
    cm = compile_model(env)
    o = Oracle(cm, 1)
    clipped = 0
    for i in range(len(g["qpos"])):
        obs, rew = o.observe(g["qpos"][i], g["qvel"][i])
        assert np.abs(obs - g["obs"][i]).max() < 1e-12 and abs(rew - g["reward"][i]) < 1e-12
        assert o.contact_mask(g["qpos"][i])[0] == g["contact_mask"][i]
        clipped += int((np.abs(g["obs"][i][:2 * cm.nlink + 3]) == 1.0).sum())
    assert clipped > 10                      # the fixture does exercise the clips


@pytest.mark.parametrize("env", ENVS3)
def test_touch_lift_reward_vs_reference(env):
    """get_reward's touch / lift terms (env_sim.py:164-178) fire in the reference only if finger geoms carry the names that code
    looks for; the fixture rigs them, the build's `touch_reward` flag stands for the same thing."""
    from oracle.oracle import Oracle
    g = _ref("ref_touch_%s.npz" % FAMILY[env])
    cm = compile_model(env, touch_reward=True)
    o = Oracle(cm, 1)
    kinds = set()
    for i in range(len(g["qpos"])):
        obs, rew = o.observe(g["qpos"][i], g["qvel"][i])
        assert abs(rew - g["reward"][i]) < 1e-12, i
        kinds.add(int(round(g["reward"][i] - (g["reward"][i] % 1.0))))
    assert kinds == {0, 1, 2}                # no touch / touch / touch + lift all occur
    off = Oracle(compile_model(env, touch_reward=False), 1)      # the reference as shipped: the terms are dead
    assert all(off.observe(g["qpos"][i], g["qvel"][i])[1] < 1.0 for i in range(len(g["qpos"])))


def test_model_constants_and_specs_vs_reference_registration():
    with open(os.path.join(GOLDEN, "ref_spaces.json")) as f:
        ref = json.load(f)
    c = ref["constants"]
    for name in ("MAX_EPISODE_STEPS", "CONTROL_TIMESTEP", "MAX_Q_VEL", "CTRL_ALPHA", "IK_RES_RAD", "IK_RES_REG_PREV",
                 "IK_RES_REG_HOME", "IK_JAC_RAD", "IK_JAC_REG", "EPSILON", "Q_POS_DELTA", "EE_S_MIN", "EE_S_MAX", "EE_S_DELTA",
                 "REWARD_SUCCESS_THRESHOLD", "REWARD_VEL_PENALTY", "REWARD_GRIP_DIST", "REWARD_TOUCH_CUBE", "REWARD_LIFT_CUBE"):
        assert getattr(M, name) == c[name], name
    assert np.array_equal(M.CUBE_SPAWN_RANGE, np.array(c["CUBE_SPAWN_RANGE"]))
    assert list(M.EE_POS_DELTA) == c["EE_POS_DELTA"] and list(M.EE_ORN_DELTA) == c["EE_ORN_DELTA"]
    assert c["OBS_DTYPE"] == "float64" and c["ACT_DTYPE"] == "float32"
    assert sorted(ref["envs"]) == sorted(M.ENV_SPECS)
    for env_id, e in ref["envs"].items():
        spec = M.ENV_SPECS[env_id]
        cm = compile_model(env_id)
        assert spec.obs_list == e["obs_list"] and spec.act_list == e["act_list"] and spec.max_episode_steps == e["max_episode_steps"]
        assert cm.asset["source"] == e["mjcf_filename"] and cm.nlink == e["q_len"]
        assert e["q_pos_home_dtype"] == "float32" and np.array_equal(np.asarray(spec.q_pos_home, dtype=np.float64), np.array(e["q_pos_home"]))
        for a, b in ((spec.q_id_r_mask, e["q_id_r_mask"]), (spec.q_id_l_mask, e["q_id_l_mask"]),
                     (spec.ctrl_id_r_grip, e["ctrl_id_r_grip"]), (spec.ctrl_id_l_grip, e["ctrl_id_l_grip"])):
            assert (a is None and b is None) or list(a) == list(b)
        # flat action layout = the action Dict's insertion order and widths (env_base.py:151-188)
        col = 0
        for key, sp in e["action_space"].items():
            assert cm.act_slices[key] == slice(col, col + sp["shape"][0]), (env_id, key)
            assert sp["dtype"] == "float32" and sp["low"] == -1 and sp["high"] == 1
            col += sp["shape"][0]
        assert col == cm.act_dim and len(e["action_space"]) == e["action_len"]
        col = 0
        for key, sp in e["observation_space"].items():
            if key.startswith("camera/"):
                cam = M.CAMERAS[key.split("/")[1]]
                assert sp["shape"] == [cam.h, cam.w, 3] and sp["dtype"] == "uint8" and (sp["low"], sp["high"]) == (0, 255)
            else:
                assert sp["dtype"] == "float64" and sp["low"] == -1 and sp["high"] == 1
                assert cm.obs_slices[key].stop - cm.obs_slices[key].start == sp["shape"][0]
        assert cm.cameras == [c_["name"] for c_ in e["cameras"]]
        for c_ in e["cameras"]:
            cam = M.CAMERAS[c_["name"]]
            assert (cam.w, cam.h, cam.c, cam.fl, list(cam.pp), cam.log_name) == (c_["w"], c_["h"], c_["c"], c_["fl"], c_["pp"], c_["log_name"])


def test_shell_info_and_spaces_vs_reference():
    """The Gymnasium-shaped shell (gym_kmanip_amd/gym_shell.py) offers the info keys and space structure of the reference's
    KManipEnv (env_base.py:117-211) -- checked without constructing an engine (no GPU here): on the class's static tables."""
    from gym_kmanip_amd import gym_shell
    with open(os.path.join(GOLDEN, "ref_spaces.json")) as f:
        ref = json.load(f)
    for env_id, e in ref["envs"].items():
        sp = gym_shell.spaces_for(env_id)
        assert list(sp["observation"]) == list(e["observation_space"])
        assert list(sp["action"]) == list(e["action_space"])
        for k, v in e["observation_space"].items():
            assert list(sp["observation"][k].shape) == v["shape"] and str(np.dtype(sp["observation"][k].dtype)) == v["dtype"]
        for k, v in e["action_space"].items():
            assert list(sp["action"][k].shape) == v["shape"] and str(np.dtype(sp["action"][k].dtype)) == v["dtype"]
        assert set(e["info_keys"]) <= set(gym_shell.INFO_KEYS)
        assert gym_shell.q_keys_for(env_id) == e["q_keys"]
        assert gym_shell.KManipEnv.metadata == e["metadata"]


@pytest.mark.parametrize("env_id", ["KManipSoloArm", "KManipDualArm", "KManipTorso"])
def test_oracle_scripted_policy_vs_reference_heuristic(env_id):
    """ko_scripted_eer_pos against the heuristic of examples/2_synthetic_data.py:28-41 as evaluated on the reference env object
    (ref_scripted_<id>.npz: `raw_action`, float64), at the 64 states of the reference's own scripted episode; then every step
    of that episode as a one-step problem through ko_step (the reference handed before_step a float64 eer_pos: the flat row's
    float32 rounding of it moves the IK goal by < 1e-9 m)."""
    from oracle.oracle import Oracle
    r = _ref("ref_scripted_%s.npz" % env_id)
    cm = compile_model(env_id, auto_reset=False)
    T = len(r["action"])
    o = Oracle(cm, T)
    for t in range(T):
        assert np.abs(o.scripted_eer_pos(r["pre_qpos"][t]) - r["raw_action"][t]).max() < 1e-12
    sl = cm.act_slices["eer_pos"]
    assert np.array_equal(r["action"][:, sl], r["raw_action"].astype(np.float32))
    o.set_state(r["pre_qpos"], r["pre_qvel"], r["pre_ctrl"], r["pre_warm"], np.arange(T, dtype=np.int32))
    obs, rew, done = o.step(r["action"])
    q, v = o.get_state()[:2]
    # IK-limited bars: two TRF runs on inputs that differ in the 8th digit may stop an iteration apart (ftol = xtol = gtol = 1e-8),
    # up to 1e-6 rad on the teleported joints (the bar of every IK comparison in this suite); a position offset d in a kp = 1000,
    # I = 0.01 servo joint becomes a velocity of up to d * sqrt(kp / I) = 316 d within the control step (explicit Euler at
    # omega dt = 0.63 overshoots a little: 400 d) -- qvel bar = 1e-6 rad x 400 / s; the median sample sits two orders below
    dq, dv = np.abs(q - r["post_qpos"]).max(1), np.abs(v - r["post_qvel"]).max(1)
    assert dq.max() < 1e-6 and dv.max() < 4e-4 and np.median(dv) < 2e-6, (dq.max(), dv.max(), np.median(dv))
    assert np.abs(obs[:, obs_columns(cm)] - r["obs"]).max() < 1e-5 and np.abs(rew - r["reward"]).max() < 1e-6
    assert np.array_equal(r["info_step"], np.arange(1, T + 1)) and not r["terminated"].any() and not r["is_success"].any()
    # the rigged success state (env_base.py:250): the oracle's reward crosses the reference's threshold where the reference's did
    o1 = Oracle(cm, 1)
    o1.set_state(r["success_qpos"][None], r["success_qvel"][None], r["success_ctrl"][None], r["success_warm"][None], np.zeros(1, dtype=np.int32))
    _, rew1, _ = o1.step(r["success_action"][None])
    assert rew1[0] > M.REWARD_SUCCESS_THRESHOLD == float(r["success_threshold"]) and abs(rew1[0] - float(r["success_reward"])) < 1e-4 * rew1[0]


@pytest.mark.skipif(not os.path.isdir("/root/reference/gym_kmanip"), reason="build container only: needs the reference checkout")
def test_committed_fixtures_are_what_the_reference_produces_today(tmp_path, monkeypatch):
    """Staleness guard (build container): regenerate a slice with the reference's Python and compare with the committed files."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
    import make_golden_ref as G
    monkeypatch.setattr(G, "OUT", str(tmp_path))
    fresh = G.gen_ik("KManipSoloArm", n_cases=12)
    old = _ref("ref_ik_solo.npz")
    for k in fresh:
        assert np.array_equal(fresh[k], old[k][:12]), k
    fresh = G.gen_run("KManipSoloArmQPos", [6], seed=101)
    old = _ref("ref_run_KManipSoloArmQPos.npz")
    for k in ("action", "ctrl_set", "post_qpos", "obs", "reward"):
        assert np.array_equal(fresh[k], old[k][:6]), k
    # round 5: the reference's own logger tree and its scripted episode, regenerated (3 steps) against the committed 64-step run
    tree, fresh = G.gen_h5_scripted("KManipSoloArm", nstep=3)
    old_tree = json.load(open(os.path.join(GOLDEN, "ref_h5_tree_KManipSoloArm.json")))
    assert tree["tree"] == old_tree["tree"] and tree["file"] == old_tree["file"]
    old = _ref("ref_scripted_KManipSoloArm.npz")
    for k in ("raw_action", "action", "post_qpos", "reward", "h5/action"):
        assert np.array_equal(fresh[k][:3], old[k][:3]), k
    assert np.array_equal(fresh["success_reward"], old["success_reward"])
