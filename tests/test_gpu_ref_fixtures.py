"""The HIP path (through the C ABI) against fixtures made by running the REFERENCE'S OWN PYTHON in the build container
(tests/golden/ref_*.npz: tests/tools/make_golden_ref.py over tests/tools/refrun.py; only the .npz data travels to the GPU box).

What the fixtures pin to the reference itself -- not to the builder's reading of it -- is everything gym-kmanip's own files
compute: before_step's grip / EE-delta / joint-delta decode incl. NumPy's float32/float64 promotions, ik()/ik_res/ik_jac over
the real scipy.optimize.least_squares, get_observation, get_reward (incl. the touch / lift branch), initialize_episode and the
k_step tuple.  MuJoCo's mj_step under them is the oracle's restatement.

Tolerances as in test_gpu_parity.py: IK 1e-6 rad vs SciPy; one control step from the fixture's state: qpos 1e-6 (IK-limited),
qvel 1e-5, obs / reward 1e-6; ctrl float32-quantised and bit-exact up to a counted 1-ulp straddle; masks / status exact."""
import os

import numpy as np
import pytest

from conftest import ENVS3, GOLDEN
from gym_kmanip_amd.model import compile_model
from test_ref_fixtures import FAMILY, RUN_IDS, obs_columns

pytestmark = pytest.mark.gpu


def _dev(env_id, n, **kw):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from gym_kmanip_amd import env_hip
    cm = compile_model(env_id, **kw)
    return cm, env_hip.KManipEnvHip(cm, num_envs=n, seed=0), torch


def _ref(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.mark.parametrize("env_id", RUN_IDS)
def test_hip_step_vs_reference_run(env_id):
    r = _ref("ref_run_%s.npz" % env_id)
    T = len(r["action"])
    cm, dev, torch = _dev(env_id, T, auto_reset=False)
    dev.set_state(r["pre_qpos"], r["pre_qvel"], r["pre_ctrl"], r["pre_warm"], r["pre_step"])
    dev.step_flat(torch.from_numpy(r["action"]).cuda())
    q, v, c, w, s = dev.get_state()
    mask, nfev, st = dev.get_diag()
    obs, rew, done = dev.obs.cpu().numpy(), dev.reward.cpu().numpy(), dev.done.cpu().numpy()
    flips = c != r["ctrl_set"]
    if flips.any():
        ulp = np.spacing(np.abs(r["ctrl_set"][flips]).astype(np.float32)).astype(np.float64)
        assert (np.abs(c[flips] - r["ctrl_set"][flips]) <= ulp).all() and flips.sum() <= 2, flips.sum()
    assert np.abs(q - r["post_qpos"]).max() < 1e-6 and np.abs(v - r["post_qvel"]).max() < 1e-5
    assert np.abs(obs[:, obs_columns(cm)] - r["obs"]).max() < 1e-6 and np.abs(rew - r["reward"]).max() < 1e-6
    assert np.array_equal(mask, r["contact_mask"])
    ran = r["ik_status"] != -3                      # (-3 = no ik() call for that arm: the device leaves its diagnostics alone)
    assert np.array_equal(st[ran], r["ik_status"][ran]) and (np.abs(nfev - r["ik_nfev"])[ran] <= 1).all()
    assert np.array_equal(done & 1, (r["pre_step"] + 1 >= cm.desc.max_episode_steps).astype(np.uint8)) and not (done & 2).any()
    assert np.array_equal(s, r["pre_step"] + 1)
    for name, (h, wd) in (("grip_r", (40, 60)), ("grip_l", (40, 60))):       # the *Vision ids' small camera observations
        if "img_" + name in r.files:
            img = dev.render_rgb(name, h, wd).cpu().numpy().astype(int)
            d = np.abs(img - r["img_" + name].astype(int))
            assert (d > 1).mean() < 2e-3, (name, (d > 1).mean())             # one grey level; silhouette-grazing rays excepted
    dev.k_close()


@pytest.mark.parametrize("env", ENVS3)
def test_hip_ik_vs_reference(env):
    g = _ref("ref_ik_%s.npz" % FAMILY[env])
    cm, dev, torch = _dev(env, 1)
    nf_mismatch = 0
    for arm in range(2):
        sel = np.where(g["arm"] == arm)[0]
        if len(sel) == 0:
            continue
        n = cm.desc.arm_nq[arm]
        m = 6 + 2 * n
        q, qp_after, nfev, st = dev.ik(arm, g["qpos"][sel], g["goal_pos"][sel], g["goal_quat"][sel])
        assert np.abs(q - g["q_out"][sel][:, :n]).max() < 1e-6 and np.abs(qp_after - g["qpos_after"][sel]).max() < 1e-6
        nf_mismatch += int((nfev != g["nfev"][sel]).sum())
        failed = g["status"][sel] == -2
        assert np.array_equal(st[failed], g["status"][sel][failed]) and (nfev[failed] == 0).all()
        res, jac = dev.ik_eval(arm, g["qpos"][sel], g["goal_pos"][sel], g["goal_quat"][sel])
        assert np.abs(res - g["res0"][sel][:, :m]).max() < 1e-12
        assert np.abs(jac.reshape(len(sel), -1) - g["jac0"][sel][:, :m * n]).max() < 1e-12
    assert nf_mismatch <= 3, nf_mismatch
    dev.k_close()


@pytest.mark.parametrize("env", ENVS3)
def test_hip_observe_vs_reference(env):
    """kmanip_observe (get_observation + get_reward of the current state) on the reference's seeded states: every clip."""
    g = _ref("ref_obs_%s.npz" % FAMILY[env])
    n = len(g["qpos"])
    cm, dev, torch = _dev(env, n)
    dev.set_state(g["qpos"], g["qvel"])
    obs, rew = dev.observe()
    assert np.abs(obs.cpu().numpy() - g["obs"]).max() < 1e-12
    assert np.abs(rew.cpu().numpy() - g["reward"]).max() < 1e-12
    assert np.array_equal(dev.get_diag()[0], g["contact_mask"])
    dev.k_close()


@pytest.mark.parametrize("env", ENVS3)
def test_hip_touch_lift_reward_vs_reference(env):
    g = _ref("ref_touch_%s.npz" % FAMILY[env])
    n = len(g["qpos"])
    cm, dev, torch = _dev(env, n, touch_reward=True)
    dev.set_state(g["qpos"], g["qvel"])
    _, rew = dev.observe()
    assert np.abs(rew.cpu().numpy() - g["reward"]).max() < 1e-12
    assert np.array_equal(dev.get_diag()[0], g["contact_mask"])
    assert {int(x) for x in np.floor(g["reward"])} == {0, 1, 2}
    dev.k_close()
    cm, dev, torch = _dev(env, n, touch_reward=False)                        # the reference as shipped: dead terms
    dev.set_state(g["qpos"], g["qvel"])
    assert (dev.observe()[1].cpu().numpy() < 1.0).all()
    dev.k_close()


@pytest.mark.parametrize("env_id", ["KManipSoloArm", "KManipDualArmQPos", "KManipTorso"])
def test_hip_reset_vs_reference_run(env_id):
    """kmanip_reset against initialize_episode as the reference ran it: home pose in qpos and ctrl, zero velocity, the cube
    at its qpos0 orientation inside the spawn box (its xyz comes from Philox here, from NumPy's global stream there:
    DESIGN.md section 4 deviation 2), and -- with the fixture's spawn put in -- the reference's first observation and warm start."""
    r = _ref("ref_run_%s.npz" % env_id)
    E = len(r["reset_qpos"])
    cm, dev, torch = _dev(env_id, E)
    dev.k_reset()
    q, v, c, w, s = dev.get_state()
    nl = cm.nlink
    assert np.array_equal(q[:, :nl], r["reset_qpos"][:, :nl]) and np.array_equal(c, r["reset_ctrl"]) and not v.any()
    assert np.array_equal(q[:, nl + 3:], r["reset_qpos"][:, nl + 3:])
    lo = np.array([cm.desc.cube_spawn_lo[k] for k in range(3)]); hi = np.array([cm.desc.cube_spawn_hi[k] for k in range(3)])
    assert (q[:, nl:nl + 3] >= lo).all() and (q[:, nl:nl + 3] <= hi).all()
    dev.set_state(r["reset_qpos"], r["reset_qvel"], r["reset_ctrl"])
    obs, _ = dev.observe()
    assert np.abs(obs.cpu().numpy()[:, obs_columns(cm)] - r["reset_obs"]).max() < 1e-12
    dev.k_close()


# --------------------------------------------------------------------------------------------------------------------
# Round 5: the reference's own logger (log_h5py.py run inside KManipEnv(log_h5py=True)), the scripted-data heuristic of
# examples/2_synthetic_data.py:28-41 evaluated on the reference env object, and info["is_success"] (env_base.py:250).
SCRIPTED_IDS = ["KManipSoloArm", "KManipDualArm", "KManipTorso"]


@pytest.mark.parametrize("env_id", SCRIPTED_IDS)
def test_hip_scripted_action_vs_reference_heuristic(env_id):
    """kmanip_scripted_action value for value: at each of the reference episode's 64 states the eer_pos columns must be the
    reference's `raw_action` (float64 there, float32 in the flat action row: one float32 ulp), every other column untouched."""
    r = _ref("ref_scripted_%s.npz" % env_id)
    T = len(r["action"])
    cm, dev, torch = _dev(env_id, T, auto_reset=False)
    dev.set_state(r["pre_qpos"], r["pre_qvel"], r["pre_ctrl"], r["pre_warm"])
    act = dev.scripted_action(torch.from_numpy(r["action_sampled"]).cuda()).cpu().numpy()
    sl = cm.act_slices["eer_pos"]
    ulp = np.spacing(np.abs(r["raw_action"]).astype(np.float32)).astype(np.float64)
    assert (np.abs(act[:, sl].astype(np.float64) - r["raw_action"]) <= ulp).all()
    other = np.ones(cm.act_dim, dtype=bool); other[sl] = False
    assert np.array_equal(act[:, other], r["action_sampled"][:, other])
    assert np.abs(np.linalg.norm(act[:, sl].astype(np.float64), axis=1) - 1).max() < 1e-6
    # ... and the step the reference then took from that state with that action (its float64 eer_pos rounded to the row's float32)
    dev.step_flat(torch.from_numpy(r["action"]).cuda())
    q, v = dev.get_state()[:2]
    # IK-limited bars (test_ref_fixtures.test_oracle_scripted_policy_vs_reference_heuristic): the reference's before_step saw the
    # float64 heuristic vector, the flat row holds its float32 rounding; qvel bar = the 1e-6 rad IK bar x 400 / s
    dq, dv = np.abs(q - r["post_qpos"]).max(1), np.abs(v - r["post_qvel"]).max(1)
    assert dq.max() < 1e-6 and dv.max() < 4e-4 and np.median(dv) < 2e-6, (dq.max(), dv.max(), np.median(dv))
    assert np.abs(dev.obs.cpu().numpy()[:, obs_columns(cm)] - r["obs"]).max() < 1e-5
    assert np.abs(dev.reward.cpu().numpy() - r["reward"]).max() < 1e-6
    dev.k_close()


@pytest.mark.parametrize("env_id", SCRIPTED_IDS)
def test_shell_is_success_vs_reference(env_id):
    """info["is_success"] = reward > REWARD_SUCCESS_THRESHOLD (env_base.py:250) on the rigged state the reference itself called a
    success (the cube at the right gripper site: 0.01 / (dist + 1e-6) > 2), and False on every step of its scripted episode."""
    from gym_kmanip_amd import gym_shell
    r = _ref("ref_scripted_%s.npz" % env_id)
    assert bool(r["success_is_success"]) and float(r["success_reward"]) > float(r["success_threshold"]) == 2.0
    env = gym_shell.KManipEnv(env_id, num_envs=2, seed=0)
    env.reset()
    st = lambda a: np.stack([a, a])
    env.env.set_state(st(r["success_qpos"]), st(r["success_qvel"]), st(r["success_ctrl"]), st(r["success_warm"]), np.zeros(2, dtype=np.int32))
    cm = env.env.cm
    action = {k: np.stack([r["success_action"][sl]] * 2) for k, sl in cm.act_slices.items()}
    obs, rew, term, trunc, info = env.step(action)
    assert np.abs(rew - float(r["success_reward"])).max() < 1e-4 * float(r["success_reward"])      # 1 / dist at dist ~ 1 mm: relative
    assert info["is_success"].all() and not term.any() and not trunc.any()
    # an ordinary state of the reference's episode: not a success there, not a success here
    k = 20
    env.env.set_state(st(r["pre_qpos"][k]), st(r["pre_qvel"][k]), st(r["pre_ctrl"][k]), st(r["pre_warm"][k]), np.zeros(2, dtype=np.int32))
    obs, rew, term, trunc, info = env.step({kk: np.stack([r["action"][k][sl]] * 2) for kk, sl in cm.act_slices.items()})
    assert not r["is_success"].any() and not info["is_success"].any() and np.abs(rew - r["reward"][k]).max() < 1e-6
    env.close()


@pytest.mark.parametrize("env_id", SCRIPTED_IDS + ["KManipSoloArmVision"])
def test_episode_logger_tree_and_data_vs_reference_logger(env_id, tmp_path, monkeypatch):
    """f-3 against the reference's OWN logger.  The reference episode of ref_scripted_<id>.npz is replayed on the device through the
    Gymnasium-shaped shell with log_h5py=True, log_reference_layout=True and the recording h5py stand-in the reference's log_h5py
    was run against (tests/tools/h5_recorder.py); the tree EpisodeLogger writes from its device rings must be the tree of
    tests/golden/ref_h5_tree_<id>.json node for node (groups, attrs incl. values, dataset shapes / dtypes / chunks) and its
    datasets must hold the reference file's numbers: `action` exactly (grip_r broadcast over a_len key columns),
    `observations/qpos|qvel` to the float32 of the 1e-6 one-step parity bar."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
    import h5_recorder as H5
    import json
    from gym_kmanip_amd import gym_shell
    from test_episode_log import assert_same_tree
    ref = json.load(open(os.path.join(GOLDEN, "ref_h5_tree_%s.json" % env_id)))
    r = _ref("ref_scripted_%s.npz" % env_id)
    T = len(r["action"])
    monkeypatch.setattr(gym_shell, "DATA_DIR", str(tmp_path))
    H5.FILES.clear()
    env = gym_shell.KManipEnv(env_id, num_envs=4, seed=0, log_h5py=True, log_prefix="sim_synth", log_backend="h5py",
                              log_h5py_module=H5, log_reference_layout=True, log_env_ids=[1])
    cm = env.env.cm
    env.reset()
    zero = np.zeros(4, dtype=np.int32)
    rep = lambda a: np.stack([a] * 4)
    env.env.set_state(rep(r["reset_qpos"]), rep(r["reset_qvel"]), rep(r["reset_ctrl"]), rep(r["reset_warm"]), zero)
    for t in range(T):
        # every step starts from the reference's own pre-step state (one-step comparison: the logged rows then differ from the
        # reference file's by the one-step parity bar, not by an episode's accumulated drift)
        env.env.set_state(rep(r["pre_qpos"][t]), rep(r["pre_qvel"][t]), rep(r["pre_ctrl"][t]), rep(r["pre_warm"][t]), zero + t)
        env.step({k: rep(r["action"][t][sl]) for k, sl in cm.act_slices.items()})
    env.close()
    (path, f), = H5.FILES.items()
    assert f.closed and os.path.basename(path) == "episode_1_env1.hdf5" and os.path.basename(os.path.dirname(path)).startswith(ref["log_dir_prefix"] + ".")
    ours = H5.tree(f, skip_attr_values=("cpu_time",))
    assert ours["groups"]["metadata"]["attrs"]["steps"]["value"] == T
    assert_same_tree(ours, ref["tree"], extra_attrs=("env", "steps"))
    data = H5.datasets(f)
    assert np.array_equal(data["action"][:T], r["h5/action"][:T]) and not data["action"][T:].any()
    for name in ("observations/qpos", "observations/qvel"):
        # (q_vel is the velocity / MAX_Q_VEL observation: the IK-limited velocity bar 400 x 1e-6 rad / pi, q_pos 1e-6 / range)
        assert data[name].dtype == np.float32 and np.abs(data[name][:T] - r["h5/" + name][:T]).max() < (2e-4 if name.endswith("qvel") else 2e-6), name
    for cam in env.cameras:
        img = data["observations/images/" + cam.name]
        assert img.shape == (64, cam.h, cam.w, 3) and img.dtype == np.uint8 and img[:T].any()
        if "h5/observations/images/" + cam.name in r.files:          # the small gripper frames: the surrogate scene, one grey level
            d = np.abs(img[:T].astype(int) - r["h5/observations/images/" + cam.name][:T].astype(int))
            assert (d > 1).mean() < 2e-3
