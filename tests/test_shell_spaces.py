"""Host logic of the caller-side shell (gym_kmanip_amd/gym_shell.py) without a GPU: the Dict spaces every env id
declares must be the reference's (gym_kmanip/env_base.py:115-190 with the kwargs of gym_kmanip/__init__.py:244-483),
and their insertion order must be the flat action / observation column order the C ABI uses (include/kmanip.h)."""
from collections import OrderedDict

import numpy as np
import pytest

from gym_kmanip_amd import gym_shell
from gym_kmanip_amd.model import ENV_SPECS, compile_model

# (obs keys -> width), (action keys -> width); arm widths: 7 joints per K-Scale arm, 6 per torso arm; two finger sliders per grip
EXPECT = {
    "KManipSoloArm":     (dict(q_pos=10, q_vel=10, cube_pos=3, cube_orn=4), dict(eer_pos=3, eer_orn=3, grip_r=1)),
    "KManipSoloArmQPos": (dict(q_pos=10, q_vel=10, cube_pos=3, cube_orn=4), dict(grip_r=1, q_pos_r=7)),
    "KManipDualArm":     (dict(q_pos=20, q_vel=20, cube_pos=3, cube_orn=4), dict(eel_pos=3, eel_orn=3, eer_pos=3, eer_orn=3, grip_l=1, grip_r=1)),
    "KManipDualArmQPos": (dict(q_pos=20, q_vel=20, cube_pos=3, cube_orn=4), dict(grip_l=1, grip_r=1, q_pos_r=7, q_pos_l=7)),
    "KManipTorso":       (dict(q_pos=20, q_vel=20, cube_pos=3, cube_orn=4), dict(eel_pos=3, eel_orn=3, eer_pos=3, eer_orn=3, grip_l=1, grip_r=1)),
}


class _StubBackend:
    def __init__(self, gym_env, **kw):
        self.kw = kw


@pytest.fixture()
def stub_backend(monkeypatch):
    monkeypatch.setattr(gym_shell.env_hip, "new", lambda gym_env, **kw: _StubBackend(gym_env, **kw))


@pytest.mark.parametrize("env_id", sorted(ENV_SPECS))
def test_spaces_match_reference_declarations(env_id, stub_backend):
    env = gym_shell.KManipEnv(env_id, num_envs=3)
    obs_e, act_e = EXPECT[env_id.replace("Vision", "")]
    cams_e = {}
    if env_id.endswith("Vision"):
        # the *Vision ids trade cube_pos / cube_orn for camera images (__init__.py:306-311,366-372,429-435): uint8 RGB Boxes of
        # the cameras' reference resolutions (env_base.py:140-146; head 480x640, grippers 40x60: __init__.py:157-161)
        obs_e = dict(q_pos=obs_e["q_pos"], q_vel=obs_e["q_vel"])
        names = ["head", "grip_r"] if "Solo" in env_id else ["head", "grip_l", "grip_r"]
        cams_e = {"camera/" + n: ((480, 640, 3) if n == "head" else (40, 60, 3)) for n in names}
        assert [c.name for c in env.cameras] == names and env.info["cameras"] is env.cameras
    obs_s, act_s = env.observation_space.spaces, env.action_space.spaces
    assert list(obs_s.keys()) == list(obs_e.keys()) + list(cams_e.keys())
    assert list(act_s.keys()) == list(act_e.keys())
    for k, w in obs_e.items():
        assert obs_s[k].shape == (w,) and obs_s[k].dtype == np.float64            # OBS_DTYPE, __init__.py:50
        assert float(obs_s[k].low.min()) == -1.0 and float(obs_s[k].high.max()) == 1.0
    for k, shp in cams_e.items():
        assert obs_s[k].shape == shp and obs_s[k].dtype == np.uint8 and int(obs_s[k].low.min()) == 0 and int(obs_s[k].high.max()) == 255
    for k, w in act_e.items():
        assert act_s[k].shape == (w,) and act_s[k].dtype == np.float32            # ACT_DTYPE, __init__.py:51
    # backend seam: the shell asks for no auto-reset (the TimeLimit wrapper owns truncation, as in the reference)
    assert env.env.kw["auto_reset"] is False and env.env.kw["num_envs"] == 3
    # the attributes env_sim.new reads from the gym env (env_sim.py:26-27,45,50-51,76-77,112,154,208-209)
    for attr in ["mjcf_filename", "seed", "q_len", "q_pos_home", "q_id_r_mask", "q_id_l_mask", "ctrl_id_r_grip",
                 "ctrl_id_l_grip", "obs_list", "act_list", "cameras"]:
        assert hasattr(env, attr), attr
    assert env.q_len == obs_e["q_pos"] and env.info["a_len"] == len(act_e)
    # sampled actions are members of the space
    a = env.action_space.sample()
    assert env.action_space.contains(a)


@pytest.mark.parametrize("env_id", sorted(EXPECT))
def test_flat_layout_is_dict_insertion_order(env_id, stub_backend):
    """Column k of the flat [num_envs, act_dim] / [num_envs, obs_dim] device tensors = the k-th scalar of the Dict
    spaces in insertion order -- the layout include/kmanip.h documents and compile_model() encodes."""
    env = gym_shell.KManipEnv(env_id, num_envs=1)
    cm = compile_model(env_id)
    col = 0
    for k, sp in env.action_space.spaces.items():
        sl = cm.act_slices[k]
        assert (sl.start, sl.stop) == (col, col + sp.shape[0]), k
        col = sl.stop
    assert col == cm.act_dim
    col = 0
    for k, sp in env.observation_space.spaces.items():
        sl = cm.obs_slices[k]
        assert (sl.start, sl.stop) == (col, col + sp.shape[0]), k
        col = sl.stop
    assert col == cm.obs_dim


def test_package_entry_points_and_gymnasium_registration():
    """gym_kmanip/__init__.py:244-483 registers eight ids with max_episode_steps = 64; register_envs() does the same when a
    gymnasium module is there (a stand-in here: gymnasium is absent from the image)."""
    import types
    import gym_kmanip_amd as k
    calls = []
    fake = types.SimpleNamespace(register=lambda **kw: calls.append(kw))
    ids = k.register_envs(fake, suffix="-HIP")
    assert ids == [i + "-HIP" for i in k.ENV_IDS] and len(ids) == 8
    assert {c["id"] for c in calls} == set(ids)
    for c in calls:
        assert c["max_episode_steps"] == 64 and c["nondeterministic"] is True
        assert c["entry_point"] == "gym_kmanip_amd.gym_shell:KManipEnv"
        assert c["kwargs"]["env_id"] + "-HIP" == c["id"] and c["kwargs"]["squeeze"] is True
    assert callable(k.make) and callable(k.make_backend)
