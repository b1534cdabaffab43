"""Caller side of the seam: a KManipEnv-compatible shell over the HIP backend.

Mirrors reference gym_kmanip/env_base.py:16-267 (constructor kwargs, Dict observation/action spaces with
the same keys, order, shapes, dtypes and bounds -- camera Boxes included --, the `info` dict, reset/step/render
signatures) and the TimeLimit(max_episode_steps=64) wrapper that `gym.make` adds (__init__.py:28,247).  `self.env` is
chosen exactly like env_base.py:192-200 does -- by calling a module-level `new(self)` -- here `env_hip.new`.
gymnasium is not installed in the build image, so spaces degrade to duck-typed shims (`contains`, `sample`,
`shape`, `dtype`, `low`, `high`); when gymnasium is importable its real spaces are used.
With num_envs == 1 and squeeze=True the return values have the reference's single-env shapes.
With device_outputs=True nothing crosses PCIe: observations, reward, terminated, truncated are device tensors and the
step never synchronises (the drop-in path at full speed); the default returns host NumPy like the reference.
ALIASING (device_outputs=True): the returned tensors are the backend's live device buffers -- the observation values are views of
its obs matrix -- and the next step overwrites them in place; clone() anything kept across steps (rollout lists, replay buffers).
log_h5py=True / log_prefix reproduce env_base.py:82-101,231-263: a fresh `<log_prefix>.<uuid6>.<date>` directory under DATA_DIR
and one episode file per reset (episode_log.EpisodeLogger: device-resident rings, written out at the next reset / close).
"""
from __future__ import annotations

import os
import time
import uuid
from collections import OrderedDict
from datetime import datetime
from typing import Any, Dict

import numpy as np

from . import env_hip
from .model import (CAMERAS, ENV_SPECS, MAX_EPISODE_STEPS, REWARD_SUCCESS_THRESHOLD, KM_DONE_DIVERGED, EnvSpec, load_asset)

OBS_DTYPE = np.float64   # __init__.py:50
ACT_DTYPE = np.float32   # __init__.py:51
DATA_DIR = os.environ.get("KMANIP_DATA_DIR", os.path.join(os.getcwd(), "data"))   # __init__.py:12 (theirs sits inside the package)
DATE_FORMAT = "%mm%dd%Yy_%Hh%Mm"                                                    # __init__.py:15

try:  # pragma: no cover - gymnasium is absent in the build image
    import gymnasium as _gym
    from gymnasium import spaces as _spaces
    _HAVE_GYM = True
    _EnvBase = _gym.Env                     # env_base.py:16: KManipEnv(gym.Env) -- what gym.make's wrappers expect
except Exception:  # noqa: BLE001
    _HAVE_GYM = False
    _EnvBase = object


class Box:
    def __init__(self, low, high, shape, dtype):
        self.low = np.full(shape, low, dtype=dtype); self.high = np.full(shape, high, dtype=dtype)
        self.shape = tuple(shape); self.dtype = np.dtype(dtype)
        self._rng = np.random.default_rng()

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and x.dtype == self.dtype and bool(np.all(x >= self.low) and np.all(x <= self.high))

    def sample(self):
        if np.issubdtype(self.dtype, np.integer):
            return self._rng.integers(self.low, self.high, endpoint=True, dtype=self.dtype)
        return self._rng.uniform(self.low, self.high).astype(self.dtype)

    def seed(self, seed=None):
        self._rng = np.random.default_rng(seed)


class DictSpace:
    def __init__(self, d):
        self.spaces = OrderedDict(d)

    def contains(self, x):
        return list(x.keys()) == list(self.spaces.keys()) and all(s.contains(x[k]) for k, s in self.spaces.items())

    def sample(self):
        return OrderedDict((k, s.sample()) for k, s in self.spaces.items())

    def seed(self, seed=None):
        for i, s in enumerate(self.spaces.values()):
            s.seed(None if seed is None else seed + i)

    def __getitem__(self, k):
        return self.spaces[k]


def _box(low, high, shape, dtype):
    return _spaces.Box(low=low, high=high, shape=shape, dtype=dtype) if _HAVE_GYM else Box(low, high, shape, dtype)


def _dict(d):
    return _spaces.Dict(d) if _HAVE_GYM else DictSpace(d)


# keys of KManipEnv.info: the constructor's (env_base.py:201-212) and the ones reset()/step() add (env_base.py:223-229,244-251)
INFO_KEYS = ("step", "episode", "is_success", "q_keys", "q_len", "a_len", "obs_list", "act_list", "cameras", "sim",
             "sim_time", "cpu_time", "reward", "terminated")


def spaces_for(env_id: str):
    """{"observation": OrderedDict, "action": OrderedDict} of Boxes for a registered id: the Dict spaces KManipEnv.__init__
    builds (env_base.py:115-190) -- same keys, insertion order (== flat column order of the action matrix), shapes, dtypes, bounds."""
    spec: EnvSpec = ENV_SPECS[env_id]
    q_len = len(spec.q_pos_home)
    od = OrderedDict()
    if "q_pos" in spec.obs_list:
        od["q_pos"] = _box(-1, 1, (q_len,), OBS_DTYPE)
    if "q_vel" in spec.obs_list:
        od["q_vel"] = _box(-1, 1, (q_len,), OBS_DTYPE)
    if "cube_pos" in spec.obs_list:
        od["cube_pos"] = _box(-1, 1, (3,), OBS_DTYPE)
    if "cube_orn" in spec.obs_list:
        od["cube_orn"] = _box(-1, 1, (4,), OBS_DTYPE)
    for o in spec.obs_list:                      # env_base.py:110-113,140-147: the Cam records of the "camera/<name>" keys
        if "camera" in o:
            cam = CAMERAS[o.split("/")[-1]]
            od[cam.log_name] = _box(cam.low, cam.high, (cam.h, cam.w, 3), cam.dtype)
    ad = OrderedDict()
    for key in ["eel_pos", "eel_orn", "eer_pos", "eer_orn"]:
        if key in spec.act_list:
            ad[key] = _box(-1, 1, (3,), ACT_DTYPE)
    for key in ["grip_l", "grip_r"]:
        if key in spec.act_list:
            ad[key] = _box(-1, 1, (1,), ACT_DTYPE)
    if "q_pos_r" in spec.act_list:
        ad["q_pos_r"] = _box(-1, 1, (len(spec.q_id_r_mask),), ACT_DTYPE)
    if "q_pos_l" in spec.act_list:
        ad["q_pos_l"] = _box(-1, 1, (len(spec.q_id_l_mask),), ACT_DTYPE)
    return {"observation": od, "action": ad}


def q_keys_for(env_id: str):
    """The joint-name list the reference registers as `q_keys` (keys of Q_*_HOME_DICT, __init__.py:53-122) = the MJCF joint
    names in qpos order -- except that the DualArm dict spells the left arm `joint_left_arm_1_*` where the XML says
    `joint_left_arm_2_*` (only teleop's URDF mapping reads these keys)."""
    spec: EnvSpec = ENV_SPECS[env_id]
    names = [l["joint"]["name"] for l in load_asset(spec.asset)["links"]]
    if spec.asset == "dual_arm":
        names = [n.replace("joint_left_arm_2_", "joint_left_arm_1_") for n in names]
    return names


class KManipEnv(_EnvBase):
    metadata = {"render_modes": ["rgb_array"], "render_fps": 30}

    def __init__(self, env_id: str = "KManipSoloArm", num_envs: int = 1, device: int = 0, seed: int = 0,
                 squeeze: bool = False, env_id_offset: int = 0, device_outputs: bool = False,
                 log_h5py: bool = False, log_prefix: str = "test", log_env_ids=None, log_backend=None,
                 log_reference_layout: bool = False, log_h5py_module=None, **overrides):
        spec: EnvSpec = ENV_SPECS[env_id]
        self.env_id = env_id
        self.seed = seed
        self.num_envs = num_envs
        self.squeeze = squeeze and num_envs == 1
        self.device_outputs = device_outputs
        assert not (self.squeeze and device_outputs), "squeeze returns host scalars; device_outputs returns device tensors"
        self.step_idx = 0
        self.episode_idx = 0
        # the attributes env_sim.new / env_hip.new read (env_base.py:61-78,110-115)
        self.mjcf_filename = {"solo_arm": "_env_solo_arm.xml", "dual_arm": "_env_dual_arm.xml", "torso": "_env_torso.xml"}[spec.asset]
        self.q_pos_home = spec.q_pos_home
        self.q_len = len(spec.q_pos_home)
        self.q_id_r_mask = spec.q_id_r_mask
        self.q_id_l_mask = spec.q_id_l_mask
        self.ctrl_id_r_grip = spec.ctrl_id_r_grip
        self.ctrl_id_l_grip = spec.ctrl_id_l_grip
        self.obs_list = [o for o in spec.obs_list]
        self.act_list = list(spec.act_list)
        # env_base.py:110-113: the Cam records of the "camera/<name>" observation keys
        self.cameras = [CAMERAS[o.split("/")[-1]] for o in self.obs_list if "camera" in o]
        # observation / action spaces, env_base.py:115-190
        sp = spaces_for(env_id)
        self.observation_space = _dict(sp["observation"])
        self.action_space = _dict(sp["action"])
        self.action_len = len(sp["action"])
        # joint names: teleop's URDF mapping reads them (env_base.py:66-68); part of `info`
        self.q_keys = q_keys_for(env_id)
        assert len(self.q_keys) == self.q_len, "q parameters do not match"          # env_base.py:68
        self.sim = True
        # backend seam, env_base.py:192-200
        self.env = env_hip.new(self, num_envs=num_envs, device=device, env_id_offset=env_id_offset,
                               auto_reset=False, **overrides)
        self.info: Dict[str, Any] = {
            "step": self.step_idx, "episode": self.episode_idx, "is_success": False, "q_keys": self.q_keys, "q_len": self.q_len,
            "a_len": self.action_len, "obs_list": self.obs_list, "act_list": self.act_list,
            "cameras": self.cameras, "sim": self.sim,
        }
        # optional episode logging, env_base.py:82-101 (rerun is a viewer, not on the path: not offered)
        self.log_h5py = log_h5py
        self.logger = None
        if log_h5py:
            self.log_dir = os.path.join(DATA_DIR, "{}.{}.{}".format(log_prefix, str(uuid.uuid4())[:6], datetime.now().strftime(DATE_FORMAT)))
            os.makedirs(self.log_dir, exist_ok=True)
            from .episode_log import EpisodeLogger
            # log_reference_layout: the reference logger's tree exactly (`action` = grip_r broadcast over a_len key columns,
            # `metadata` = the info dict at reset; episode_log.py, tests/golden/ref_h5_tree_*.json); default: the flat action row
            grip = self.env.cm.act_slices.get("grip_r")
            self.logger = EpisodeLogger(self.log_dir, num_envs, self.q_len, self.env.cm.act_dim, device=self.env.obs.device,
                                        env_ids=[0] if log_env_ids is None else log_env_ids, info=self.info, backend=log_backend,
                                        grip_r_col=None if grip is None else grip.start,
                                        reference_action_quirk=log_reference_layout, h5py_module=log_h5py_module)
            for cam in self.cameras:                                                 # env_base.py:233-234
                self.logger.cam(cam)

    def _log_step(self, action, obs_dev):
        """env_base.py:255-257: append this step's action / q_pos / q_vel (and camera frames) to the episode's rings."""
        act = self.env.last_act                                                 # the flat row k_step packed and ran on
        q = self.q_len
        frames = {cam.name: obs_dev[cam.log_name] for cam in self.cameras} or None
        self.logger.step(act, self.env.obs[:, :q], self.env.obs[:, q:2 * q], images=frames)

    def _log_flush(self):
        if self.logger is not None and self.logger.t > 0:
            self.log_paths = self.logger.end_episode()

    # ------------------------------------------------------------------ observations
    def _observation(self, state_obs):
        """state keys from the backend + the RGB render of every camera observation (env_sim.py:140-145; all of them in one
        launch: kmanip_render_rgb_multi), in obs_list order."""
        obs = OrderedDict((k, v) for k, v in state_obs.items() if k in self.obs_list)
        if self.cameras:
            imgs = self.env.render_cameras(self.cameras)
            for cam in self.cameras:
                obs[cam.log_name] = imgs[cam.name]
        self._obs_dev = obs
        if self.device_outputs:
            return obs
        out = OrderedDict()
        for k, v in obs.items():
            a = v.detach().cpu().numpy()
            if a.dtype != np.uint8:
                a = a.astype(OBS_DTYPE, copy=False)
            out[k] = a[0] if self.squeeze else a
        return out

    # ------------------------------------------------------------------ gym API
    def reset(self, seed=None, options=None):
        """env_base.py:219-239.  reset(seed=s) re-keys the cube-spawn stream and restarts its episode counter, so two resets
        with the same seed give the same first observation (the reference ignores the seed: its spawn draws from the global
        NumPy RNG, env_sim.py:34, and its ids are registered nondeterministic=True)."""
        if _HAVE_GYM:  # pragma: no cover
            super().reset(seed=seed)                                            # env_base.py:220 (seeds gymnasium's np_random)
        if seed is not None:
            self.seed = int(seed)
            self.env.set_seed(self.seed, restart_episodes=True)
        self._log_flush()                                                       # env_base.py:231-232: one file per episode
        terminated, reward, _, observation, sim_time = self.env.k_reset()
        self.step_idx = 0
        self.episode_idx += 1
        self.info.update(step=self.step_idx, episode=self.episode_idx, sim_time=sim_time, cpu_time=time.time(),
                         reward=reward, is_success=False, terminated=False)
        return self._observation(observation), self.info

    def step(self, action):
        """env_base.py:241-259 + the TimeLimit wrapper: (observation, reward, terminated, truncated, info)."""
        if self.squeeze and isinstance(action, dict):
            action = {k: np.asarray(v)[None] for k, v in action.items()}
        terminated, reward, _, observation, sim_time = self.env.k_step(action)
        self.step_idx += 1
        trunc_now = self.step_idx >= MAX_EPISODE_STEPS                              # TimeLimit wrapper
        obs_out = self._observation(observation)
        if self.logger is not None:
            self._log_step(action, self._obs_dev)
        if self.device_outputs:
            torch = env_hip._torch()
            trunc = torch.full((self.num_envs,), trunc_now, dtype=torch.bool, device=reward.device)
            self.info.update(step=self.step_idx, episode=self.episode_idx, sim_time=sim_time, cpu_time=time.time(),
                             reward=reward, is_success=reward > REWARD_SUCCESS_THRESHOLD, terminated=terminated,
                             diverged=(self.env.done & KM_DONE_DIVERGED) != 0)
            return obs_out, reward, terminated, trunc, self.info
        r = reward.detach().cpu().numpy()
        term = terminated.cpu().numpy()
        trunc = np.full(self.num_envs, trunc_now)
        self.info.update(step=self.step_idx, episode=self.episode_idx, sim_time=sim_time.cpu().numpy(), cpu_time=time.time(),
                         reward=r, is_success=r > REWARD_SUCCESS_THRESHOLD, terminated=term,
                         diverged=(self.env.done.cpu().numpy() & KM_DONE_DIVERGED) != 0)
        if self.squeeze:
            return obs_out, float(r[0]), bool(term[0]), bool(trunc[0]), self.info
        return obs_out, r, term, trunc, self.info

    def render(self):
        """env_base.py:215-217: the `top` camera as uint8 RGB [h, w, 3] ([num_envs, h, w, 3] for a batch)."""
        img = self.env.k_render(CAMERAS["top"])
        if self.device_outputs:
            return img
        img = img.cpu().numpy()
        return img[0] if self.squeeze else img

    def close(self):
        self._log_flush()                                                       # env_base.py:262-263
        self.env.k_close()
