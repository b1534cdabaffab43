"""`env_hip` -- the MI355X backend behind gym-kmanip's own backend seam.

The reference picks a backend by calling a module-level `new(gym_env)` and then only ever calls
k_reset / k_step / k_render / k_close on the returned object (reference gym_kmanip/env_base.py:192-200,
:217,:221,:242,:266); `env_sim.new` (env_sim.py:206-211) and `env_real.new` are the two existing
backends.  `env_hip.new(gym_env, num_envs, device)` is the third: same attribute reads from `gym_env`
(mjcf_filename, seed, q_len, q_pos_home, q_id_*_mask, ctrl_id_*_grip, obs_list, act_list), same return
tuple `(terminated, reward, discount, observation, sim_time)` (env_sim.py:194,200), every element with a
leading [num_envs] dimension and the observation an OrderedDict in obs_list order.

Buffers are PyTorch-ROCm tensors (device memory + streams only; all arithmetic is in the HIP library).
"""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict
from typing import Optional

import numpy as np

from . import lib as _libmod
from .model import (CAMERAS, CONTROL_TIMESTEP, ENV_SPECS, KM_ACT_KEYS, KM_CAM_INDEX, CompiledModel, EnvSpec, compile_model)

MJCF_TO_ASSET = {"_env_solo_arm.xml": "solo_arm", "_env_dual_arm.xml": "dual_arm", "_env_torso.xml": "torso"}


def _torch():
    import torch
    return torch


class KManipEnvHip:
    """Batched simulated backend.  One instance <-> one device <-> one stream at a time."""

    def __init__(self, cm: CompiledModel, num_envs: int = 1, device: int = 0, seed: int = 0,
                 env_id_offset: int = 0):
        torch = _torch()
        if not torch.cuda.is_available():
            raise _libmod.KManipError("env_hip needs a HIP device (torch.cuda.is_available() is False); "
                                      "there is no CPU fallback")
        self.L = _libmod.load()
        self.cm = cm
        self.num_envs = int(num_envs)
        self.device = torch.device("cuda", device)
        self.device_index = device
        h = C.c_void_p()
        rc = self.L.kmanip_create(C.byref(cm.desc), self.num_envs, device, C.c_uint64(seed),
                                  C.c_int64(env_id_offset), C.byref(h))
        if rc != 0:
            raise _libmod.KManipError("kmanip_create failed (%d): %s" % (rc, self.L.kmanip_last_error(None).decode()))
        self.h = h
        n = self.num_envs
        self.obs = torch.zeros((n, cm.obs_dim), dtype=torch.float64, device=self.device)
        self.reward = torch.zeros((n,), dtype=torch.float64, device=self.device)
        self.done = torch.zeros((n,), dtype=torch.uint8, device=self.device)
        self.act = torch.zeros((n, cm.act_dim), dtype=torch.float32, device=self.device)
        # k_step return values that never change (terminated is always False in the reference: get_termination -> None,
        # discount 1.0) and the device-side counters behind sim_time: allocated once, no per-step allocation or sync
        self.terminated = torch.zeros((n,), dtype=torch.bool, device=self.device)
        self.discount = torch.ones((n,), dtype=torch.float64, device=self.device)
        self.sim_time = torch.zeros((n,), dtype=torch.float64, device=self.device)
        self._check(self.L.kmanip_bind_sim_time(self.h, C.c_void_p(self.sim_time.data_ptr())), "kmanip_bind_sim_time")

    # ------------------------------------------------------------------ helpers
    def _check(self, rc, what):
        if rc != 0:
            raise _libmod.KManipError("%s failed (%d): %s" % (what, rc, self.L.kmanip_last_error(self.h).decode()))

    def _stream(self):
        return C.c_void_p(_torch().cuda.current_stream(self.device).cuda_stream)

    def obs_dict(self, obs=None) -> "OrderedDict":
        """Zero-copy per-key views of the flat observation, in obs_list order (env_sim.py:111-139)."""
        obs = self.obs if obs is None else obs
        out = OrderedDict()
        for key in self.cm.spec.obs_list:
            if key in self.cm.obs_slices:
                out[key] = obs[:, self.cm.obs_slices[key]]
        return out

    def _check_buf(self, t, shape, dtype, what):
        """Device / dtype / layout checks of a caller-supplied tensor whose raw pointer goes to the kernel (a float64
        action, a CPU tensor, a strided view or a wrong leading dimension would otherwise be silent garbage or a fault)."""
        torch = _torch()
        if not isinstance(t, torch.Tensor):
            raise _libmod.KManipError("%s must be a torch tensor on %s, got %s" % (what, self.device, type(t).__name__))
        if (not t.is_cuda) or t.device != self.device or t.dtype != dtype or tuple(t.shape) != tuple(shape) or not t.is_contiguous():
            raise _libmod.KManipError("%s must be a contiguous %s tensor of shape %s on %s; got %s %s on %s%s" % (
                what, dtype, tuple(shape), self.device, t.dtype, tuple(t.shape), t.device,
                "" if t.is_contiguous() else " (non-contiguous)"))

    def pack_action(self, action) -> "object":
        """dict of arrays keyed like the reference action space (env_base.py:151-188) -> flat [N, act_dim].  A dict of
        DEVICE tensors is packed on the device (one strided copy per key into the handle's action buffer: no host
        round trip, no allocation); NumPy / list values go through one host buffer and a single upload."""
        torch = _torch()
        if isinstance(action, dict):
            if any(isinstance(v, torch.Tensor) and v.is_cuda for v in action.values()):
                keys = list(self.cm.act_slices)
                if all(k in action and isinstance(action[k], torch.Tensor) and action[k].is_cuda
                       and action[k].dtype == torch.float32 for k in keys):
                    # the usual case: every key present, on the device -> ONE concatenation kernel into the action buffer
                    torch.cat([action[k].reshape(self.num_envs, -1) for k in keys], dim=1, out=self.act)
                    return self.act
                self.act.zero_()
                for key, sl in self.cm.act_slices.items():
                    if key in action:
                        v = action[key]
                        if not isinstance(v, torch.Tensor):
                            v = torch.as_tensor(np.asarray(v, dtype=np.float32))
                        self.act[:, sl].copy_(v.reshape(self.num_envs, -1), non_blocking=True)
                return self.act
            flat = np.zeros((self.num_envs, self.cm.act_dim), dtype=np.float32)
            for key, sl in self.cm.act_slices.items():
                if key in action:
                    flat[:, sl] = np.asarray(action[key], dtype=np.float32).reshape(self.num_envs, -1)
            return torch.from_numpy(flat).to(self.device)
        if isinstance(action, torch.Tensor):
            return action.to(device=self.device, dtype=torch.float32).reshape(self.num_envs, self.cm.act_dim).contiguous()
        return torch.as_tensor(np.asarray(action, dtype=np.float32).reshape(self.num_envs, self.cm.act_dim),
                               device=self.device)

    # ------------------------------------------------------------------ the seam (k_* methods)
    def k_reset(self, mask=None):
        """KManipEnvSim.k_reset (env_sim.py:190-194): (terminated, reward, discount, observation, sim_time)."""
        torch = _torch()
        mp = None
        if mask is not None:
            mask = torch.as_tensor(mask, device=self.device).to(torch.uint8).contiguous()
            mp = C.c_void_p(mask.data_ptr())
        self._check(self.L.kmanip_reset(self.h, mp, C.c_void_p(self.obs.data_ptr()), self._stream()), "kmanip_reset")
        return self.terminated, None, None, self.obs_dict(), 0.0

    def step_flat(self, act):
        """Raw batched step on device tensors: returns (obs, reward, done) views of the handle's buffers."""
        self._check_buf(act, (self.num_envs, self.cm.act_dim), _torch().float32, "act")
        self._check(self.L.kmanip_step(self.h, C.c_void_p(act.data_ptr()), C.c_void_p(self.obs.data_ptr()),
                                       C.c_void_p(self.reward.data_ptr()), C.c_void_p(self.done.data_ptr()),
                                       self._stream()), "kmanip_step")
        return self.obs, self.reward, self.done

    def bind_reward_done_record(self, rec0=None, rec1=None):
        """kmanip_bind_reward_done_record: every step_flat / k_step then also writes the packed (reward, done) record of the
        multi-GPU exchange (gym_kmanip_amd/dist.py) into rec0 or rec1 -- float64 [num_envs, 2] device tensors; which one is the
        CALLER's choice (select_reward_done_record, rec0 after the bind) -- so that the all-gather needs no packing kernel on the
        step's stream.  None, None unbinds."""
        torch = _torch()
        if (rec0 is None) != (rec1 is None):
            raise _libmod.KManipError("bind_reward_done_record: two buffers or none")
        for t in (rec0, rec1):
            if t is not None:
                self._check_buf(t, (self.num_envs, 2), torch.float64, "record")
        self._rd_rec = (rec0, rec1)                          # (keeps the tensors alive while bound)
        p = [C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0) for t in (rec0, rec1)]
        self._check(self.L.kmanip_bind_reward_done_record(self.h, p[0], p[1]), "kmanip_bind_reward_done_record")

    def select_reward_done_record(self, index: int):
        """kmanip_select_reward_done_record: the bound buffer (0 / 1) the following steps fill.  The caller must have made the
        step's stream wait for whatever still reads that buffer (include/kmanip.h: ordering invariant)."""
        self._check(self.L.kmanip_select_reward_done_record(self.h, int(index)), "kmanip_select_reward_done_record")

    def observe(self, obs=None, reward=None):
        """kmanip_observe: get_observation + get_reward (env_sim.py:110-179) of the CURRENT state, no step; fills and returns
        (self.obs, self.reward) unless other float64 device tensors are given."""
        torch = _torch()
        obs = self.obs if obs is None else obs
        reward = self.reward if reward is None else reward
        self._check_buf(obs, (self.num_envs, self.cm.obs_dim), torch.float64, "obs")
        self._check_buf(reward, (self.num_envs,), torch.float64, "reward")
        self._check(self.L.kmanip_observe(self.h, C.c_void_p(obs.data_ptr()), C.c_void_p(reward.data_ptr()), self._stream()),
                    "kmanip_observe")
        return obs, reward

    def step_chunk(self, acts, obs=None, reward=None, done=None):
        """K control steps in one launch (kmanip_step_chunk): acts float32 [K, num_envs, act_dim] on the device ->
        (obs [K, N, obs_dim] f64, reward [K, N] f64, done [K, N] u8).  Same results as K step_flat calls; self.obs /
        self.reward / self.done are left holding the last step."""
        torch = _torch()
        K = int(acts.shape[0])
        n = self.num_envs
        self._check_buf(acts, (K, n, self.cm.act_dim), torch.float32, "acts")
        for t, shp, dt, nm in ((obs, (K, n, self.cm.obs_dim), torch.float64, "obs"), (reward, (K, n), torch.float64, "reward"),
                               (done, (K, n), torch.uint8, "done")):
            if t is not None:
                self._check_buf(t, shp, dt, nm)
        if obs is None:
            obs = torch.empty((K, n, self.cm.obs_dim), dtype=torch.float64, device=self.device)
        if reward is None:
            reward = torch.empty((K, n), dtype=torch.float64, device=self.device)
        if done is None:
            done = torch.empty((K, n), dtype=torch.uint8, device=self.device)
        self._check(self.L.kmanip_step_chunk(self.h, K, C.c_void_p(acts.data_ptr()), C.c_void_p(obs.data_ptr()),
                                             C.c_void_p(reward.data_ptr()), C.c_void_p(done.data_ptr()), self._stream()),
                    "kmanip_step_chunk")
        self.obs.copy_(obs[-1]); self.reward.copy_(reward[-1]); self.done.copy_(done[-1])
        return obs, reward, done

    def k_step(self, action):
        """KManipEnvSim.k_step (env_sim.py:196-200).  `terminated` is always False in the reference
        (get_termination -> None); the TimeLimit truncation and the divergence flag are in `self.done`.
        ALIASING: the returned reward / terminated / sim_time tensors and the observation dict's values are the handle's live
        device buffers (views of self.obs), overwritten in place by the next step -- clone() what goes into a rollout list or
        a replay buffer."""
        torch = _torch()
        act = self.pack_action(action)
        self.last_act = act                      # the flat float32 row this step ran on (episode loggers read it)
        self.step_flat(act)
        # sim_time = data.time of each env (env_sim.py:194,200) = steps since its last reset x control_timestep: the step
        # kernel itself fills the bound device buffer -- no extra launch, no host synchronisation
        return self.terminated, self.reward, self.discount, self.obs_dict(), self.sim_time

    def _cam_index(self, cam):
        name = getattr(cam, "name", cam)
        if name not in KM_CAM_INDEX or not self.cm.desc.cam_present[KM_CAM_INDEX[name]]:
            have = [n for n, i in KM_CAM_INDEX.items() if self.cm.desc.cam_present[i]]
            raise _libmod.KManipError("no camera %r in this model; cameras: %s" % (name, ", ".join(have)))
        return KM_CAM_INDEX[name]

    def render_depth(self, cam="grip_r", height: int = 64, width: int = 64, out=None):
        """float32 depth image [num_envs, height, width] (metres along the optical axis) of every env's current state
        -- BASELINE.json config 5's observation (camera branch of env_sim.py:140-145)."""
        torch = _torch()
        ci = self._cam_index(cam)
        if out is None:
            out = torch.empty((self.num_envs, height, width), dtype=torch.float32, device=self.device)
        else:
            self._check_buf(out, (self.num_envs, height, width), torch.float32, "depth")
        self._check(self.L.kmanip_render_depth(self.h, ci, height, width, C.c_void_p(out.data_ptr()), self._stream()),
                    "kmanip_render_depth")
        return out

    def render_rgb(self, cam="top", height=None, width=None, out=None):
        """uint8 RGB image [num_envs, height, width, 3] -- what physics.render(height, width, camera_id) returns in the
        reference (env_sim.py:141-145,187-188).  Size defaults to the camera's reference resolution (__init__.py:157-161)."""
        torch = _torch()
        ci = self._cam_index(cam)
        spec = CAMERAS[getattr(cam, "name", cam)]
        height = spec.h if height is None else height
        width = spec.w if width is None else width
        if out is None:
            out = torch.empty((self.num_envs, height, width, 3), dtype=torch.uint8, device=self.device)
        else:
            self._check_buf(out, (self.num_envs, height, width, 3), torch.uint8, "rgb")
        self._check(self.L.kmanip_render_rgb(self.h, ci, height, width, C.c_void_p(out.data_ptr()), self._stream()),
                    "kmanip_render_rgb")
        return out

    def render_cameras(self, cams=None, out=None):
        """Every camera of the observation (default: this id's `cameras`, head first) at its reference resolution in ONE launch
        (kmanip_render_rgb_multi): {name: uint8 [num_envs, h, w, 3]}.  `out` = such a dict of buffers to fill."""
        torch = _torch()
        names = [getattr(c, "name", c) for c in (self.cm.cameras if cams is None else cams)]
        if not names:
            return {}
        bufs = {}
        for nm in names:
            spec = CAMERAS[nm]
            t = None if out is None else out[nm]
            if t is None:
                t = torch.empty((self.num_envs, spec.h, spec.w, 3), dtype=torch.uint8, device=self.device)
            else:
                self._check_buf(t, (self.num_envs, spec.h, spec.w, 3), torch.uint8, "rgb")
            bufs[nm] = t
        n = len(names)
        ci = (C.c_int32 * n)(*[self._cam_index(nm) for nm in names])
        hh = (C.c_int32 * n)(*[CAMERAS[nm].h for nm in names])
        ww = (C.c_int32 * n)(*[CAMERAS[nm].w for nm in names])
        pp = (C.c_void_p * n)(*[bufs[nm].data_ptr() for nm in names])
        self._check(self.L.kmanip_render_rgb_multi(self.h, n, ci, hh, ww, pp, self._stream()), "kmanip_render_rgb_multi")
        return bufs

    def snapshot_render_state(self, slot: int):
        """Copy qpos -- all a render reads of the state -- into snapshot `slot` (0 / 1) on the current stream
        (kmanip_snapshot_render_state); `set_render_source(slot)` then points the render_* calls at it."""
        self._check(self.L.kmanip_snapshot_render_state(self.h, int(slot), self._stream()), "kmanip_snapshot_render_state")

    def set_render_source(self, slot: int = -1):
        """-1: the render_* calls read the live state (default); 0 / 1: that snapshot (pipeline.RenderBehind)."""
        self._check(self.L.kmanip_set_render_source(self.h, int(slot)), "kmanip_set_render_source")

    def bind_step_depth(self, cam="grip_r", height: int = 64, width: int = 64, out=None):
        """BASELINE config 5: every step_flat / k_step from now on also renders `cam` into the returned buffer
        (float32 [num_envs, height, width]), in the same C call.  bind_step_depth(None) unbinds."""
        torch = _torch()
        if cam is None:
            self._check(self.L.kmanip_bind_step_depth(self.h, 0, 0, 0, None), "kmanip_bind_step_depth")
            self.step_depth = None
            return None
        if out is None:
            out = torch.empty((self.num_envs, height, width), dtype=torch.float32, device=self.device)
        self._check_buf(out, (self.num_envs, height, width), torch.float32, "depth")
        self._check(self.L.kmanip_bind_step_depth(self.h, self._cam_index(cam), height, width, C.c_void_p(out.data_ptr())),
                    "kmanip_bind_step_depth")
        self.step_depth = out                     # keeps the buffer alive while it is bound
        return out

    def scripted_action(self, act=None, generator=None):
        """The reference's synthetic-data policy (examples/2_synthetic_data.py:28-41) for every env, on device:
        action_space.sample() with eer_pos replaced by the unit vector from the right EE site to the cube.
        `act` (float32 [num_envs, act_dim] on the device) is filled with U(-1, 1) when None; returns it."""
        torch = _torch()
        if act is None:
            act = torch.rand((self.num_envs, self.cm.act_dim), device=self.device, generator=generator) * 2 - 1
        self._check(self.L.kmanip_scripted_action(self.h, C.c_void_p(act.data_ptr()), self._stream()), "kmanip_scripted_action")
        return act

    def sample_action(self, act=None, ahead=0):
        """action_space.sample() for every env on the device (examples/2_log_with_h5py.py:22-26): U[-1, 1) float32 from the
        counter-based stream keyed (seed; global env id, episode, step) -- identical to the CPU oracle's.  `ahead`: the action
        the env needs that many control steps from now (TimeLimit-only episodes).  Returns the [num_envs, act_dim] tensor."""
        torch = _torch()
        if act is None:
            act = torch.empty((self.num_envs, self.cm.act_dim), dtype=torch.float32, device=self.device)
        self._check_buf(act, (self.num_envs, self.cm.act_dim), torch.float32, "act")
        self._check(self.L.kmanip_sample_action(self.h, C.c_void_p(act.data_ptr()), int(ahead), self._stream()), "kmanip_sample_action")
        return act

    def k_render(self, cam):
        """KManipEnvSim.k_render (env_sim.py:187-188): physics.render(cam.h, cam.w, camera_id=cam.name) -> uint8 RGB,
        for every env ([num_envs, h, w, 3], device tensor).  `cam` is a Cam (model.CAMERAS) or a camera name."""
        return self.render_rgb(cam)

    def k_close(self):
        if getattr(self, "h", None):
            self.L.kmanip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.k_close()
        except Exception:
            pass

    # ------------------------------------------------------------------ state / diagnostics (parity tests)
    def get_state(self):
        cm, n = self.cm, self.num_envs
        qpos = np.zeros((n, cm.nq)); qvel = np.zeros((n, cm.nv)); ctrl = np.zeros((n, cm.nu)); warm = np.zeros((n, cm.nv))
        step = np.zeros(n, dtype=np.int32)
        p = lambda a, t=C.c_double: a.ctypes.data_as(C.POINTER(t))
        self._check(self.L.kmanip_get_state(self.h, p(qpos), p(qvel), p(ctrl), p(warm), p(step, C.c_int32)), "kmanip_get_state")
        return qpos, qvel, ctrl, warm, step

    def step_counters(self):
        """Per-env step index inside the current episode (only the int32 counters cross PCIe)."""
        step = np.zeros(self.num_envs, dtype=np.int32)
        self._check(self.L.kmanip_get_state(self.h, None, None, None, None, step.ctypes.data_as(C.POINTER(C.c_int32))),
                    "kmanip_get_state")
        return step

    def set_seed(self, seed: int, restart_episodes: bool = True):
        """Re-key the cube-spawn stream (KManipEnv.reset(seed=...)); with restart_episodes the next k_reset is episode 0."""
        self._check(self.L.kmanip_set_seed(self.h, C.c_uint64(int(seed)), int(restart_episodes)), "kmanip_set_seed")

    def get_episode(self):
        """Per-env episode counter (keys the cube-spawn stream together with seed and global env id)."""
        ep = np.zeros(self.num_envs, dtype=np.int32)
        self._check(self.L.kmanip_get_episode(self.h, ep.ctypes.data_as(C.POINTER(C.c_int32))), "kmanip_get_episode")
        return ep

    def set_episode(self, episode):
        ep = np.ascontiguousarray(episode, dtype=np.int32)
        assert ep.shape == (self.num_envs,)
        self._check(self.L.kmanip_set_episode(self.h, ep.ctypes.data_as(C.POINTER(C.c_int32))), "kmanip_set_episode")

    def checkpoint(self):
        """Complete restartable state: (qpos, qvel, ctrl, qacc_warmstart, step_idx, episode) as host arrays."""
        return self.get_state() + (self.get_episode(),)

    def restore(self, ckpt):
        qpos, qvel, ctrl, warm, step, episode = ckpt
        self.set_state(qpos, qvel, ctrl, warm, step)
        self.set_episode(episode)

    def set_state(self, qpos=None, qvel=None, ctrl=None, warm=None, step=None):
        def p(a, dt, t):
            if a is None:
                return None, None
            a = np.ascontiguousarray(a, dtype=dt)
            return a, a.ctypes.data_as(C.POINTER(t))
        keep = []
        args = []
        for a, dt, t in [(qpos, np.float64, C.c_double), (qvel, np.float64, C.c_double), (ctrl, np.float64, C.c_double),
                         (warm, np.float64, C.c_double), (step, np.int32, C.c_int32)]:
            arr, ptr = p(a, dt, t)
            keep.append(arr); args.append(ptr)
        self._check(self.L.kmanip_set_state(self.h, *args), "kmanip_set_state")

    def get_diag(self):
        n = self.num_envs
        mask = np.zeros(n, dtype=np.uint32); nfev = np.zeros((n, 2), dtype=np.int32); st = np.zeros((n, 2), dtype=np.int32)
        self._check(self.L.kmanip_get_diag(self.h, mask.ctypes.data_as(C.POINTER(C.c_uint32)),
                                           nfev.ctypes.data_as(C.POINTER(C.c_int32)),
                                           st.ctypes.data_as(C.POINTER(C.c_int32))), "kmanip_get_diag")
        return mask, nfev, st

    def ik(self, arm, qpos, goal_pos, goal_quat):
        """Batched ik_mujoco.ik on device: returns (q_out, qpos_after, nfev, status)."""
        qpos = np.ascontiguousarray(qpos, dtype=np.float64).copy()
        n = qpos.shape[0]
        gp = np.ascontiguousarray(goal_pos, dtype=np.float64); gq = np.ascontiguousarray(goal_quat, dtype=np.float64)
        nik = self.cm.desc.arm_nq[arm]
        q = np.zeros((n, nik)); nfev = np.zeros(n, dtype=np.int32); st = np.zeros(n, dtype=np.int32)
        p = lambda a, t=C.c_double: a.ctypes.data_as(C.POINTER(t))
        self._check(self.L.kmanip_ik(self.h, arm, n, p(qpos), p(gp), p(gq), p(q), p(nfev, C.c_int32), p(st, C.c_int32)), "kmanip_ik")
        return q, qpos, nfev, st

    def ik_eval(self, arm, qpos, goal_pos, goal_quat):
        """(ik_res, ik_jac) of the device IK at x = qpos[q_mask], q_pos_prev = x: res [n, 6+2N], jac [n, 6+2N, N]."""
        qpos = np.ascontiguousarray(qpos, dtype=np.float64)
        n = qpos.shape[0]
        gp = np.ascontiguousarray(goal_pos, dtype=np.float64); gq = np.ascontiguousarray(goal_quat, dtype=np.float64)
        nik = self.cm.desc.arm_nq[arm]
        res = np.zeros((n, 6 + 2 * nik)); jac = np.zeros((n, 6 + 2 * nik, nik))
        p = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
        self._check(self.L.kmanip_ik_eval(self.h, arm, n, p(qpos), p(gp), p(gq), p(res), p(jac)), "kmanip_ik_eval")
        return res, jac

    def enable_timing(self, on=True):
        """on = True / 1: events around every step; an int k > 1: around every k-th step (the sampled average; an event pair costs
        the stream about 5 us); False / 0: off."""
        self._check(self.L.kmanip_enable_timing(self.h, int(on)), "kmanip_enable_timing")

    def timing_summary(self):
        """(ik_ms_sum, dyn_ms_sum, render_ms_sum, nsteps) of the steps recorded since enable_timing / the last summary: the
        launch gap / stand-alone IK leg, k_step, and the in-step render kernel (bind_step_depth; 0 when nothing is bound)."""
        a = C.c_double(); b = C.c_double(); r = C.c_double(); n = C.c_int32()
        self._check(self.L.kmanip_timing_summary(self.h, C.byref(a), C.byref(b), C.byref(r), C.byref(n)), "kmanip_timing_summary")
        return float(a.value), float(b.value), float(r.value), int(n.value)


def spec_from_gym_env(gym_env) -> EnvSpec:
    """Read the same attributes env_sim.new reads from the KManipEnv instance."""
    return EnvSpec(env_id=getattr(gym_env, "env_id", "custom"), asset=MJCF_TO_ASSET[gym_env.mjcf_filename],
                   obs_list=list(gym_env.obs_list), act_list=list(gym_env.act_list),
                   q_pos_home=np.asarray(gym_env.q_pos_home, dtype=np.float32),
                   q_id_r_mask=None if gym_env.q_id_r_mask is None else list(gym_env.q_id_r_mask),
                   q_id_l_mask=None if getattr(gym_env, "q_id_l_mask", None) is None else list(gym_env.q_id_l_mask),
                   ctrl_id_r_grip=None if gym_env.ctrl_id_r_grip is None else list(gym_env.ctrl_id_r_grip),
                   ctrl_id_l_grip=None if getattr(gym_env, "ctrl_id_l_grip", None) is None else list(gym_env.ctrl_id_l_grip))


def new(gym_env, num_envs: int = 1, device: int = 0, env_id_offset: int = 0, **compile_kw) -> KManipEnvHip:
    """Drop-in third backend: `self.env = env_hip.new(self)` in KManipEnv.__init__ (env_base.py:192-200)."""
    cm = compile_model(spec_from_gym_env(gym_env), **compile_kw)
    return KManipEnvHip(cm, num_envs=num_envs, device=device, seed=int(getattr(gym_env, "seed", 0) or 0),
                        env_id_offset=env_id_offset)


def make(env_id: str, num_envs: int = 1, device: int = 0, seed: int = 0, env_id_offset: int = 0, **compile_kw) -> KManipEnvHip:
    """Shortcut used by bench/tests: build straight from a registered env id (__init__.py:244-483)."""
    return KManipEnvHip(compile_model(ENV_SPECS[env_id], **compile_kw), num_envs=num_envs, device=device, seed=seed,
                        env_id_offset=env_id_offset)
