"""ctypes front-end of the C oracle (oracle/kmanip_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (gym_kmanip_amd) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libkmanip_oracle.so")
_lib = None

NLMAX = 20
NVMAX = NLMAX + 6
NQMAX = NLMAX + 7


class KoState(C.Structure):
    _fields_ = [("qpos", C.c_double * NQMAX), ("qvel", C.c_double * NVMAX), ("ctrl", C.c_double * NLMAX),
                ("qacc_warm", C.c_double * NVMAX), ("time", C.c_double), ("step_idx", C.c_int32),
                ("episode", C.c_int32)]


class KoDiag(C.Structure):
    _fields_ = [("contact_mask", C.c_uint32), ("ik_nfev", C.c_int32 * 2), ("ik_status", C.c_int32 * 2),
                ("nefc", C.c_int32), ("solver_iter", C.c_int32), ("diverged", C.c_int32)]


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "kmanip_oracle.c")
    hdr = os.path.join(_HERE, "..", "include", "kmanip.h")
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.exists(p) and os.path.getmtime(p) > os.path.getmtime(_LIB_PATH) for p in (src, hdr))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.ko_state_size.restype = C.c_int
        assert _lib.ko_state_size() == C.sizeof(KoState), (_lib.ko_state_size(), C.sizeof(KoState))
        assert _lib.ko_diag_size() == C.sizeof(KoDiag)
    return _lib


def _p(a, t=C.c_double):
    return a.ctypes.data_as(C.POINTER(t))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Oracle:
    """Batched float64 CPU oracle for one compiled model (gym_kmanip_amd.model.CompiledModel)."""

    def __init__(self, cm, num_envs: int = 1, seed: int = 0, env_id_offset: int = 0):
        self.cm = cm
        self.desc = cm.desc
        self.n = num_envs
        self.seed = seed
        self.env_id_offset = env_id_offset
        self.L = lib()
        assert self.L.ko_model_desc_size() == C.sizeof(type(cm.desc))
        self.states = (KoState * num_envs)()
        self.diag = (KoDiag * num_envs)()

    # ---- episode API
    def reset(self, episode=None):
        obs = np.zeros((self.n, self.cm.obs_dim))
        ep = None if episode is None else np.ascontiguousarray(episode, dtype=np.int32)
        self.L.ko_reset_batch(C.byref(self.desc), C.c_uint64(self.seed), C.c_int64(self.env_id_offset), self.n,
                              None if ep is None else _p(ep, C.c_int32), self.states, _p(obs))
        return obs

    def step(self, act, nthreads: int = 1):
        act = np.ascontiguousarray(act, dtype=np.float32).reshape(self.n, self.cm.act_dim)
        obs = np.zeros((self.n, self.cm.obs_dim))
        rew = np.zeros(self.n)
        done = np.zeros(self.n, dtype=np.uint8)
        self.L.ko_step_batch(C.byref(self.desc), C.c_uint64(self.seed), C.c_int64(self.env_id_offset), self.n,
                             self.states, _p(act, C.c_float), _p(obs), _p(rew), _p(done, C.c_uint8), self.diag,
                             nthreads)
        return obs, rew, done

    def sample_action(self, ahead: int = 0):
        """action_space.sample() for every env from the counter-based stream (the device's kmanip_sample_action draws the
        same bits)."""
        act = np.zeros((self.n, self.cm.act_dim), dtype=np.float32)
        self.L.ko_sample_action(C.byref(self.desc), C.c_uint64(self.seed), C.c_int64(self.env_id_offset), self.n,
                                self.states, int(ahead), _p(act, C.c_float))
        return act

    # ---- state access (numpy views of the ctypes array: no per-element Python loops)
    _DT = np.dtype([("qpos", np.float64, NQMAX), ("qvel", np.float64, NVMAX), ("ctrl", np.float64, NLMAX),
                    ("qacc_warm", np.float64, NVMAX), ("time", np.float64), ("step_idx", np.int32), ("episode", np.int32)])

    def _view(self):
        assert self._DT.itemsize == C.sizeof(KoState)
        return np.frombuffer(self.states, dtype=self._DT)

    def get_state(self):
        nq, nv, nu = self.cm.nq, self.cm.nv, self.cm.nu
        v = self._view()
        return (v["qpos"][:, :nq].copy(), v["qvel"][:, :nv].copy(), v["ctrl"][:, :nu].copy(), v["qacc_warm"][:, :nv].copy(),
                v["step_idx"].copy())

    def set_state(self, qpos=None, qvel=None, ctrl=None, warm=None, step=None):
        v = self._view()
        for name, val in (("qpos", qpos), ("qvel", qvel), ("ctrl", ctrl), ("qacc_warm", warm)):
            if val is not None:
                val = np.asarray(val, dtype=np.float64)
                v[name][:, :val.shape[1]] = val
        if step is not None:
            v["step_idx"][:] = np.asarray(step, dtype=np.int32)

    def get_episode(self):
        return self._view()["episode"].copy()

    def set_episode(self, episode):
        self._view()["episode"][:] = np.asarray(episode, dtype=np.int32)

    def get_diag(self):
        mask = np.array([d.contact_mask for d in self.diag], dtype=np.uint32)
        nfev = np.array([list(d.ik_nfev) for d in self.diag], dtype=np.int32)
        status = np.array([list(d.ik_status) for d in self.diag], dtype=np.int32)
        return mask, nfev, status

    # ---- pieces
    def fk(self, qpos):
        qpos = _f64(qpos)
        nl = self.cm.nlink
        xpos = np.zeros((nl, 3)); xquat = np.zeros((nl, 4)); sp = np.zeros((2, 3)); sm = np.zeros((2, 9))
        self.L.ko_fk(C.byref(self.desc), _p(qpos), _p(xpos), _p(xquat), _p(sp), _p(sm))
        return xpos, xquat, sp, sm.reshape(2, 3, 3)

    def ik_res(self, arm, qpos, x, q_prev, goal_pos, goal_quat):
        n = self.desc.arm_nq[arm]
        f = np.zeros(6 + 2 * n)
        qp = _f64(qpos).copy()
        self.L.ko_ik_res(C.byref(self.desc), arm, _p(qp), _p(_f64(x)), _p(_f64(q_prev)), _p(_f64(goal_pos)),
                         _p(_f64(goal_quat)), _p(f))
        return f

    def ik_jac(self, arm, qpos, x, q_prev, goal_pos, goal_quat):
        n = self.desc.arm_nq[arm]
        J = np.zeros((6 + 2 * n, n))
        qp = _f64(qpos).copy()
        self.L.ko_ik_jac(C.byref(self.desc), arm, _p(qp), _p(_f64(x)), _p(_f64(q_prev)), _p(_f64(goal_pos)),
                         _p(_f64(goal_quat)), _p(J))
        return J

    def ik(self, arm, qpos, goal_pos, goal_quat):
        """Returns (q_out, qpos_after, nfev, status)."""
        n = self.desc.arm_nq[arm]
        qp = _f64(qpos).copy()
        q = np.zeros(n)
        nfev = C.c_int32(); st = C.c_int32()
        self.L.ko_ik(C.byref(self.desc), arm, _p(qp), _p(_f64(goal_pos)), _p(_f64(goal_quat)), _p(q),
                     C.byref(nfev), C.byref(st))
        return q, qp, nfev.value, st.value

    def euler_goal(self, site_mat, delta):
        q = np.zeros(4)
        self.L.ko_euler_goal(_p(_f64(site_mat).reshape(-1)), _p(_f64(delta)), _p(q))
        return q

    def dynamics(self, qpos, qvel, ctrl):
        nv = self.cm.nv
        ne_max = self.L.ko_nefc_max()
        M = np.zeros((nv, nv)); bias = np.zeros(nv); qs = np.zeros(nv); qa = np.zeros(nv)
        J = np.zeros((ne_max, nv)); aref = np.zeros(ne_max); R = np.zeros(ne_max)
        nefc = C.c_int32()
        rc = self.L.ko_dynamics(C.byref(self.desc), _p(_f64(qpos)), _p(_f64(qvel)), _p(_f64(ctrl)), _p(M), _p(bias),
                                _p(qs), _p(qa), C.byref(nefc), _p(J), _p(aref), _p(R))
        assert rc == 0
        ne = nefc.value
        return dict(M=M, bias=bias, qacc_smooth=qs, qacc=qa, nefc=ne, J=J[:ne], aref=aref[:ne], R=R[:ne])

    def constraint_rows(self, qpos, qvel):
        ne_max = self.L.ko_nefc_max()
        t = np.zeros(ne_max, dtype=np.int32); fl = np.zeros(ne_max)
        n = self.L.ko_constraint_rows(C.byref(self.desc), _p(_f64(qpos)), _p(_f64(qvel)), _p(t, C.c_int32), _p(fl))
        assert n >= 0
        return t[:n], fl[:n]

    # ---- pieces used by tests/tools/refrun.py (the reference's own Python running over this physics)
    def physics_step(self, qpos, qvel, ctrl, warm, qpos_step1, nsub):
        """dm_control Physics.step(nsub), legacy order, for ONE env.  Returns (qpos, qvel, warm, bad, mask, touch_finger_cube,
        touch_cube_table)."""
        st = KoState()
        nq, nv, nu = self.cm.nq, self.cm.nv, self.cm.nu
        st.qpos[:nq] = list(qpos); st.qvel[:nv] = list(qvel); st.ctrl[:nu] = list(ctrl); st.qacc_warm[:nv] = list(warm)
        mask = C.c_uint32(); tf = C.c_int32(); tt = C.c_int32()
        bad = self.L.ko_physics_step(C.byref(self.desc), C.byref(st), _p(_f64(qpos_step1)), int(nsub), C.byref(mask),
                                     C.byref(tf), C.byref(tt))
        return (np.array(st.qpos[:nq]), np.array(st.qvel[:nv]), np.array(st.qacc_warm[:nv]), int(bad), mask.value,
                tf.value, tt.value)

    def after_reset(self, qpos, qvel, ctrl):
        """mj_forward with actuation disabled: returns qacc_warmstart."""
        st = KoState()
        nq, nv, nu = self.cm.nq, self.cm.nv, self.cm.nu
        st.qpos[:nq] = list(qpos); st.qvel[:nv] = list(qvel); st.ctrl[:nu] = list(ctrl)
        self.L.ko_after_reset(C.byref(self.desc), C.byref(st))
        return np.array(st.qacc_warm[:nv])

    def contact_mask(self, qpos):
        mask = C.c_uint32(); tf = C.c_int32(); tt = C.c_int32()
        self.L.ko_contact_mask(C.byref(self.desc), _p(_f64(qpos)), C.byref(mask), C.byref(tf), C.byref(tt))
        return mask.value, tf.value, tt.value

    def observe(self, qpos, qvel):
        """(get_observation's state keys, get_reward) at a state: env_sim.py:110-179."""
        obs = np.zeros(self.cm.obs_dim)
        rew = C.c_double()
        self.L.ko_observe(C.byref(self.desc), _p(_f64(qpos)), _p(_f64(qvel)), _p(obs), C.byref(rew))
        return obs, rew.value

    def render_depth(self, qpos, cam=0, h=64, w=64):
        out = np.zeros((h, w), dtype=np.float32)
        self.L.ko_render_depth(C.byref(self.desc), _p(_f64(qpos)), cam, h, w, _p(out, C.c_float))
        return out

    def render_rgb(self, qpos, cam=0, h=40, w=60):
        out = np.zeros((h, w, 3), dtype=np.uint8)
        self.L.ko_render_rgb(C.byref(self.desc), _p(_f64(qpos)), cam, h, w, _p(out, C.c_uint8))
        return out

    def scripted_eer_pos(self, qpos):
        o = np.zeros(3)
        self.L.ko_scripted_eer_pos(C.byref(self.desc), _p(_f64(qpos)), _p(o))
        return o

    def philox(self, ctr, key):
        c = np.ascontiguousarray(ctr, dtype=np.uint32); k = np.ascontiguousarray(key, dtype=np.uint32)
        o = np.zeros(4, dtype=np.uint32)
        self.L.ko_philox(_p(c, C.c_uint32), _p(k, C.c_uint32), _p(o, C.c_uint32))
        return o
