"""Run the reference's OWN Python hot path in the build container.  TEST INFRASTRUCTURE, BUILD CONTAINER ONLY.

`gym_kmanip/env_base.py`, `env_sim.py`, `ik_mujoco.py` and `__init__.py` are pure Python/NumPy/SciPy over four
third-party packages this image lacks (`gymnasium`, `dm_env`, `dm_control`, `mujoco`; SURVEY.md 8c: ordinary
ModuleNotFoundError, nothing was denied).  `install()` puts small stand-ins for those packages into `sys.modules`, puts
`/root/reference` on `sys.path` and imports the reference modules UNMODIFIED -- nothing is copied, nothing is written under
/root/reference (bytecode writing is switched off first), and neither this module's output nor the reference travels to the
GPU box: `tests/tools/make_golden_ref.py` turns runs of it into small `.npz` fixtures (inputs + expected outputs).

What then runs is the reference's own code:
    KManipEnv.__init__/reset/step (env_base.py:16-267)      spaces, info dict, 5-tuple unpacking
    the eight `register(...)` kwargs (__init__.py:244-483)   recorded by the stand-in `register`
    KManipEnvSim.k_reset/k_step, new() (env_sim.py:182-211)
    KManipTask.initialize_episode / before_step / get_observation / get_reward (env_sim.py:23-179)
    ik / ik_res / ik_jac (ik_mujoco.py:20-155) calling the REAL scipy.optimize.least_squares
What does NOT come from the reference (it lives in the absent third-party packages) and is stood in for here:
    mujoco.mj_kinematics / mj_comPos / mj_jacSite / mju_mat2Quat / mju_subQuat / mjd_subQuat   -> oracle/ik_scipy.py (NumPy)
    dm_control mujoco.Physics (data/model/named views, reset_context, step, set_control, render) -> `Physics` below; its
        step(n) is the C oracle's restatement of MuJoCo's mj_step (oracle/kmanip_oracle.c: ko_physics_step), on the build's
        surrogate model (the reference's meshes are absent)
    dm_control rl.control.Environment / suite.base.Task (reset/step call order), dm_env.TimeStep/StepType -> below, restated
        from dm_control's published source
    gymnasium.Env / spaces.Box / spaces.Dict / register -> below (attribute holders)
    h5py (File / attrs / create_group / create_dataset / row writes; round 5) -> tests/tools/h5_recorder.py, an in-memory
        recorder: the reference's log_h5py.new / cam / step / end (log_h5py.py:13-61) then run unmodified from inside
        KManipEnv.reset / step / close (env_base.py:231-263) and what they built is read back as a tree.  The reference puts its
        log directory under its own package (DATA_DIR, __init__.py:12); make_env(log_dir_root=...) points that module attribute
        at a scratch directory first -- nothing is written under /root/reference
So a fixture made with this module pins everything the reference itself wrote (decode, casts, IK call, clips, obs/reward
packing, reset, tuple/ info plumbing, the constants and masks of the eight ids) to the reference; MuJoCo's mj_step and
dm_control's call order stay pinned to the build's reading of their documentation.
"""
from __future__ import annotations

import collections
import contextlib
import enum
import os
import sys
import types

import numpy as np

REFERENCE_ROOT = "/root/reference"
_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.abspath(os.path.join(_HERE, "..", ".."))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)
import h5_recorder  # noqa: E402  (the in-memory `h5py` stand-in: log_h5py.py's calls are recorded as a tree)

REGISTRY = collections.OrderedDict()     # env id -> dict(entry_point, max_episode_steps, nondeterministic, kwargs)
IK_LOG = []                              # one (nfev, status) or (0, -2) ["IK failed"] per ik() call, appended by the observer
RIG_FINGER_GEOM_NAMES = False            # True: the finger colliders carry the geom names env_sim.py:171-174 looks for


# ------------------------------------------------------------------------------------------------ gymnasium stand-in
class _Box:
    def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
        self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), np.dtype(dtype)

    def sample(self, rng):
        if np.issubdtype(self.dtype, np.floating):
            return rng.uniform(self.low, self.high, self.shape).astype(self.dtype)
        return rng.integers(self.low, self.high + 1, self.shape).astype(self.dtype)


class _DictSpace:
    def __init__(self, spaces=None, seed=None):
        self.spaces = collections.OrderedDict(spaces or {})

    def sample(self, rng):
        return collections.OrderedDict((k, s.sample(rng)) for k, s in self.spaces.items())


class _Env:
    metadata = {}

    def reset(self, *, seed=None, options=None):
        self._np_random_seed = seed

    def close(self):
        pass


def _register(id, entry_point=None, max_episode_steps=None, nondeterministic=False, kwargs=None, **other):
    REGISTRY[id] = dict(entry_point=entry_point, max_episode_steps=max_episode_steps, nondeterministic=nondeterministic,
                        kwargs=dict(kwargs or {}))


# ------------------------------------------------------------------------------------------------ dm_env stand-in
class StepType(enum.IntEnum):
    FIRST = 0
    MID = 1
    LAST = 2


TimeStep = collections.namedtuple("TimeStep", ["step_type", "reward", "discount", "observation"])


# ------------------------------------------------------------------------------------------------ mujoco / dm_control stand-in
class _Site:
    def __init__(self, sid):
        self.id = sid
        self.xpos = np.zeros(3)
        self.xmat = np.zeros(9)


class _Contact:
    def __init__(self, g1, g2):
        self.geom1, self.geom2 = g1, g2


class _NamedXpos:
    def __init__(self, physics):
        self._p = physics

    def __getitem__(self, name):
        return self._p._body_xpos[name]


class _Holder:
    pass


class Physics:
    """Duck-typed dm_control.mujoco.Physics: exactly the attributes env_sim.py / ik_mujoco.py touch."""

    _ASSET_OF = {"_env_solo_arm.xml": ("solo_arm", "KManipSoloArm"), "_env_dual_arm.xml": ("dual_arm", "KManipDualArm"),
                 "_env_torso.xml": ("torso", "KManipTorso")}

    @classmethod
    def from_xml_path(cls, path):
        return cls(os.path.basename(path))

    def __init__(self, mjcf_filename):
        from gym_kmanip_amd.model import compile_model
        from oracle import ik_scipy as S
        from oracle.oracle import Oracle
        asset_name, env_id = self._ASSET_OF[mjcf_filename]
        self.cm = compile_model(env_id)
        self.oracle = Oracle(self.cm, 1)
        self._S = S
        asset = self.cm.asset
        self.arm = S.NumpyArm(asset)
        nl = self.nl = self.cm.nlink
        self.model = _Holder()
        self.model.ptr = self
        self.model.nq, self.model.nv, self.model.nu = nl + 7, nl + 6, nl
        self.model.jnt_range = np.vstack([np.array([l["joint"]["range"] for l in asset["links"]], dtype=float),
                                          np.zeros((1, 2))])           # the cube's free joint is the last joint
        self.model.name2id = self._name2id
        self.model.id2name = self._id2name
        self.data = _Holder()
        self.data.ptr = self
        self.data.qpos = np.zeros(nl + 7)
        self.data.qvel = np.zeros(nl + 6)
        self.data.ctrl = np.zeros(nl)
        nmocap = 1 if nl == 10 else 2                                  # hand_r (and hand_l): _env_*.xml mocap bodies
        self.data.mocap_pos = np.zeros((nmocap, 3))
        self.data.mocap_quat = np.tile(np.array([1.0, 0, 0, 0]), (nmocap, 1))
        self.data.time = 0.0
        self.data.ncon = 0
        self.data.contact = []
        self._site_names = [n for n in ("eer_site_pos", "eel_site_pos") if n in asset["sites"]]
        self._sites = {n: _Site(i) for i, n in enumerate(self._site_names)}
        self.data.site = lambda name: self._sites[name]
        self.named = _Holder()
        self.named.data = _Holder()
        self.named.data.xpos = _NamedXpos(self)
        self._body_xpos = {}
        self._warm = np.zeros(nl + 6)
        self._qpos_step1 = np.zeros(nl + 7)
        self._qpos0 = np.zeros(nl + 7)
        self._qpos0[nl:nl + 3] = asset["cube"]["pos0"]
        self._qpos0[nl + 3:] = asset["cube"]["quat0"]
        # geoms: 0 = cube, 1 = table (the only named geoms of the reference scene: scene.xml:15,20), 2 + s = collider sphere s
        self._geom_names = ["cube", "table"] + [None] * len(asset["spheres"])
        if RIG_FINGER_GEOM_NAMES:
            for s in range(2 * (nl // 10)):                            # fingers come first, two per arm, right arm first
                self._geom_names[2 + s] = "right_gripper_finger" if s < 2 else "left_gripper_finger"
        self.contact_mask = 0
        self.reset()

    # ---- names
    def _name2id(self, name, kind):
        if kind == "joint" and name == "cube_joint":
            return self.nl
        raise KeyError((name, kind))

    def _id2name(self, i, kind):
        assert kind == "geom"
        return self._geom_names[i]

    # ---- stages
    def kinematics(self):
        """mj_kinematics (+ mj_comPos: nothing extra is read from it) at data.qpos."""
        xpos, xquat, axis = self.arm.fk(self.data.qpos)
        self._fk = (xpos, xquat, axis)
        for n, s in self._sites.items():
            p, m = self.arm.site(n, xpos, xquat)
            s.xpos[:] = p
            s.xmat[:] = m.reshape(9)
            self._body_xpos[n[:-4]] = s.xpos                          # body "eer_site": the site sits at its origin
        self._body_xpos["cube"] = self.data.qpos[self.nl:self.nl + 3]

    def _collide(self, mask):
        self.contact_mask = mask
        con = []
        for c in range(8):
            if mask & (1 << c):
                con.append(_Contact(0, 1))                             # box < mesh: geom1 = cube
        for s in range(len(self._geom_names) - 2):
            if mask & (1 << (8 + s)):
                con.append(_Contact(0, 2 + s))
            if mask & (1 << (20 + s)):
                con.append(_Contact(1, 2 + s))
        self.data.contact = con
        self.data.ncon = len(con)

    def forward(self):
        self.kinematics()
        self._collide(self.oracle.contact_mask(self.data.qpos)[0])
        self._qpos_step1[:] = self.data.qpos

    def reset(self):
        self.data.qpos[:] = self._qpos0
        self.data.qvel[:] = 0
        self.data.ctrl[:] = 0
        self.data.time = 0.0
        self._warm[:] = 0
        self.forward()

    def after_reset(self):
        # dm_control: `with self.model.disable('actuation'): self.forward()`
        self._warm[:] = self.oracle.after_reset(self.data.qpos, self.data.qvel, self.data.ctrl)
        self.forward()

    @contextlib.contextmanager
    def reset_context(self):
        self.reset()
        yield self
        self.after_reset()

    def set_control(self, control):
        np.copyto(self.data.ctrl, control)

    def timestep(self):
        return self.cm.desc.timestep

    def time(self):
        return self.data.time

    def step(self, nstep=1):
        q, v, w, bad, mask, _, _ = self.oracle.physics_step(self.data.qpos, self.data.qvel, self.data.ctrl, self._warm,
                                                            self._qpos_step1, nstep)
        if bad:
            raise RuntimeError("PhysicsError: the simulation diverged")
        self.data.qpos[:] = q
        self.data.qvel[:] = v
        self._warm[:] = w
        self.data.time += nstep * self.cm.desc.timestep
        self.kinematics()
        self._collide(mask)
        self._qpos_step1[:] = self.data.qpos

    def render(self, height=240, width=320, camera_id=-1):
        from gym_kmanip_amd.model import KM_CAM_INDEX
        return self.oracle.render_rgb(self.data.qpos, KM_CAM_INDEX[camera_id], height, width)


def _mj_kinematics(m, d):
    d.kinematics()


def _mj_comPos(m, d):
    pass


def _mj_jacSite(m, d, jacp, jacr, site_id):
    name = d._site_names[site_id]
    xpos, xquat, axis = d._fk
    p, _ = d.arm.site(name, xpos, xquat)
    jp, jr = d.arm.jac_site(name, xpos, axis, p)
    if jacp is not None:
        jacp[:] = 0
        jacp[:, :d.nl] = jp
    if jacr is not None:
        jacr[:] = 0
        jacr[:, :d.nl] = jr


def _mju_mat2Quat(quat, mat):
    from oracle import ik_scipy as S
    quat[:] = S.mju_mat2quat(np.asarray(mat).reshape(9))


def _mju_subQuat(res, qa, qb):
    from oracle import ik_scipy as S
    res[:] = S.mju_subquat(np.asarray(qa).reshape(4), np.asarray(qb).reshape(4))


def _mjd_subQuat(qa, qb, Da, Db):
    from oracle import ik_scipy as S
    db = S.mjd_subquat_b(np.asarray(qa).reshape(4), np.asarray(qb).reshape(4))
    if Db is not None:
        Db[:] = db.reshape(Db.shape)
    if Da is not None:
        Da[:] = (-db.T).reshape(Da.shape)


class Task:
    """dm_control.suite.base.Task: the parts env_sim.KManipTask inherits."""

    def __init__(self, random=None):
        if not isinstance(random, np.random.RandomState):
            random = np.random.RandomState(random)
        self._random = random
        self._visualize_reward = False

    @property
    def random(self):
        return self._random

    def initialize_episode(self, physics):
        pass

    def before_step(self, action, physics):
        action = getattr(action, "continuous_actions", action)
        physics.set_control(action)

    def after_step(self, physics):
        pass

    def get_termination(self, physics):
        return None


class Environment:
    """dm_control.rl.control.Environment: reset / step call order (legacy_step=True, no time limit), SURVEY.md A.3."""

    def __init__(self, physics, task, time_limit=float("inf"), control_timestep=None, n_sub_steps=None,
                 flat_observation=False, legacy_step=True):
        self._physics, self._task = physics, task
        if control_timestep is not None:
            n = control_timestep / physics.timestep()
            assert abs(n - round(n)) < 1e-6
            self._n_sub_steps = int(round(n))
        else:
            self._n_sub_steps = n_sub_steps or 1
        self._step_limit = float("inf") if time_limit == float("inf") else time_limit / (physics.timestep() * self._n_sub_steps)
        self._step_count = 0
        self._reset_next_step = True

    @property
    def physics(self):
        return self._physics

    @property
    def task(self):
        return self._task

    def reset(self):
        self._reset_next_step = False
        self._step_count = 0
        with self._physics.reset_context():
            self._task.initialize_episode(self._physics)
        observation = self._task.get_observation(self._physics)
        return TimeStep(StepType.FIRST, None, None, observation)

    def step(self, action):
        if self._reset_next_step:
            return self.reset()
        self._task.before_step(action, self._physics)
        self._physics.step(self._n_sub_steps)
        self._task.after_step(self._physics)
        reward = self._task.get_reward(self._physics)
        observation = self._task.get_observation(self._physics)
        self._step_count += 1
        if self._step_count >= self._step_limit:
            discount = 1.0
        else:
            discount = self._task.get_termination(self._physics)
        if discount is not None:
            self._reset_next_step = True
            return TimeStep(StepType.LAST, reward, discount, observation)
        return TimeStep(StepType.MID, reward, 1.0, observation)

    def close(self):
        pass


# ------------------------------------------------------------------------------------------------ install + import
def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install():
    """Stand-ins into sys.modules, /root/reference onto sys.path, import the reference modules unmodified.
    Returns (gym_kmanip, env_base, env_sim, ik_mujoco)."""
    if not os.path.isdir(os.path.join(REFERENCE_ROOT, "gym_kmanip")):
        raise RuntimeError("the reference checkout is not present (build container only)")
    if "gym_kmanip" in sys.modules:
        import gym_kmanip
        return gym_kmanip, sys.modules["gym_kmanip.env_base"], sys.modules["gym_kmanip.env_sim"], sys.modules["gym_kmanip.ik_mujoco"]
    sys.dont_write_bytecode = True          # nothing may be written under /root/reference
    spaces = _module("gymnasium.spaces", Box=_Box, Dict=_DictSpace, Space=object)
    registration = _module("gymnasium.envs.registration", register=_register)
    envs = _module("gymnasium.envs", registration=registration)
    _module("gymnasium", Env=_Env, spaces=spaces, envs=envs, register=_register)
    _module("dm_env", TimeStep=TimeStep, StepType=StepType)
    mj = _module("dm_control.mujoco", Physics=Physics, mj_kinematics=_mj_kinematics, mj_comPos=_mj_comPos,
                 mj_jacSite=_mj_jacSite, mju_mat2Quat=_mju_mat2Quat, mju_subQuat=_mju_subQuat, mjd_subQuat=_mjd_subQuat)
    base = _module("dm_control.suite.base", Task=Task)
    suite = _module("dm_control.suite", base=base)
    control = _module("dm_control.rl.control", Environment=Environment)
    rl = _module("dm_control.rl", control=control)
    _module("dm_control", mujoco=mj, suite=suite, rl=rl)
    _module("h5py", File=h5_recorder.File, Group=h5_recorder.Group, Dataset=h5_recorder.Dataset)
    sys.path.insert(0, REFERENCE_ROOT)
    import gym_kmanip
    from gym_kmanip import env_base, env_sim, ik_mujoco

    # observer: nfev / status of every least_squares call of ik() (ik_mujoco.py:129-135); the call itself is untouched
    real_ls = ik_mujoco.least_squares

    def observed_least_squares(*a, **kw):
        try:
            r = real_ls(*a, **kw)
        except ValueError:
            IK_LOG.append((0, -2))
            raise
        IK_LOG.append((int(r.nfev), int(r.status)))
        return r

    ik_mujoco.least_squares = observed_least_squares
    return gym_kmanip, env_base, env_sim, ik_mujoco


def make_env(env_id, log_dir_root=None, **overrides):
    """gymnasium.make(env_id) minus the wrappers: the reference's KManipEnv built from its own registered kwargs.
    log_dir_root: where `log_h5py=True` may create its log directory (replaces the package-relative DATA_DIR)."""
    k, env_base, _, _ = install()
    if overrides.get("log_h5py") or overrides.get("log_rerun"):
        assert log_dir_root is not None and not os.path.abspath(log_dir_root).startswith(REFERENCE_ROOT)
        k.DATA_DIR = log_dir_root
    spec = REGISTRY[env_id]
    assert spec["entry_point"] == "gym_kmanip.env_base:KManipEnv"
    kw = dict(spec["kwargs"])
    kw.update(overrides)
    return env_base.KManipEnv(**kw)
