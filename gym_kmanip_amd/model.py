"""Host-side model compiler: asset JSON + env config -> KModelDesc (include/kmanip.h).

Mirrors what `env_sim.new(gym_env)` reads from the gym env (reference env_sim.py:26-27,45,
50-51,76-77,112,140,154,208-209: mjcf_filename, q_len, q_pos_home, q_id_r_mask, q_id_l_mask,
ctrl_id_r_grip, ctrl_id_l_grip, obs_list, act_list) and the module constants of
gym_kmanip/__init__.py:28-41,164-208.  No reference file is read at run time: the tree comes
from the build-owned JSON written by tools/mjcf_extract.py.
"""
from __future__ import annotations

import ctypes as C
import json
import math
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

ASSETS_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")

KM_MAX_LINKS = 20
KM_MAX_ARMS = 2
KM_MAX_IK = 7
KM_MAX_SPHERES = 12
KM_MAX_CAMS = 4
KM_CAM_INDEX = {"grip_r": 0, "grip_l": 1, "top": 2, "head": 3}


class Cam:
    """gym_kmanip/__init__.py:143-161: image size / channels / value range of one camera (fl, pp are logging metadata there)."""

    def __init__(self, w, h, name, fl=0, pp=(0, 0)):
        self.w, self.h, self.c, self.name, self.log_name = w, h, 3, name, "camera/" + name
        self.fl, self.pp = fl, tuple(pp)
        self.low, self.high, self.dtype = 0, 255, np.uint8

    def __repr__(self):
        return "Cam(%s %dx%d)" % (self.name, self.w, self.h)


CAMERAS = {"head": Cam(640, 480, "head", 448, (320, 240)), "top": Cam(640, 480, "top", 448, (320, 240)),
           "grip_r": Cam(60, 40, "grip_r", 45, (30, 20)), "grip_l": Cam(60, 40, "grip_l", 45, (30, 20))}
KM_ACT_KEYS = ["eel_pos", "eel_orn", "eer_pos", "eer_orn", "grip_l", "grip_r", "q_pos_r", "q_pos_l"]
KM_DONE_TRUNCATED = 1
KM_DONE_DIVERGED = 2

# ---- constants of gym_kmanip/__init__.py (values restated; line numbers cite the reference)
MAX_EPISODE_STEPS = 64          # :28
CONTROL_TIMESTEP = 0.02         # :30
MAX_Q_VEL = math.pi             # :31
CTRL_ALPHA = 1.0                # :34 (identity filter; folded away)
IK_RES_RAD = 0.02               # :37
IK_RES_REG_PREV = 6e-3          # :38
IK_RES_REG_HOME = 2e-6          # :39
IK_JAC_RAD = 0.02               # :40
IK_JAC_REG = 9e-3               # :41
CUBE_SPAWN_RANGE = np.array([[0.1, 0.3], [0.5, 0.7], [0.6, 0.7]])  # :164-170
EE_POS_DELTA = [0.01, 0.01, 0.01]   # :174-180
EE_ORN_DELTA = [0.1, 0.1, 0.1]      # :181-187
EPSILON = 1e-6                  # :192
Q_POS_DELTA = 0.1               # :196
EE_S_MIN = -0.029               # :199
EE_S_MAX = 0.005                # :200
EE_S_DELTA = 0.0001             # :201
REWARD_SUCCESS_THRESHOLD = 2.0  # :204
REWARD_VEL_PENALTY = 0.01       # :205
REWARD_GRIP_DIST = 0.01         # :206
REWARD_TOUCH_CUBE = 1.0         # :207
REWARD_LIFT_CUBE = 1.0          # :208

# MuJoCo defaults (no <option>/<default> element exists in any reference XML)
MJ_TIMESTEP = 0.002
MJ_DEFAULT_SOLREF = [0.02, 1.0]
MJ_DEFAULT_SOLIMP = [0.9, 0.95, 0.001, 0.5, 2.0]
MJ_DEFAULT_FRICTION = [1.0, 0.005, 0.0001]

# ---- home poses, __init__.py:53-122 (float32 like ACT_DTYPE there)
Q_SOLO_ARM_HOME = np.array([0.0, 0.75, 1.0, 1.0, 2.0, -2.0, 0.0, 0.0, 0.005, 0.005], dtype=np.float32)
Q_DUAL_ARM_HOME = np.array([0.0, 0.75, 1.0, 1.0, 2.0, -2.7, 0.0, 0.0, 0.005, 0.005,
                            0.0, -0.75, -1.0, -1.0, 2.0, 0.0, 0.0, 0.0, 0.005, 0.005], dtype=np.float32)
Q_TORSO_HOME = np.array([-1.0, 0.0, 1.7, 1.6, 0.34, 1.6, 1.4, -0.26, 0.0, 0.0, 0.0,
                         -1.7, -1.6, -0.34, -1.6, -1.4, -1.7, 0.0, 0.0, 0.0], dtype=np.float32)

OBS_STATE = ["q_pos", "q_vel", "cube_pos", "cube_orn"]


@dataclass
class EnvSpec:
    """kwargs of one `gym.register` call, gym_kmanip/__init__.py:244-483."""
    env_id: str
    asset: str
    obs_list: List[str]
    act_list: List[str]
    q_pos_home: np.ndarray
    q_id_r_mask: Optional[List[int]] = None
    q_id_l_mask: Optional[List[int]] = None
    ctrl_id_r_grip: Optional[List[int]] = None
    ctrl_id_l_grip: Optional[List[int]] = None
    max_episode_steps: int = MAX_EPISODE_STEPS


_SOLO = dict(asset="solo_arm", q_pos_home=Q_SOLO_ARM_HOME, q_id_r_mask=[0, 1, 2, 3, 4, 5, 6],
             ctrl_id_r_grip=[8, 9])
_DUAL = dict(asset="dual_arm", q_pos_home=Q_DUAL_ARM_HOME, q_id_r_mask=[0, 1, 2, 3, 4, 5, 6],
             q_id_l_mask=[10, 11, 12, 13, 14, 15, 16], ctrl_id_r_grip=[8, 9], ctrl_id_l_grip=[18, 19])
_TORSO = dict(asset="torso", q_pos_home=Q_TORSO_HOME, q_id_r_mask=[2, 3, 4, 5, 6, 7],
              q_id_l_mask=[11, 12, 13, 14, 15, 16], ctrl_id_r_grip=[8, 9], ctrl_id_l_grip=[17, 18])
_ACT_SOLO = ["eer_pos", "eer_orn", "grip_r"]
_ACT_BOTH = ["eel_pos", "eel_orn", "eer_pos", "eer_orn", "grip_l", "grip_r"]

ENV_SPECS: Dict[str, EnvSpec] = {
    "KManipSoloArm": EnvSpec("KManipSoloArm", obs_list=OBS_STATE, act_list=_ACT_SOLO, **_SOLO),
    "KManipSoloArmQPos": EnvSpec("KManipSoloArmQPos", obs_list=OBS_STATE, act_list=["q_pos_r", "grip_r"], **_SOLO),
    "KManipSoloArmVision": EnvSpec("KManipSoloArmVision", obs_list=["q_pos", "q_vel", "camera/head", "camera/grip_r"],
                                   act_list=_ACT_SOLO, **_SOLO),
    "KManipDualArm": EnvSpec("KManipDualArm", obs_list=OBS_STATE, act_list=_ACT_BOTH, **_DUAL),
    "KManipDualArmQPos": EnvSpec("KManipDualArmQPos", obs_list=OBS_STATE,
                                 act_list=["q_pos_r", "q_pos_l", "grip_l", "grip_r"], **_DUAL),
    "KManipDualArmVision": EnvSpec("KManipDualArmVision",
                                   obs_list=["q_pos", "q_vel", "camera/head", "camera/grip_l", "camera/grip_r"],
                                   act_list=_ACT_BOTH, **_DUAL),
    "KManipTorso": EnvSpec("KManipTorso", obs_list=OBS_STATE, act_list=_ACT_BOTH, **_TORSO),
    "KManipTorsoVision": EnvSpec("KManipTorsoVision",
                                 obs_list=["q_pos", "q_vel", "camera/head", "camera/grip_l", "camera/grip_r"],
                                 act_list=_ACT_BOTH, **_TORSO),
}


class KModelDesc(C.Structure):
    """ctypes mirror of `struct KModelDesc` in include/kmanip.h (field order matters)."""
    _fields_ = [
        ("nlink", C.c_int32), ("narm", C.c_int32), ("nsphere", C.c_int32), ("act_dim", C.c_int32),
        ("obs_dim", C.c_int32), ("max_episode_steps", C.c_int32), ("n_sub_steps", C.c_int32),
        ("solver_iterations", C.c_int32), ("touch_reward_enabled", C.c_int32), ("auto_reset", C.c_int32),
        ("act_col", C.c_int32 * len(KM_ACT_KEYS)), ("solver", C.c_int32), ("ik_max_nfev", C.c_int32), ("pad0_", C.c_int32),
        ("link_parent", C.c_int32 * KM_MAX_LINKS), ("jnt_type", C.c_int32 * KM_MAX_LINKS),
        ("forcelimited", C.c_int32 * KM_MAX_LINKS), ("pad1_", C.c_int32 * KM_MAX_LINKS),
        ("link_pos", (C.c_double * 3) * KM_MAX_LINKS), ("link_quat", (C.c_double * 4) * KM_MAX_LINKS),
        ("jnt_axis", (C.c_double * 3) * KM_MAX_LINKS), ("jnt_range", (C.c_double * 2) * KM_MAX_LINKS),
        ("frictionloss", C.c_double * KM_MAX_LINKS), ("kp", C.c_double * KM_MAX_LINKS),
        ("ctrlrange", (C.c_double * 2) * KM_MAX_LINKS), ("forcerange", (C.c_double * 2) * KM_MAX_LINKS),
        ("mass", C.c_double * KM_MAX_LINKS), ("com", (C.c_double * 3) * KM_MAX_LINKS),
        ("inertia", (C.c_double * 3) * KM_MAX_LINKS), ("q_home", C.c_double * KM_MAX_LINKS),
        ("arm_present", C.c_int32 * KM_MAX_ARMS), ("arm_nq", C.c_int32 * KM_MAX_ARMS),
        ("arm_q_id", (C.c_int32 * (KM_MAX_IK + 1)) * KM_MAX_ARMS), ("arm_grip_id", (C.c_int32 * 2) * KM_MAX_ARMS),
        ("arm_site_link", C.c_int32 * KM_MAX_ARMS), ("arm_mode", C.c_int32 * KM_MAX_ARMS),
        ("arm_has_grip", C.c_int32 * KM_MAX_ARMS), ("pad2_", C.c_int32 * 2),
        ("arm_site_pos", (C.c_double * 3) * KM_MAX_ARMS), ("arm_site_quat", (C.c_double * 4) * KM_MAX_ARMS),
        ("sphere_link", C.c_int32 * KM_MAX_SPHERES), ("sphere_visible", C.c_int32 * KM_MAX_SPHERES), ("sphere_pos", (C.c_double * 3) * KM_MAX_SPHERES),
        ("sphere_radius", C.c_double * KM_MAX_SPHERES), ("sphere_seg", (C.c_double * 3) * KM_MAX_SPHERES), ("table_z", C.c_double), ("table_rect", C.c_double * 4),
        ("cube_mass", C.c_double), ("cube_inertia", C.c_double * 3), ("cube_half", C.c_double * 3),
        ("cube_frictionloss", C.c_double), ("cube_quat0", C.c_double * 4),
        ("cube_spawn_lo", C.c_double * 3), ("cube_spawn_hi", C.c_double * 3),
        ("con_cube_solref", C.c_double * 2), ("con_cube_solimp", C.c_double * 5),
        ("con_cube_friction", C.c_double * 3), ("con_def_solref", C.c_double * 2),
        ("con_def_solimp", C.c_double * 5), ("con_def_friction", C.c_double * 3),
        ("timestep", C.c_double), ("gravity", C.c_double * 3), ("solver_tolerance", C.c_double),
        ("ik_res_rad", C.c_double), ("ik_res_reg_prev", C.c_double), ("ik_res_reg_home", C.c_double),
        ("ik_jac_rad", C.c_double), ("ik_jac_reg", C.c_double),
        ("ee_pos_delta", C.c_double * 3), ("ee_orn_delta", C.c_double * 3), ("q_pos_delta", C.c_double),
        ("ee_s_min", C.c_double), ("ee_s_max", C.c_double), ("ee_s_delta", C.c_double),
        ("max_q_vel", C.c_double), ("epsilon", C.c_double),
        ("reward_vel_penalty", C.c_double), ("reward_grip_dist", C.c_double),
        ("reward_touch_cube", C.c_double), ("reward_lift_cube", C.c_double),
        ("cam_present", C.c_int32 * KM_MAX_CAMS), ("cam_link", C.c_int32 * KM_MAX_CAMS),
        ("cam_target_link", C.c_int32 * KM_MAX_CAMS),
        ("cam_pos", (C.c_double * 3) * KM_MAX_CAMS), ("cam_target_pos", (C.c_double * 3) * KM_MAX_CAMS),
        ("cam_fovy", C.c_double * KM_MAX_CAMS), ("cam_znear", C.c_double), ("cam_zfar", C.c_double),
        ("dof_invweight0", C.c_double * KM_MAX_LINKS), ("body_invweight0", (C.c_double * 2) * KM_MAX_LINKS),
        ("cube_invweight0", C.c_double * 2), ("meaninertia", C.c_double),
    ]


def load_asset(name: str) -> dict:
    with open(os.path.join(ASSETS_DIR, name + ".json")) as f:
        return json.load(f)


def _mix(a, b, n):
    """MuJoCo contact parameter mixing with equal solmix (mix = 0.5)."""
    a = list(a) + list(MJ_DEFAULT_SOLIMP[len(a):n]) if len(a) < n else list(a)
    return [0.5 * x + 0.5 * y for x, y in zip(a[:n], b[:n])]


@dataclass
class CompiledModel:
    spec: EnvSpec
    asset: dict
    desc: KModelDesc
    nlink: int
    nq: int
    nv: int
    nu: int
    act_dim: int
    obs_dim: int
    act_slices: Dict[str, slice] = field(default_factory=dict)
    obs_slices: Dict[str, slice] = field(default_factory=dict)
    cameras: List[str] = field(default_factory=list)


SOLVERS = {"pgs": 0, "newton": 1}


def _quat2mat(q):
    w, x, y, z = np.asarray(q, dtype=np.float64) / np.linalg.norm(q)
    return np.array([[w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z]])


def invweight0(d: "KModelDesc"):
    """What MuJoCo's mj_setConst computes at qpos0 (every joint at 0; no `ref`/`springref` in any reference XML) and the
    constraint code then uses for efc_diagApprox and the solver scale -- from the model's own (surrogate) inertias:
      dof_invweight0[i]   = (M^-1)_ii
      body_invweight0[b]  = (mean of the translational, mean of the rotational) diagonal of J_b M^-1 J_b^T, J_b = the
                            6 x nv Jacobian of body b at its centre of mass
      cube (free body)    = (1/m, mean_k 1/I_k); free-joint dof_invweight0 is averaged per 3-dof block the same way
      meaninertia         = trace(M) / nv over all nv = nlink + 6 dofs
    Returns (dofw[nl], bodyw[nl][2], cubew[2], meaninertia)."""
    nl = d.nlink
    xpos = np.zeros((nl, 3)); xmat = np.zeros((nl, 3, 3)); axis = np.zeros((nl, 3)); cpos = np.zeros((nl, 3))
    for i in range(nl):
        p = d.link_parent[i]
        R0 = _quat2mat(list(d.link_quat[i]))
        lp = np.array(list(d.link_pos[i]))
        if p < 0:
            xpos[i], xmat[i] = lp, R0
        else:
            xpos[i], xmat[i] = xpos[p] + xmat[p] @ lp, xmat[p] @ R0          # joint value 0: no joint motion
        axis[i] = xmat[i] @ np.array(list(d.jnt_axis[i]))
        cpos[i] = xpos[i] + xmat[i] @ np.array(list(d.com[i]))
    anc = []
    for i in range(nl):
        a, j = [], i
        while j >= 0:
            a.append(j); j = d.link_parent[j]
        anc.append(a)

    def jac(b, pt):
        J = np.zeros((6, nl))
        for j in anc[b]:
            if d.jnt_type[j] == 1:
                J[:3, j] = axis[j]
            else:
                J[:3, j] = np.cross(axis[j], pt - xpos[j]); J[3:, j] = axis[j]
        return J

    M = np.zeros((nl, nl))
    Js = []
    for b in range(nl):
        J = jac(b, cpos[b]); Js.append(J)
        Iw = xmat[b] @ np.diag(list(d.inertia[b])) @ xmat[b].T
        M += d.mass[b] * J[:3].T @ J[:3] + J[3:].T @ Iw @ J[3:]
    Minv = np.linalg.inv(M)
    dofw = np.diag(Minv).copy()
    bodyw = np.zeros((nl, 2))
    for b in range(nl):
        A = Js[b] @ Minv @ Js[b].T
        bodyw[b] = [np.trace(A[:3, :3]) / 3.0, np.trace(A[3:, 3:]) / 3.0]
    cubew = (1.0 / d.cube_mass, float(np.mean([1.0 / d.cube_inertia[k] for k in range(3)])))
    trace = np.trace(M) + 3 * d.cube_mass + sum(d.cube_inertia[k] for k in range(3))
    return dofw, bodyw, cubew, float(trace / (nl + 6))


def compile_model(env_id_or_spec, *, auto_reset: bool = True, touch_reward: bool = False,
                  solver: str = "newton", solver_iterations: int = 100, solver_tolerance: float = 1e-8,
                  ik_max_nfev: int = 0) -> CompiledModel:
    spec = ENV_SPECS[env_id_or_spec] if isinstance(env_id_or_spec, str) else env_id_or_spec
    asset = load_asset(spec.asset)
    links = asset["links"]
    nl = len(links)
    assert nl <= KM_MAX_LINKS and nl == len(spec.q_pos_home)
    d = KModelDesc()
    d.nlink = nl
    d.max_episode_steps = spec.max_episode_steps
    d.n_sub_steps = int(round(CONTROL_TIMESTEP / MJ_TIMESTEP))
    d.solver = SOLVERS[solver]
    d.solver_iterations = solver_iterations
    d.ik_max_nfev = int(ik_max_nfev)      # 0 = the reference's least_squares default (100 n); > 0: opt-in cap (include/kmanip.h)
    d.solver_tolerance = solver_tolerance
    d.touch_reward_enabled = int(touch_reward)
    d.auto_reset = int(auto_reset)
    for i, l in enumerate(links):
        assert l["parent"] < i
        d.link_parent[i] = l["parent"]
        j = l["joint"]
        d.jnt_type[i] = 1 if j["type"] == "slide" else 0
        assert j["limited"]
        for k in range(3):
            d.link_pos[i][k] = l["pos"][k]
            d.jnt_axis[i][k] = j["axis"][k]
            d.com[i][k] = l["inertial"]["com"][k]
            d.inertia[i][k] = l["inertial"]["diaginertia"][k]
        for k in range(4):
            d.link_quat[i][k] = l["quat"][k]
        d.jnt_range[i][0], d.jnt_range[i][1] = j["range"]
        d.frictionloss[i] = j["frictionloss"]
        a = l["actuator"]
        d.kp[i] = a["kp"]
        d.ctrlrange[i][0], d.ctrlrange[i][1] = a["ctrlrange"]
        if a["forcerange"] is not None:
            d.forcelimited[i] = 1
            d.forcerange[i][0], d.forcerange[i][1] = a["forcerange"]
        d.mass[i] = l["inertial"]["mass"]
        d.q_home[i] = float(np.float32(spec.q_pos_home[i]))

    # ---- action layout: Dict-space insertion order, env_base.py:151-188
    col = 0
    act_slices = {}
    masks = {"q_pos_r": spec.q_id_r_mask, "q_pos_l": spec.q_id_l_mask}
    for k, key in enumerate(KM_ACT_KEYS):
        if key in spec.act_list:
            width = {"grip_l": 1, "grip_r": 1}.get(key, 3)
            if key in masks:
                width = len(masks[key])
            d.act_col[k] = col
            act_slices[key] = slice(col, col + width)
            col += width
        else:
            d.act_col[k] = -1
    d.act_dim = col
    d.obs_dim = 2 * nl + 7

    # ---- arms
    narm = 0
    for arm, (side, mask, grip, site) in enumerate([
            ("r", spec.q_id_r_mask, spec.ctrl_id_r_grip, "eer_site_pos"),
            ("l", spec.q_id_l_mask, spec.ctrl_id_l_grip, "eel_site_pos")]):
        if mask is None:
            continue
        d.arm_present[arm] = 1
        d.arm_nq[arm] = len(mask)
        assert len(mask) <= KM_MAX_IK
        for k, q in enumerate(mask):
            d.arm_q_id[arm][k] = q
        s = asset["sites"][site]
        d.arm_site_link[arm] = s["link"]
        for k in range(3):
            d.arm_site_pos[arm][k] = s["pos"][k]
        for k in range(4):
            d.arm_site_quat[arm][k] = s["quat"][k]
        if ("ee%s_pos" % side) in spec.act_list:
            assert ("ee%s_orn" % side) in spec.act_list
            d.arm_mode[arm] = 1
        elif ("q_pos_%s" % side) in spec.act_list:
            d.arm_mode[arm] = 2
        if ("grip_%s" % side) in spec.act_list:
            d.arm_has_grip[arm] = 1
            d.arm_grip_id[arm][0], d.arm_grip_id[arm][1] = grip
        narm += 1
    d.narm = narm

    # ---- colliders
    sph = asset["spheres"]
    assert len(sph) <= KM_MAX_SPHERES
    d.nsphere = len(sph)
    for i, s in enumerate(sph):
        d.sphere_link[i] = s["link"]
        d.sphere_visible[i] = s.get("visible", 1)
        d.sphere_radius[i] = s["radius"]
        for k in range(3):
            d.sphere_pos[i][k] = s["pos"][k]
            d.sphere_seg[i][k] = s.get("seg", (0.0, 0.0, 0.0))[k]
    d.table_z = asset["table"]["plane_z"]
    for k, v in enumerate(asset["table"].get("rect", (-np.inf, np.inf, -np.inf, np.inf))):    # x_lo, x_hi, y_lo, y_hi of the table top
        d.table_rect[k] = v
    cube = asset["cube"]
    d.cube_mass = cube["mass"]
    d.cube_frictionloss = cube["frictionloss"]
    for k in range(3):
        d.cube_inertia[k] = cube["diaginertia"][k]
        d.cube_half[k] = cube["half_size"][k]
        d.cube_spawn_lo[k] = CUBE_SPAWN_RANGE[k, 0]
        d.cube_spawn_hi[k] = CUBE_SPAWN_RANGE[k, 1]
    for k in range(4):
        d.cube_quat0[k] = cube["quat0"][k]
    assert cube["condim"] == 4
    # pair parameters: condim = max, friction = element-wise max, solref/solimp = equal-weight mix
    for k, v in enumerate(_mix(cube["solref"], MJ_DEFAULT_SOLREF, 2)):
        d.con_cube_solref[k] = v
    for k, v in enumerate(_mix(cube["solimp"], MJ_DEFAULT_SOLIMP, 5)):
        d.con_cube_solimp[k] = v
    for k in range(3):
        d.con_cube_friction[k] = max(cube["friction"][k], MJ_DEFAULT_FRICTION[k])
        d.con_def_friction[k] = MJ_DEFAULT_FRICTION[k]
    for k in range(2):
        d.con_def_solref[k] = MJ_DEFAULT_SOLREF[k]
    for k in range(5):
        d.con_def_solimp[k] = MJ_DEFAULT_SOLIMP[k]

    # ---- options / constants
    d.timestep = asset["option"]["timestep"]
    for k in range(3):
        d.gravity[k] = asset["option"]["gravity"][k]
        d.ee_pos_delta[k] = EE_POS_DELTA[k]
        d.ee_orn_delta[k] = EE_ORN_DELTA[k]
    d.ik_res_rad, d.ik_res_reg_prev, d.ik_res_reg_home = IK_RES_RAD, IK_RES_REG_PREV, IK_RES_REG_HOME
    d.ik_jac_rad, d.ik_jac_reg = IK_JAC_RAD, IK_JAC_REG
    d.q_pos_delta = Q_POS_DELTA
    d.ee_s_min, d.ee_s_max, d.ee_s_delta = EE_S_MIN, EE_S_MAX, EE_S_DELTA
    d.max_q_vel = MAX_Q_VEL
    d.epsilon = EPSILON
    d.reward_vel_penalty, d.reward_grip_dist = REWARD_VEL_PENALTY, REWARD_GRIP_DIST
    d.reward_touch_cube, d.reward_lift_cube = REWARD_TOUCH_CUBE, REWARD_LIFT_CUBE

    # ---- gripper cameras (targetbody mode); znear from scene.xml:5 (<map znear="0.1"/> x extent ~1 m)
    for cname, ci in KM_CAM_INDEX.items():
        cams = [c for c in asset["cameras"] if c["name"] == cname]
        if not cams:
            continue
        cam = cams[0]
        tgt = asset["targets"][cam["target"]]
        d.cam_present[ci] = 1
        d.cam_link[ci] = cam["link"]
        d.cam_target_link[ci] = tgt["link"]
        d.cam_fovy[ci] = cam["fovy"]
        for k in range(3):
            d.cam_pos[ci][k] = cam["pos"][k]
            d.cam_target_pos[ci][k] = tgt["pos"][k]
    d.cam_znear, d.cam_zfar = 0.01, 5.0

    # ---- qpos0-time constants (MuJoCo mj_setConst)
    dofw, bodyw, cubew, meaninertia = invweight0(d)
    for i in range(nl):
        d.dof_invweight0[i] = dofw[i]
        d.body_invweight0[i][0], d.body_invweight0[i][1] = bodyw[i]
    d.cube_invweight0[0], d.cube_invweight0[1] = cubew
    d.meaninertia = meaninertia

    obs_slices = {"q_pos": slice(0, nl), "q_vel": slice(nl, 2 * nl),
                  "cube_pos": slice(2 * nl, 2 * nl + 3), "cube_orn": slice(2 * nl + 3, 2 * nl + 7)}
    cams = [o.split("/")[-1] for o in spec.obs_list if o.startswith("camera/")]
    return CompiledModel(spec=spec, asset=asset, desc=d, nlink=nl, nq=nl + 7, nv=nl + 6, nu=nl,
                         act_dim=col, obs_dim=2 * nl + 7, act_slices=act_slices,
                         obs_slices=obs_slices, cameras=cams)
