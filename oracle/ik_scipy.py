"""NumPy restatement of ik_mujoco.py driven by the REAL scipy.optimize.least_squares.

TEST INFRASTRUCTURE ONLY (oracle).  This is the half of the reference's arithmetic that can be
reproduced in the build container: SciPy is installed, so the exact TRF solver the reference
calls (reference gym_kmanip/ik_mujoco.py:129-135) and scipy.spatial.transform.Rotation (the
euler/quaternion decode of env_sim.py:66-69) run here.  MuJoCo is not installed, so its helper
calls (mj_kinematics, mj_jacSite, mju_mat2Quat, mju_subQuat, mjd_subQuat) are restated in NumPy
from the public MuJoCo source/documentation.  It pins the C oracle's IK (tests/test_oracle_ik.py)
and produces tests/golden/ik_*.npz through tests/tools/make_golden.py.

Functions follow the reference one-to-one:
  ik_res  <- ik_mujoco.py:20-53      ik_jac <- ik_mujoco.py:56-97      ik <- ik_mujoco.py:100-155
"""
from __future__ import annotations

import math
from functools import partial

import numpy as np
from scipy.optimize import least_squares
from scipy.spatial.transform import Rotation as R

XYZW_2_WXYZ = np.array([3, 0, 1, 2])  # __init__.py:212


# ----------------------------------------------------------------------------- quaternion utils
def qmul(a, b):
    return np.array([
        a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
        a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
        a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1],
        a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]])


def quat2mat(q):
    w, x, y, z = q
    return np.array([
        [w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
        [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
        [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z]])


def mju_mat2quat(m):
    m = np.asarray(m).reshape(9)
    q = np.zeros(4)
    if m[0] + m[4] + m[8] > 0:
        q[0] = 0.5 * math.sqrt(1 + m[0] + m[4] + m[8])
        q[1] = 0.25 * (m[7] - m[5]) / q[0]; q[2] = 0.25 * (m[2] - m[6]) / q[0]; q[3] = 0.25 * (m[3] - m[1]) / q[0]
    elif m[0] > m[4] and m[0] > m[8]:
        q[1] = 0.5 * math.sqrt(1 + m[0] - m[4] - m[8])
        q[0] = 0.25 * (m[7] - m[5]) / q[1]; q[2] = 0.25 * (m[1] + m[3]) / q[1]; q[3] = 0.25 * (m[2] + m[6]) / q[1]
    elif m[4] > m[8]:
        q[2] = 0.5 * math.sqrt(1 - m[0] + m[4] - m[8])
        q[0] = 0.25 * (m[2] - m[6]) / q[2]; q[1] = 0.25 * (m[1] + m[3]) / q[2]; q[3] = 0.25 * (m[5] + m[7]) / q[2]
    else:
        q[3] = 0.5 * math.sqrt(1 - m[0] - m[4] + m[8])
        q[0] = 0.25 * (m[3] - m[1]) / q[3]; q[1] = 0.25 * (m[2] + m[6]) / q[3]; q[2] = 0.25 * (m[5] + m[7]) / q[3]
    return q / np.linalg.norm(q)


def mju_subquat(qa, qb):
    qdif = qmul(np.array([qb[0], -qb[1], -qb[2], -qb[3]]), qa)
    axis = qdif[1:].copy()
    n = np.linalg.norm(axis)
    if n < 1e-15:
        axis = np.array([1.0, 0, 0]); n_ret = 0.0
    else:
        axis /= n; n_ret = n
    speed = 2 * math.atan2(n_ret, qdif[0])
    if speed > math.pi:
        speed -= 2 * math.pi
    return axis * speed


def mjd_subquat_b(qa, qb):
    axis = mju_subquat(qa, qb)
    n = np.linalg.norm(axis)
    if n < 1e-15:
        axis = np.array([1.0, 0, 0]); n = 0.0
    else:
        axis = axis / n
    half = 0.5 * n
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    coef = 1.0 - (1.0 if half < 6e-8 else half / math.tan(half))
    Da = np.eye(3) + half * K + coef * (K @ K)
    return -Da.T


# ----------------------------------------------------------------------------- kinematics (asset JSON)
class NumpyArm:
    """Kinematic tree straight from the build-owned asset JSON (not from KModelDesc)."""

    def __init__(self, asset: dict):
        self.links = asset["links"]
        self.nl = len(self.links)
        self.sites = asset["sites"]

    def fk(self, qpos):
        nl = self.nl
        xpos = np.zeros((nl, 3)); xquat = np.zeros((nl, 4)); axis = np.zeros((nl, 3))
        for i, l in enumerate(self.links):
            p = l["parent"]
            pos = np.array(l["pos"], dtype=float); quat = np.array(l["quat"], dtype=float)
            if p >= 0:
                pos = xpos[p] + quat2mat(xquat[p]) @ pos
                quat = qmul(xquat[p], quat)
            ax_local = np.array(l["joint"]["axis"], dtype=float)
            axis[i] = quat2mat(quat) @ ax_local
            if l["joint"]["type"] == "slide":
                pos = pos + axis[i] * qpos[i]
            else:
                a = qpos[i]
                quat = qmul(quat, np.concatenate([[math.cos(a / 2)], ax_local * math.sin(a / 2)]))
            quat = quat / np.linalg.norm(quat)
            xpos[i] = pos; xquat[i] = quat
        return xpos, xquat, axis

    def site(self, name, xpos, xquat):
        s = self.sites[name]
        l = s["link"]
        pos = xpos[l] + quat2mat(xquat[l]) @ np.array(s["pos"])
        q = qmul(xquat[l], np.array(s["quat"]))
        q = q / np.linalg.norm(q)
        return pos, quat2mat(q)

    def jac_site(self, name, xpos, axis, point):
        nl = self.nl
        jacp = np.zeros((3, nl)); jacr = np.zeros((3, nl))
        j = self.sites[name]["link"]
        while j >= 0:
            if self.links[j]["joint"]["type"] == "slide":
                jacp[:, j] = axis[j]
            else:
                jacp[:, j] = np.cross(axis[j], point - xpos[j])
                jacr[:, j] = axis[j]
            j = self.links[j]["parent"]
        return jacp, jacr


class FakePhysics:
    """Holds the mutable qpos the reference's IK writes into (ik_mujoco.py:34,67)."""

    def __init__(self, arm: NumpyArm, qpos, jnt_range):
        self.arm = arm
        self.qpos = np.array(qpos, dtype=float)
        self.jnt_range = np.asarray(jnt_range, dtype=float)


# ----------------------------------------------------------------------------- reference functions
def ik_res(q_pos, physics=None, goal_pos=None, goal_orn=None, q_mask=None, q_pos_home=None, q_pos_prev=None,
           ee_site=None, rad=0.02, reg_home=2e-6, reg_prev=6e-3):
    physics.qpos[q_mask] = q_pos
    xpos, xquat, axis = physics.arm.fk(physics.qpos)
    ee_pos, ee_mat = physics.arm.site(ee_site, xpos, xquat)
    res_pos = ee_pos - goal_pos
    curr_quat = mju_mat2quat(ee_mat)
    res_quat = mju_subquat(np.asarray(goal_orn).flatten(), curr_quat) * rad
    res_reg_home = reg_home * (q_pos - q_pos_home)
    res_reg_prev = reg_prev * (q_pos - q_pos_prev)
    return np.hstack((res_pos.flatten(), res_quat, res_reg_prev, res_reg_home))


def ik_jac(q_pos, physics=None, goal_orn=None, q_mask=None, ee_site=None, rad=0.02, reg=9e-3):
    physics.qpos[q_mask] = q_pos
    xpos, xquat, axis = physics.arm.fk(physics.qpos)
    ee_pos, ee_mat = physics.arm.site(ee_site, xpos, xquat)
    jac_pos, jac_quat = physics.arm.jac_site(ee_site, xpos, axis, ee_pos)
    ee_orn = mju_mat2quat(ee_mat)
    D_ee = mjd_subquat_b(goal_orn, ee_orn)
    mat = rad * D_ee.T @ ee_mat.T
    jac_quat = mat @ jac_quat
    nv = physics.arm.nl
    jac_reg = reg * np.eye(nv)
    jac_pos = jac_pos[:, q_mask]
    jac_quat = jac_quat[:, q_mask]
    jac_reg = jac_reg[q_mask, :][:, q_mask]
    return np.vstack((jac_pos, jac_quat, jac_reg, jac_reg))


def ik(physics, goal_pos=None, goal_orn=None, q_mask=None, q_pos_home=None, q_pos_prev=None, ee_site=None):
    """Returns (q_pos, result or None).  physics.qpos is left mutated like the reference."""
    q_mask = np.asarray(q_mask)
    q_pos = physics.qpos[q_mask].copy()
    ik_func = partial(ik_res, physics=physics, goal_pos=goal_pos, goal_orn=goal_orn,
                      q_pos_home=q_pos_home[q_mask], q_pos_prev=q_pos_prev[q_mask], q_mask=q_mask, ee_site=ee_site)
    ik_jac_func = partial(ik_jac, physics=physics, goal_orn=goal_orn, q_mask=q_mask, ee_site=ee_site)
    result = None
    try:
        result = least_squares(ik_func, q_pos, jac=ik_jac_func,
                               bounds=(physics.jnt_range[q_mask, 0], physics.jnt_range[q_mask, 1]), verbose=0)
        q_pos = result.x
    except ValueError:
        pass
    q_pos = np.clip(q_pos, physics.jnt_range[q_mask, 0], physics.jnt_range[q_mask, 1])
    return q_pos, result


def euler_goal(site_mat, delta):
    """env_sim.py:66-69: euler('xyz') of the site matrix + delta -> wxyz quaternion."""
    e = np.asarray(delta, dtype=float) + R.from_matrix(np.asarray(site_mat).reshape(3, 3)).as_euler("xyz")
    return R.from_euler("xyz", e).as_quat()[XYZW_2_WXYZ]
