"""Shared by tests/tools/make_golden_mujoco.py (writes tests/golden/mj_<asset>.npz where `import mujoco` works) and
tests/test_mujoco_pin.py (compares the oracle and the HIP engine with those files): what a fixture holds, how a MuJoCo contact
list becomes the build's KM_CON_* mask, and how MuJoCo's constraint rows are matched with the oracle's.

Why it exists: MuJoCo's mj_step under the path (CRBA, RNE, contact rows, soft-constraint constants, Euler,
the legacy step order) is the one part of the path NO fixture pins -- `mujoco` is absent from the build image and from the GPU
box.  These files make the pin a one-command act on the first box that has it (see make_golden_mujoco.py).

A fixture (one per surrogate model: solo_arm, dual_arm, torso; all through a joint-delta "QPos" action mode, which needs no IK):
  meta                  JSON: engine ("mujoco"; "fake" = the oracle-backed stand-in of the plumbing test -- never a pin),
                        engine version, generator version, asset, act_list, nstate
  qpos, qvel, ctrl, warm      [S, nq | nv | nu | nv]   the seeded states (state BEFORE a control step; ctrl = the previous step's)
  qM [S, nv, nv], qfrc_bias, qacc_smooth, qacc [S, nv]   mj_forward at the state with data.ctrl = ctrl
  nefc [S], efc_J [S, E, nv], efc_R, efc_aref [S, E], efc_type [S, E]   (rows beyond nefc: zero; mjtConstraint codes)
  mask [S]              KM_CON_* bits of data.contact at the state (contacts_to_mask below)
  action [S, act_dim] f32, ctrl_set [S, nu]      the control step's action and what before_step hands to set_control
  post_qpos, post_qvel, post_warm [S, ...], post_mask [S]   after `mj_step2; 9 x mj_step; mj_step1` (dm_control's legacy order)
"""
import json

import numpy as np

GENERATOR_VERSION = 1
ASSETS = ("solo_arm", "dual_arm", "torso")
MJ_CNSTR_FRICTION_DOF, MJ_CNSTR_LIMIT_JOINT, MJ_CNSTR_CONTACT_PYRAMIDAL = 1, 3, 5        # mjtConstraint


def qpos_spec(asset):
    """An EnvSpec of `asset` in the reference's joint-delta action mode (env_sim.py:100-103; the Torso has no registered
    *QPos id, but the seam takes any act_list): ctrl = qpos[mask] + 0.1 a, no IK, so a control step is before_step's two lines
    + Physics.step(10)."""
    from gym_kmanip_amd.model import ENV_SPECS, EnvSpec
    base = {"solo_arm": "KManipSoloArm", "dual_arm": "KManipDualArm", "torso": "KManipTorso"}[asset]
    s = ENV_SPECS[base]
    acts = ["q_pos_r", "grip_r"] if s.q_id_l_mask is None else ["q_pos_r", "q_pos_l", "grip_l", "grip_r"]
    return EnvSpec("custom-%s-qpos" % asset, asset=s.asset, obs_list=list(s.obs_list), act_list=acts, q_pos_home=s.q_pos_home,
                   q_id_r_mask=s.q_id_r_mask, q_id_l_mask=s.q_id_l_mask, ctrl_id_r_grip=s.ctrl_id_r_grip, ctrl_id_l_grip=s.ctrl_id_l_grip)


def contacts_to_mask(contacts, cube_pos, cube_mat, sphere_names):
    """KM_CON_* mask (include/kmanip.h:70-72) of a MuJoCo contact list.  contacts: iterable of (geom1 name, geom2 name, pos[3]);
    sphere_names: collider names in sphere-index order (the exported geoms are `<name>`, `<name>__seg` for a link-capsule's end
    sphere and `<name>__capsule` for its segment; tools/mjcf_export.py).  A table-cube contact sets the bit of the cube corner
    nearest to it (corner i: bit 0 = +x, bit 1 = +y, bit 2 = +z of the cube frame, the oracle's `collide`)."""
    idx = {}
    for s, n in enumerate(sphere_names):
        idx[n] = idx[n + "__seg"] = idx[n + "__capsule"] = s
    mask = 0
    for g1, g2, pos in contacts:
        pair = {g1, g2}
        if pair == {"table", "cube"}:
            loc = np.asarray(cube_mat, dtype=np.float64).reshape(3, 3).T @ (np.asarray(pos, dtype=np.float64) - np.asarray(cube_pos, dtype=np.float64))
            mask |= 1 << (int(loc[0] > 0) | int(loc[1] > 0) << 1 | int(loc[2] > 0) << 2)
        elif "cube" in pair:
            (other,) = pair - {"cube"}
            mask |= 1 << (8 + idx[other])
        elif "table" in pair:
            (other,) = pair - {"table"}
            mask |= 1 << (20 + idx[other])
        else:
            raise ValueError("a contact pair the surrogate does not have: %s / %s" % (g1, g2))
    return mask


def match_rows(J_a, J_b, tol=1e-6):
    """Constraint rows are a SET as far as the primal cost goes (and contact order is the engine's business): a bijection
    a -> b that pairs every row of J_a with its nearest row of J_b (None if the counts differ or a row has no partner within
    tol, relative to the row's norm)."""
    J_a, J_b = np.asarray(J_a), np.asarray(J_b)
    if J_a.shape != J_b.shape:
        return None
    free = list(range(len(J_b)))
    perm = []
    for r in J_a:
        if not free:
            return None
        d = [np.abs(J_b[j] - r).max() for j in free]
        k = int(np.argmin(d))
        if d[k] > tol * max(1.0, np.abs(r).max()):
            return None
        perm.append(free.pop(k))
    return np.array(perm, dtype=np.int64)


def pack_meta(**kw):
    return np.array(json.dumps(dict(kw, generator_version=GENERATOR_VERSION)))


def read_meta(npz):
    return json.loads(str(npz["meta"]))
