#!/usr/bin/env python3
"""Generate tests/golden/ref_*.npz + ref_spaces.json by running the REFERENCE'S OWN PYTHON (tests/tools/refrun.py).

BUILD CONTAINER ONLY (needs /root/reference; nothing from it is copied -- the outputs are data: inputs + expected outputs).
    python tests/tools/make_golden_ref.py

Files (all under tests/golden/):
  ref_spaces.json          for the eight registered ids: the kwargs the reference registers (__init__.py:244-483), the
                           observation / action spaces KManipEnv builds (env_base.py:115-190), info keys, camera list
  ref_run_<id>.npz         one rollout of the reference's KManipEnv.reset()/step() per id: per step the state before the
                           step, the flat float32 action, what before_step left (ctrl handed to set_control, qpos after the
                           IK's teleport, mocap goal pose, nfev/status of each least_squares call) and what k_step returned
                           (obs, reward, terminated, sim_time), plus the state after the step
  ref_ik_<family>.npz      direct calls of ik_mujoco.ik / ik_res / ik_jac on seeded cases (the cases of make_golden.gen_ik:
                           random poses, a start on a bound, an infeasible start -> the "IK failed" branch)
  ref_obs_<family>.npz     get_observation / get_reward on seeded states that exercise every clip
  ref_h5_tree_<id>.json    (round 5) the HDF5 tree the reference's OWN logger builds -- log_h5py.new / cam / step / end
                           (log_h5py.py:13-61) running unmodified from inside KManipEnv(log_h5py=True).reset / step / close
                           (env_base.py:231-263) against the recording h5py stand-in tests/tools/h5_recorder.py: groups, attrs
                           (dtype, shape, value), datasets (shape, dtype, chunks), file name, flush count
  ref_scripted_<id>.npz    (round 5) the same episode: every step's action is action_space.sample() with eer_pos overwritten
                           by the move-toward-the-cube heuristic of examples/2_synthetic_data.py:28-41, evaluated on the
                           reference env object; per step the state before, the sampled flat action, the heuristic's float64
                           vector, what env.step returned (obs, reward, info) and the logger's datasets (`h5/...`); plus one
                           rigged state whose step returns reward > REWARD_SUCCESS_THRESHOLD (info["is_success"], env_base.py:250)
  ref_touch_<family>.npz   get_reward with finger geoms that carry the names env_sim.py:171-174 looks for (the reference's
                           meshes are unnamed, so its own touch / lift terms never fire: SURVEY A.5 #5) on states with a finger
                           on the cube, the cube on / off the table
"""
import collections
import contextlib
import io
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import refrun  # noqa: E402
from gym_kmanip_amd.model import compile_model  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
FAMILY = {"KManipSoloArm": "solo", "KManipDualArm": "dual", "KManipTorso": "torso"}
ARMS = {"KManipSoloArm": [(0, "eer_site_pos")], "KManipDualArm": [(0, "eer_site_pos"), (1, "eel_site_pos")],
        "KManipTorso": [(0, "eer_site_pos"), (1, "eel_site_pos")]}


def quiet():
    return contextlib.redirect_stdout(io.StringIO())          # ik() prints per call (ik_mujoco.py:154)


def jsonable(v):
    if isinstance(v, np.ndarray):
        return v.tolist()
    if isinstance(v, (np.floating, np.integer)):
        return v.item()
    if isinstance(v, dict):
        return {k: jsonable(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [jsonable(x) for x in v]
    return v


def flat_action(cm, a):
    row = np.zeros(cm.act_dim, dtype=np.float32)
    for key, v in a.items():
        row[cm.act_slices[key]] = v
    return row


def flat_obs(cm, obs):
    return np.concatenate([np.asarray(obs[k], dtype=np.float64) for k in ("q_pos", "q_vel", "cube_pos", "cube_orn") if k in obs])


def gen_spaces():
    k, env_base, env_sim, ikm = refrun.install()
    out = {}
    for env_id, reg in refrun.REGISTRY.items():
        with quiet():
            env = refrun.make_env(env_id)
        kw = reg["kwargs"]
        out[env_id] = dict(
            max_episode_steps=reg["max_episode_steps"], nondeterministic=reg["nondeterministic"],
            entry_point=reg["entry_point"],
            mjcf_filename=kw["mjcf_filename"], obs_list=kw["obs_list"], act_list=kw["act_list"],
            q_pos_home=jsonable(np.asarray(kw["q_pos_home"], dtype=np.float64)), q_pos_home_dtype=str(np.asarray(kw["q_pos_home"]).dtype),
            q_keys=list(kw["q_keys"]),
            q_id_r_mask=jsonable(kw.get("q_id_r_mask")), q_id_l_mask=jsonable(kw.get("q_id_l_mask")),
            ctrl_id_r_grip=jsonable(kw.get("ctrl_id_r_grip")), ctrl_id_l_grip=jsonable(kw.get("ctrl_id_l_grip")),
            q_len=env.q_len, action_len=env.action_len,
            observation_space={n: dict(low=jsonable(s.low), high=jsonable(s.high), shape=list(s.shape), dtype=str(s.dtype))
                               for n, s in env.observation_space.spaces.items()},
            action_space={n: dict(low=jsonable(s.low), high=jsonable(s.high), shape=list(s.shape), dtype=str(s.dtype))
                          for n, s in env.action_space.spaces.items()},
            cameras=[dict(name=c.name, log_name=c.log_name, w=c.w, h=c.h, c=c.c, fl=c.fl, pp=list(c.pp)) for c in env.cameras],
            info_keys=sorted(env.info.keys()), metadata=jsonable(env.metadata), render_mode=env.render_mode)
    consts = {n: jsonable(getattr(k, n)) for n in (
        "MAX_EPISODE_STEPS", "FPS", "CONTROL_TIMESTEP", "MAX_Q_VEL", "CTRL_ALPHA", "IK_RES_RAD", "IK_RES_REG_PREV",
        "IK_RES_REG_HOME", "IK_JAC_RAD", "IK_JAC_REG", "CUBE_SPAWN_RANGE", "EE_POS_DELTA", "EE_ORN_DELTA", "EPSILON",
        "Q_POS_DELTA", "EE_S_MIN", "EE_S_MAX", "EE_S_DELTA", "REWARD_SUCCESS_THRESHOLD", "REWARD_VEL_PENALTY",
        "REWARD_GRIP_DIST", "REWARD_TOUCH_CUBE", "REWARD_LIFT_CUBE", "XYZW_2_WXYZ", "MOCAP_ID_R", "MOCAP_ID_L")}
    consts["OBS_DTYPE"], consts["ACT_DTYPE"] = str(np.dtype(k.OBS_DTYPE)), str(np.dtype(k.ACT_DTYPE))
    with open(os.path.join(OUT, "ref_spaces.json"), "w") as f:
        json.dump(dict(envs=out, constants=consts), f, indent=1)       # (insertion order of the space Dicts is data: no sort_keys)
    return out


def gen_run(env_id, plan, seed):
    """plan = [steps of episode 0, steps of episode 1, ...]; every episode starts with the reference's reset()."""
    cm = compile_model(env_id)
    with quiet():
        env = refrun.make_env(env_id)
    ph = env.env.physics
    rng = np.random.default_rng(seed)
    np.random.seed(seed)                       # the cube spawn draws from the GLOBAL NumPy stream (env_sim.py:34)
    rec = {k: [] for k in ("pre_qpos", "pre_qvel", "pre_ctrl", "pre_warm", "pre_step", "action", "ctrl_set", "qpos_teleport",
                           "mocap_pos", "mocap_quat", "ik_nfev", "ik_status", "post_qpos", "post_qvel", "post_warm", "obs",
                           "reward", "terminated", "sim_time", "contact_mask", "is_success", "info_step", "info_episode")}
    resets = {k: [] for k in ("reset_qpos", "reset_qvel", "reset_ctrl", "reset_warm", "reset_obs", "reset_sim_time", "reset_at")}
    images = {}
    real_step = ph.step

    def stepped(n=1):                          # observer: what before_step left in data, before the physics runs
        rec["qpos_teleport"].append(ph.data.qpos.copy())
        rec["ctrl_set"].append(ph.data.ctrl.copy())
        real_step(n)

    ph.step = stepped
    for ep, nstep in enumerate(plan):
        with quiet():
            obs, info = env.reset(seed=seed + ep)
        resets["reset_qpos"].append(ph.data.qpos.copy()); resets["reset_qvel"].append(ph.data.qvel.copy())
        resets["reset_ctrl"].append(ph.data.ctrl.copy()); resets["reset_warm"].append(ph._warm.copy())
        resets["reset_obs"].append(flat_obs(cm, obs)); resets["reset_sim_time"].append(info["sim_time"])
        resets["reset_at"].append(len(rec["action"]))
        assert info["terminated"] is False and info["reward"] is None and info["step"] == 0
        for t in range(nstep):
            a = env.action_space.sample(rng)
            rec["pre_qpos"].append(ph.data.qpos.copy()); rec["pre_qvel"].append(ph.data.qvel.copy())
            rec["pre_ctrl"].append(ph.data.ctrl.copy()); rec["pre_warm"].append(ph._warm.copy()); rec["pre_step"].append(t)
            rec["action"].append(flat_action(cm, a))
            n0 = len(refrun.IK_LOG)
            with quiet():
                obs, reward, terminated, truncated, info = env.step(a)
            assert truncated is False
            calls = refrun.IK_LOG[n0:]
            # before_step runs the right arm's ik() first, then the left's (env_sim.py:60-99)
            arms = [arm for arm, key in ((0, "eer_pos"), (1, "eel_pos")) if key in a]
            assert len(calls) == len(arms)
            nf, st = [0, 0], [-3, -3]
            for arm, (n, s) in zip(arms, calls):
                nf[arm], st[arm] = n, s
            rec["ik_nfev"].append(nf); rec["ik_status"].append(st)
            rec["mocap_pos"].append(ph.data.mocap_pos.copy()); rec["mocap_quat"].append(ph.data.mocap_quat.copy())
            rec["post_qpos"].append(ph.data.qpos.copy()); rec["post_qvel"].append(ph.data.qvel.copy())
            rec["post_warm"].append(ph._warm.copy())
            rec["obs"].append(flat_obs(cm, obs)); rec["reward"].append(reward); rec["terminated"].append(bool(terminated))
            rec["sim_time"].append(info["sim_time"]); rec["contact_mask"].append(ph.contact_mask)
            rec["is_success"].append(bool(info["is_success"])); rec["info_step"].append(info["step"])
            rec["info_episode"].append(info["episode"])
            for cam in env.cameras:
                img = obs[cam.log_name]
                assert img.shape == (cam.h, cam.w, 3) and img.dtype == np.uint8
                if cam.w <= 64:                 # the small gripper images are kept; the 480x640 ones by shape only
                    images.setdefault("img_" + cam.name, []).append(img)
    out = {k: np.array(v) for k, v in rec.items()}
    out.update({k: np.array(v) for k, v in resets.items()})
    out.update({k: np.array(v) for k, v in images.items()})
    out["contact_mask"] = out["contact_mask"].astype(np.uint32)
    np.savez_compressed(os.path.join(OUT, "ref_run_%s.npz" % env_id), **out)
    return out


def gen_ik(env_id, n_cases=48, seed=0):
    """The cases of tests/tools/make_golden.py:gen_ik, through the reference's ik / ik_res / ik_jac."""
    k, env_base, env_sim, ikm = refrun.install()
    cm = compile_model(env_id)
    with quiet():
        env = refrun.make_env(env_id)
    ph = env.env.physics
    rng = np.random.default_rng(seed)
    rg = ph.model.jnt_range[:cm.nlink]
    hm = np.array([cm.desc.q_home[i] for i in range(cm.nlink)])
    masks = {0: env.q_id_r_mask, 1: env.q_id_l_mask}
    rec = dict(arm=[], qpos=[], action=[], goal_pos=[], goal_quat=[], q_out=[], qpos_after=[], nfev=[], status=[], res0=[], jac0=[])
    for t in range(n_cases):
        ai, site = ARMS[env_id][t % len(ARMS[env_id])]
        mask = np.asarray(masks[ai])
        qpos = np.zeros(cm.nq)
        qpos[:cm.nlink] = np.clip(hm + rng.normal(0, 0.3, cm.nlink) * (t > 1), rg[:, 0] + 1e-3, rg[:, 1] - 1e-3)
        if t % 8 == 3:
            qpos[mask[1]] = rg[mask[1], 0]
        if t % 8 == 5:
            qpos[mask[2]] = rg[mask[2], 1] + 1e-3
        qpos[cm.nlink:cm.nlink + 3] = [0.2, 0.5, 0.65]
        qpos[cm.nlink + 3] = 1
        ph.data.qpos[:] = qpos
        ph.forward()
        s = ph.data.site(site)
        a = rng.uniform(-1, 1, 6).astype(np.float32)
        # the decode of env_sim.py:62-69, statement for statement on the reference's constants
        gp = a[:3] * k.EE_POS_DELTA
        gp += s.xpos.copy()
        go = a[3:] * k.EE_ORN_DELTA
        go += env_sim.R.from_matrix(s.xmat.reshape(3, 3)).as_euler("xyz")
        gq = env_sim.R.from_euler("xyz", go).as_quat()[k.XYZW_2_WXYZ]
        r0 = ikm.ik_res(qpos[mask].copy(), physics=ph, goal_pos=gp, goal_orn=gq, q_mask=mask, q_pos_home=env.q_pos_home[mask],
                        q_pos_prev=qpos[mask], ee_site=site)
        j0 = ikm.ik_jac(qpos[mask].copy(), physics=ph, goal_orn=gq, q_mask=mask, ee_site=site)
        ph.data.qpos[:] = qpos
        ph.forward()
        n0 = len(refrun.IK_LOG)
        with quiet():
            q = ikm.ik(ph, goal_pos=gp, goal_orn=gq, ee_site=site, q_mask=mask, q_pos_home=env.q_pos_home, q_pos_prev=qpos.copy())
        nfev, status = refrun.IK_LOG[n0]
        pad = lambda v, n: np.pad(np.asarray(v, dtype=float).ravel(), (0, n - np.size(v)))
        rec["arm"].append(ai); rec["qpos"].append(qpos); rec["action"].append(a)
        rec["goal_pos"].append(gp); rec["goal_quat"].append(gq)
        rec["q_out"].append(pad(q, 7)); rec["qpos_after"].append(ph.data.qpos.copy())
        rec["nfev"].append(nfev); rec["status"].append(status)
        rec["res0"].append(pad(r0, 20)); rec["jac0"].append(pad(j0, 140))
    out = {k_: np.array(v) for k_, v in rec.items()}
    np.savez_compressed(os.path.join(OUT, "ref_ik_%s.npz" % FAMILY[env_id]), **out)
    return out


def gen_obs(env_id, n_cases=24, seed=3):
    """get_observation / get_reward at seeded states incl. out-of-range values (every clip of env_sim.py:110-139)."""
    cm = compile_model(env_id)
    with quiet():
        env = refrun.make_env(env_id)
    ph, task = env.env.physics, env.env.task
    rng = np.random.default_rng(seed)
    rg = ph.model.jnt_range[:cm.nlink]
    rec = dict(qpos=[], qvel=[], obs=[], reward=[], contact_mask=[])
    for t in range(n_cases):
        wide = 1.0 + 0.5 * (t % 3)             # t % 3 != 0: joints / cube outside their ranges, velocities beyond MAX_Q_VEL
        qpos = np.zeros(cm.nq)
        mid, half = 0.5 * (rg[:, 0] + rg[:, 1]), 0.5 * (rg[:, 1] - rg[:, 0])
        qpos[:cm.nlink] = mid + half * rng.uniform(-wide, wide, cm.nlink)
        qpos[cm.nlink:cm.nlink + 3] = np.array([0.2, 0.6, 0.65]) + np.array([0.1, 0.1, 0.05]) * rng.uniform(-wide, wide, 3) * 1.2
        quat = rng.normal(size=4)
        qpos[cm.nlink + 3:] = quat / np.linalg.norm(quat)
        qvel = rng.normal(0, 2.0 * wide, cm.nv)
        ph.data.qpos[:] = qpos
        ph.data.qvel[:] = qvel
        ph.forward()
        obs = task.get_observation(ph)
        rew = task.get_reward(ph)
        for name in ("q_pos", "q_vel", "cube_pos", "cube_orn"):
            assert obs[name].dtype == np.float64
        rec["qpos"].append(qpos); rec["qvel"].append(qvel); rec["obs"].append(flat_obs(cm, obs)); rec["reward"].append(rew)
        rec["contact_mask"].append(ph.contact_mask)
    out = {k_: np.array(v) for k_, v in rec.items()}
    out["contact_mask"] = out["contact_mask"].astype(np.uint32)
    np.savez_compressed(os.path.join(OUT, "ref_obs_%s.npz" % FAMILY[env_id]), **out)
    return out


def gen_touch(env_id, seed=5):
    """get_reward's touch / lift branch (env_sim.py:164-178) with finger geoms named as that code expects."""
    cm = compile_model(env_id, touch_reward=True)
    refrun.RIG_FINGER_GEOM_NAMES = True
    try:
        with quiet():
            env = refrun.make_env(env_id)
    finally:
        refrun.RIG_FINGER_GEOM_NAMES = False
    ph, task = env.env.physics, env.env.task
    ikm = refrun.install()[3]
    from oracle import ik_scipy as S
    rng = np.random.default_rng(seed)
    hm = np.array([cm.desc.q_home[i] for i in range(cm.nlink)])
    rec = dict(qpos=[], qvel=[], reward=[], contact_mask=[], case=[])
    nfinger = 2 * (cm.nlink // 10)
    for t in range(6 * nfinger):
        f, case = t % nfinger, (t // nfinger) % 3
        qpos = np.zeros(cm.nq)
        qpos[:cm.nlink] = hm + rng.normal(0, 0.05, cm.nlink)
        xp, xq, _ = ph.arm.fk(qpos)
        sp = cm.asset["spheres"][f]
        c = xp[sp["link"]] + S.quat2mat(xq[sp["link"]]) @ np.array(sp["pos"])       # finger sphere centre
        if case == 2:      # finger on a cube that rests on the table -> touch, no lift: the reference's ik() lowers the hand first
            arm = f // 2
            site, mask = ARMS[env_id][arm][1], np.asarray([env.q_id_r_mask, env.q_id_l_mask][arm])
            for _ in range(4):
                ph.data.qpos[:] = qpos
                ph.forward()
                goal = ph.data.site(site).xpos.copy()
                goal[2] = cm.desc.table_z + 0.035
                gq = np.empty(4)
                refrun._mju_mat2Quat(gq, ph.data.site(site).xmat)
                with quiet():
                    qpos[mask] = ikm.ik(ph, goal_pos=goal, goal_orn=gq, ee_site=site, q_mask=mask, q_pos_home=env.q_pos_home,
                                        q_pos_prev=qpos.copy())
            xp, xq, _ = ph.arm.fk(qpos)
            c = xp[sp["link"]] + S.quat2mat(xq[sp["link"]]) @ np.array(sp["pos"])
            cube = np.array([c[0], c[1], cm.desc.table_z + 0.0195])                 # resting on the table under the finger
        elif case == 1:    # no finger contact: the cube far from the hand, resting on the table
            cube = np.array([-0.3, 0.6, cm.desc.table_z + 0.0195])
        else:              # finger on the cube, cube in the air -> touch + lift
            cube = c + np.array([0.0, 0.0, -0.025])
        qpos[cm.nlink:cm.nlink + 3] = cube
        qpos[cm.nlink + 3] = 1.0
        qvel = rng.normal(0, 0.1, cm.nv)
        ph.data.qpos[:] = qpos
        ph.data.qvel[:] = qvel
        ph.forward()
        rec["qpos"].append(qpos); rec["qvel"].append(qvel); rec["reward"].append(task.get_reward(ph))
        rec["contact_mask"].append(ph.contact_mask); rec["case"].append(case)
    out = {k_: np.array(v) for k_, v in rec.items()}
    out["contact_mask"] = out["contact_mask"].astype(np.uint32)
    np.savez_compressed(os.path.join(OUT, "ref_touch_%s.npz" % FAMILY[env_id]), **out)
    return out


def gen_h5_scripted(env_id, nstep=64, seed=7):
    """One episode of the reference's KManipEnv(log_h5py=True) driven by the heuristic loop of examples/2_synthetic_data.py:28-41."""
    import tempfile
    import h5_recorder as H
    k = refrun.install()[0]
    cm = compile_model(env_id)
    scratch = tempfile.mkdtemp(prefix="kmanip_ref_h5_")
    H.FILES.clear()
    with quiet():
        env = refrun.make_env(env_id, log_dir_root=scratch, log_h5py=True, log_prefix="sim_synth")
    ph = env.env.physics
    rng = np.random.default_rng(seed)
    np.random.seed(seed)
    rec = {n: [] for n in ("pre_qpos", "pre_qvel", "pre_ctrl", "pre_warm", "action_sampled", "raw_action", "action", "post_qpos",
                           "post_qvel", "obs", "reward", "is_success", "info_step", "sim_time", "terminated")}
    with quiet():
        obs, info = env.reset(seed=seed)
    rec_reset = dict(reset_qpos=ph.data.qpos.copy(), reset_qvel=ph.data.qvel.copy(), reset_ctrl=ph.data.ctrl.copy(),
                     reset_warm=ph._warm.copy(), reset_obs=flat_obs(cm, obs))
    for t in range(nstep):
        action = env.action_space.sample(rng)
        rec["action_sampled"].append(flat_action(cm, action))
        # examples/2_synthetic_data.py:32-37, on the reference env object (env.unwrapped is the env itself without gym.make's wrappers)
        cube_pos = env.env.physics.data.qpos[-7:-4].copy()
        eer_pos = env.env.physics.data.site("eer_site_pos").xpos.copy()
        raw_action = cube_pos - eer_pos
        raw_action /= np.linalg.norm(raw_action)
        action["eer_pos"] = raw_action
        rec["raw_action"].append(raw_action.copy()); rec["action"].append(flat_action(cm, action))
        rec["pre_qpos"].append(ph.data.qpos.copy()); rec["pre_qvel"].append(ph.data.qvel.copy())
        rec["pre_ctrl"].append(ph.data.ctrl.copy()); rec["pre_warm"].append(ph._warm.copy())
        with quiet():
            obs, reward, terminated, truncated, info = env.step(action)
        rec["post_qpos"].append(ph.data.qpos.copy()); rec["post_qvel"].append(ph.data.qvel.copy())
        rec["obs"].append(flat_obs(cm, obs)); rec["reward"].append(reward); rec["is_success"].append(bool(info["is_success"]))
        rec["info_step"].append(info["step"]); rec["sim_time"].append(info["sim_time"]); rec["terminated"].append(bool(terminated))
    with quiet():
        env.close()
    (path, f), = H.FILES.items()
    assert f.closed and path.startswith(scratch)
    tree = dict(file=os.path.basename(path), log_dir_prefix=os.path.basename(os.path.dirname(path)).split(".")[0],
                rdcc_nbytes=f.rdcc_nbytes, flushes=f.flushes, steps=nstep, tree=H.tree(f, skip_attr_values=("cpu_time",)))
    with open(os.path.join(OUT, "ref_h5_tree_%s.json" % env_id), "w") as fp:
        json.dump(tree, fp, indent=1)
    out = {n: np.array(v) for n, v in rec.items()}
    out.update(rec_reset)
    for dpath, arr in H.datasets(f).items():
        if "images" not in dpath or arr.shape[2] <= 64:          # the 480 x 640 frames by shape only (the tree has it)
            out["h5/" + dpath] = arr
    # ---- info["is_success"] = reward > REWARD_SUCCESS_THRESHOLD (env_base.py:250): the grip-distance term is 0.01 / (dist + 1e-6),
    # so the cube's centre has to sit within 5 mm of the gripper site.  Rigged state: the cube AT the right gripper site, one step.
    if True:
        with quiet():
            env2 = refrun.make_env(env_id)
            env2.reset(seed=seed)
        ph2 = env2.env.physics
        site = ph2.data.site("eer_site_pos").xpos.copy()
        best = None
        for dz in (0.0, 0.002, 0.004, -0.002, 0.006, 0.008, 0.01):     # the cube falls ~2 mm per control step when nothing holds it
            ph2.data.qpos[cm.nlink:cm.nlink + 3] = site + np.array([0.0, 0.0, dz])
            ph2.data.qvel[:] = 0
            ph2.forward()
            pre = (ph2.data.qpos.copy(), ph2.data.qvel.copy(), ph2.data.ctrl.copy(), ph2._warm.copy())
            a0 = collections.OrderedDict((n, np.zeros(sp.shape, dtype=np.float32)) for n, sp in env2.action_space.spaces.items())
            with quiet():
                o2, r2, _, _, i2 = env2.step(a0)
            if r2 > k.REWARD_SUCCESS_THRESHOLD:
                best = (pre, flat_action(cm, a0), r2, bool(i2["is_success"]), flat_obs(cm, o2))
                break
            with quiet():
                env2.reset(seed=seed)
        assert best is not None, "no rigged state reached reward > REWARD_SUCCESS_THRESHOLD"
        (q, v, c, w), a, r, ok, o = best
        assert ok is True
        out.update(success_qpos=q, success_qvel=v, success_ctrl=c, success_warm=w, success_action=a, success_reward=r,
                   success_is_success=ok, success_obs=o, success_threshold=k.REWARD_SUCCESS_THRESHOLD)
    np.savez_compressed(os.path.join(OUT, "ref_scripted_%s.npz" % env_id), **out)
    import shutil
    shutil.rmtree(scratch, ignore_errors=True)
    return tree, out


PLANS = {"KManipSoloArm": [64, 16], "KManipSoloArmQPos": [40], "KManipSoloArmVision": [3],
         "KManipDualArm": [64, 8], "KManipDualArmQPos": [40], "KManipDualArmVision": [2],
         "KManipTorso": [64, 8], "KManipTorsoVision": [2]}

if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    sp = gen_spaces()
    print("ref_spaces.json:", list(sp))
    for i, (env_id, plan) in enumerate(PLANS.items()):
        r = gen_run(env_id, plan, seed=100 + i)
        print("ref_run_%s: %d steps, mean nfev %s, contacts in %d steps" % (
            env_id, len(r["action"]), r["ik_nfev"].mean(0), int((r["contact_mask"] != 0).sum())))
    for env_id, nstep in (("KManipSoloArm", 64), ("KManipDualArm", 64), ("KManipTorso", 64), ("KManipSoloArmVision", 3)):
        t, r = gen_h5_scripted(env_id, nstep)
        print("ref_h5_tree_%s / ref_scripted_%s: %s, %d flushes, rewards %.3f .. %.3f, rigged success reward %.3f" % (
            env_id, env_id, t["file"], t["flushes"], r["reward"].min(), r["reward"].max(), r["success_reward"]))
    for env_id in FAMILY:
        r = gen_ik(env_id)
        print("ref_ik_%s: nfev %s status %s" % (FAMILY[env_id], r["nfev"][:12], sorted(set(r["status"].tolist()))))
        r = gen_obs(env_id)
        print("ref_obs_%s: reward %.4f .. %.4f" % (FAMILY[env_id], r["reward"].min(), r["reward"].max()))
        r = gen_touch(env_id)
        print("ref_touch_%s: rewards %s" % (FAMILY[env_id], np.round(r["reward"], 3)))
