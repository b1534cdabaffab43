#!/usr/bin/env python3
"""ONE command that pins MuJoCo's mj_step under the path -- on a box where `import mujoco` works:

    python tests/tools/make_golden_mujoco.py            # -> tests/golden/mj_{solo_arm,dual_arm,torso}.npz
    python -m pytest tests/test_mujoco_pin.py           # oracle vs MuJoCo (CPU) and, with -m gpu, the HIP engine vs MuJoCo

It cannot run on this pool (no `mujoco` in the build image or on the GPU box; the reference's own MJCF needs STL meshes that
are not in its checkout), so until someone runs it the two tests skip with that reason and DESIGN.md says "mj_step UNPINNED".

What it does, per surrogate model: writes the build's mesh-free MJCF (tools/mjcf_export.py) and loads it into MuJoCo; takes
seeded states from rollouts of the ORACLE under random joint-delta actions (states with sphere-cube / sphere-table contacts
first); and for every state records MuJoCo's own qM, qfrc_bias, qacc_smooth, qacc, efc_J / R / aref / type and contact set at
the state, and a whole control step in dm_control's legacy order (`mj_step2; 9 x mj_step; mj_step1`, Physics.step(10),
env_sim.py:196-200,210) from the ctrl the reference's before_step would set (joint-delta mode: no IK; that decode is pinned to
the reference's Python by tests/golden/ref_run_*QPos.npz).  File layout: tests/tools/mujoco_pin.py.

The plumbing (joint order checks, array shapes, the legacy step sequence, the contact-to-mask map) is unit-tested against an
oracle-backed stand-in of the `mujoco` module (tests/tools/fake_mujoco.py) -- files made that way say engine "fake" and the
pin tests refuse them."""
import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..", "..")
for p in (ROOT, HERE, os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)

import mjcf_export  # noqa: E402
import mujoco_pin as MP  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
N_SUB = 10                      # CONTROL_TIMESTEP / model timestep (gym_kmanip/__init__.py:30, dm_control n_sub_steps)


def load_model(mj, asset):
    """The exported surrogate in MuJoCo, with the layout the fixtures assume checked: joint i <-> qpos i <-> actuator i for the
    links, the cube's free joint last, dense constraint Jacobian."""
    xml = mjcf_export.export(asset)
    m = mj.MjModel.from_xml_string(xml)
    nl = len(asset["links"])
    assert (m.nq, m.nv, m.nu) == (nl + 7, nl + 6, nl), ("model sizes", m.nq, m.nv, m.nu)
    for i, l in enumerate(asset["links"]):
        assert mj.mj_id2name(m, mj.mjtObj.mjOBJ_JOINT, i) == l["joint"]["name"] and int(m.jnt_qposadr[i]) == i and int(m.jnt_dofadr[i]) == i, \
            "link %d is not joint / qpos / dof %d of the exported model" % (i, i)
    assert mj.mj_id2name(m, mj.mjtObj.mjOBJ_JOINT, nl) == "cube_joint" and int(m.jnt_qposadr[nl]) == nl
    assert abs(float(m.opt.timestep) - float(asset["option"]["timestep"])) < 1e-15
    assert not mj.mj_isSparse(m), "dense efc_J expected (nv < 60)"
    return m


def pick_states(asset_name, n_states, seed):
    """Seeded states from oracle rollouts under random joint-delta actions, contact-rich ones first.  Returns (cm, list of
    (qpos, qvel, ctrl, warm, action, ctrl_set))."""
    from gym_kmanip_amd.model import compile_model
    from oracle.oracle import Oracle
    cm = compile_model(MP.qpos_spec(asset_name), auto_reset=False)
    n_env, horizon = 48, 36
    o = Oracle(cm, n_env, seed=seed)
    o.reset()
    rng = np.random.default_rng(seed)
    pool = []
    for k in range(horizon):
        act = rng.uniform(-1, 1, (n_env, cm.act_dim)).astype(np.float32)
        before = o.get_state()
        o.step(act)
        after = o.get_state()
        mask = o.get_diag()[0]
        if k >= 6 and k % 3 == 0:               # (the cube has landed by then)
            for e in range(n_env):
                score = 2 * bool(mask[e] & 0x000FFF00) + bool(mask[e] & 0xFFF00000)      # sphere-cube, sphere-table
                pool.append((score, k, e, tuple(x[e].copy() for x in before[:4]) + (act[e].copy(), after[2][e].copy())))
    pool.sort(key=lambda t: (-t[0], t[1], t[2]))
    rich = [p for p in pool if p[0] > 0][: (2 * n_states) // 3]
    plain = [p for p in pool if p[0] == 0]
    step = max(1, len(plain) // max(1, n_states - len(rich)))
    chosen = rich + plain[::step][: n_states - len(rich)]
    return cm, [c[3] for c in chosen]


def contact_list(mj, m, d):
    out = []
    for i in range(int(d.ncon)):
        c = d.contact[i]
        if int(getattr(c, "exclude", 0)) != 0:
            continue
        g1 = mj.mj_id2name(m, mj.mjtObj.mjOBJ_GEOM, int(c.geom1))
        g2 = mj.mj_id2name(m, mj.mjtObj.mjOBJ_GEOM, int(c.geom2))
        out.append((g1, g2, np.array(c.pos, dtype=np.float64)))
    return out


def state_mask(mj, m, d, asset):
    cube = mj.mj_name2id(m, mj.mjtObj.mjOBJ_BODY, "cube")
    return MP.contacts_to_mask(contact_list(mj, m, d), np.array(d.xpos[cube]), np.array(d.xmat[cube]).reshape(3, 3),
                               [s["name"] for s in asset["spheres"]])


def put(mj, m, d, qpos, qvel, ctrl, warm):
    mj.mj_resetData(m, d)
    d.qpos[:] = qpos; d.qvel[:] = qvel; d.ctrl[:] = ctrl; d.qacc_warmstart[:] = warm


def gen(mj, asset_name, n_states=24, seed=7, engine="mujoco"):
    from gym_kmanip_amd.model import load_asset
    asset = load_asset(asset_name)
    m = load_model(mj, asset)
    cm, states = pick_states(asset_name, n_states, seed)
    nq, nv, nu = cm.nq, cm.nv, cm.nu
    S = len(states)
    d = mj.MjData(m)
    rows = []
    for (qpos, qvel, ctrl, warm, act, ctrl_set) in states:
        # ---- the state as mj_forward sees it
        put(mj, m, d, qpos, qvel, ctrl, warm)
        mj.mj_forward(m, d)
        qM = np.zeros((nv, nv)); mj.mj_fullM(m, qM, d.qM)
        ne = int(d.nefc)
        r = dict(qpos=qpos, qvel=qvel, ctrl=ctrl, warm=warm, action=act, ctrl_set=ctrl_set, qM=qM,
                 qfrc_bias=np.array(d.qfrc_bias), qacc_smooth=np.array(d.qacc_smooth), qacc=np.array(d.qacc), nefc=ne,
                 efc_J=np.asarray(d.efc_J, dtype=np.float64).reshape(-1)[: ne * nv].reshape(ne, nv).copy(),
                 efc_R=np.array(d.efc_R[:ne]), efc_aref=np.array(d.efc_aref[:ne]), efc_type=np.array(d.efc_type[:ne], dtype=np.int32),
                 mask=state_mask(mj, m, d, asset))
        # ---- one control step, dm_control's legacy order: the first mj_step2 consumes the mj_step1 products of the state the
        # previous step left (here: of this state); ctrl is set in between (before_step -> set_control)
        put(mj, m, d, qpos, qvel, ctrl, warm)
        mj.mj_step1(m, d)
        d.ctrl[:] = ctrl_set
        mj.mj_step2(m, d)
        for _ in range(N_SUB - 1):
            mj.mj_step(m, d)
        mj.mj_step1(m, d)
        r.update(post_qpos=np.array(d.qpos), post_qvel=np.array(d.qvel), post_warm=np.array(d.qacc_warmstart),
                 post_mask=state_mask(mj, m, d, asset))
        rows.append(r)
    E = max([r["nefc"] for r in rows] + [1])

    def padded(key, width):
        a = np.zeros((S, E) + ((width,) if width else ()), dtype=rows[0][key].dtype)
        for i, r in enumerate(rows):
            a[i, : r["nefc"]] = r[key]
        return a
    out = {k: np.array([r[k] for r in rows]) for k in ("qpos", "qvel", "ctrl", "warm", "action", "ctrl_set", "qM", "qfrc_bias", "qacc_smooth",
                                                       "qacc", "post_qpos", "post_qvel", "post_warm")}
    out.update(nefc=np.array([r["nefc"] for r in rows], dtype=np.int32), efc_J=padded("efc_J", nv), efc_R=padded("efc_R", 0),
               efc_aref=padded("efc_aref", 0), efc_type=padded("efc_type", 0),
               mask=np.array([r["mask"] for r in rows], dtype=np.uint32), post_mask=np.array([r["post_mask"] for r in rows], dtype=np.uint32),
               meta=MP.pack_meta(engine=engine, engine_version=str(getattr(mj, "__version__", "?")), asset=asset_name,
                                 act_list=list(cm.spec.act_list), nstate=S, seed=seed, n_sub=N_SUB))
    assert out["qpos"].shape == (S, nq) and out["ctrl_set"].shape == (S, nu)
    return out


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=OUT)
    ap.add_argument("--states", type=int, default=24)
    ap.add_argument("assets", nargs="*", default=list(MP.ASSETS))
    a = ap.parse_args(argv)
    try:
        import mujoco
    except ImportError:
        sys.stderr.write("make_golden_mujoco.py needs the `mujoco` package (absent here): run it on a box that has it; nothing was written\n")
        return 3
    for name in a.assets:
        out = gen(mujoco, name, a.states)
        path = os.path.join(a.out, "mj_%s.npz" % name)
        np.savez_compressed(path, **out)
        print(path, "states", len(out["qpos"]), "max nefc", out["efc_J"].shape[1], "mujoco", mujoco.__version__)
    return 0


if __name__ == "__main__":
    sys.exit(main())
