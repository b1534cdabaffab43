"""An oracle-backed stand-in for the part of the `mujoco` Python API tests/tools/make_golden_mujoco.py uses -- for the
PLUMBING test only (tests/test_mujoco_pin.py::test_generator_plumbing_with_a_fake_mujoco): joint-order checks, array shapes,
the legacy step sequence, the contact list -> mask map.  Every number it returns is the C oracle's (oracle/kmanip_oracle.c), so
a fixture made with it pins nothing; the generator labels it engine "fake" and the pin tests refuse such a file.
(Same idea as tests/tools/refrun.py, which stands in for the packages the reference's Python imports.)"""
import enum
import json
import os
import re
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, HERE)
import mujoco_pin as MP  # noqa: E402

__version__ = "fake-oracle"


class mjtObj(enum.IntEnum):
    mjOBJ_BODY = 1
    mjOBJ_JOINT = 3
    mjOBJ_GEOM = 5


def _quat2mat(q):
    w, x, y, z = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


class MjModel:
    @classmethod
    def from_xml_string(cls, xml):
        from gym_kmanip_amd.model import compile_model, load_asset
        from oracle.oracle import Oracle
        self = cls()
        name = re.search(r'<mujoco model="(\w+)_surrogate">', xml).group(1)
        self.asset = load_asset(name)
        self.cm = compile_model(MP.qpos_spec(name), auto_reset=False)
        self.o = Oracle(self.cm, 1)
        nl = self.cm.nlink
        self.nq, self.nv, self.nu = nl + 7, nl + 6, nl
        self.jnt_qposadr = np.arange(nl + 1); self.jnt_dofadr = np.arange(nl + 1)
        self.opt = types.SimpleNamespace(timestep=float(self.asset["option"]["timestep"]))
        self.names = {mjtObj.mjOBJ_JOINT: [l["joint"]["name"] for l in self.asset["links"]] + ["cube_joint"],
                      mjtObj.mjOBJ_BODY: ["world", "cube"], mjtObj.mjOBJ_GEOM: ["table", "cube"]}
        for s in self.asset["spheres"]:
            seg = s.get("seg")
            if seg is not None and any(float(x) != 0.0 for x in seg):
                self.names[mjtObj.mjOBJ_GEOM] += [s["name"] + "__seg", s["name"] + "__capsule"]
            else:
                self.names[mjtObj.mjOBJ_GEOM].append(s["name"] + ("__seg" if seg is not None else ""))
        return self


def mj_id2name(m, t, i):
    return m.names[t][i]


def mj_name2id(m, t, name):
    return m.names[t].index(name)


def mj_isSparse(m):
    return False


class MjData:
    def __init__(self, m):
        self.m = m
        mj_resetData(m, self)


def mj_resetData(m, d):
    d.qpos = np.zeros(m.nq); d.qvel = np.zeros(m.nv); d.ctrl = np.zeros(m.nu); d.qacc_warmstart = np.zeros(m.nv)
    d.qacc = np.zeros(m.nv); d.qacc_smooth = np.zeros(m.nv); d.qfrc_bias = np.zeros(m.nv); d.qM = None
    d.nefc = 0; d.efc_J = np.zeros(0); d.efc_R = np.zeros(0); d.efc_aref = np.zeros(0); d.efc_type = np.zeros(0, dtype=np.int32)
    d.ncon = 0; d.contact = []; d.xpos = np.zeros((2, 3)); d.xmat = np.zeros((2, 9)); d._s1 = None


def _collide(m, d):
    """data.contact / xpos / xmat at data.qpos, from the oracle's mask."""
    nl = m.cm.nlink
    mask = m.o.contact_mask(d.qpos)[0]
    cp, cR = d.qpos[nl:nl + 3], _quat2mat(d.qpos[nl + 3:nl + 7])
    d.xpos[1] = cp; d.xmat[1] = cR.reshape(-1)
    g = m.names[mjtObj.mjOBJ_GEOM]
    half = np.array(m.asset["cube"]["half_size"], dtype=np.float64)
    con = []
    for i in range(8):
        if mask >> i & 1:
            loc = half * np.array([1 if i & 1 else -1, 1 if i & 2 else -1, 1 if i & 4 else -1])
            con.append(types.SimpleNamespace(geom1=g.index("table"), geom2=g.index("cube"), pos=cp + cR @ loc, exclude=0))
    for s, sp in enumerate(m.asset["spheres"]):
        seg = sp.get("seg")
        cap = seg is not None and any(float(x) != 0.0 for x in seg)
        if mask >> (8 + s) & 1:          # the cube meets a link-capsule's segment geom, or the plain sphere
            nm = sp["name"] + ("__capsule" if cap else ("__seg" if seg is not None else ""))
            con.append(types.SimpleNamespace(geom1=g.index(nm), geom2=g.index("cube"), pos=cp.copy(), exclude=0))
        if mask >> (20 + s) & 1:         # the table meets its end sphere
            nm = sp["name"] + ("__seg" if seg is not None else "")
            con.append(types.SimpleNamespace(geom1=g.index("table"), geom2=g.index(nm), pos=cp.copy(), exclude=0))
    d.contact = con; d.ncon = len(con)


def mj_forward(m, d):
    r = m.o.dynamics(d.qpos, d.qvel, d.ctrl)
    t, _ = m.o.constraint_rows(d.qpos, d.qvel)
    nl = m.cm.nlink
    d.qM = r["M"]; d.qfrc_bias = r["bias"]; d.qacc_smooth = r["qacc_smooth"]; d.qacc = r["qacc"]; d.nefc = r["nefc"]
    d.efc_J = r["J"].reshape(-1).copy(); d.efc_R = r["R"]; d.efc_aref = r["aref"]
    single = [(np.count_nonzero(row) == 1 and np.flatnonzero(row)[0] < nl) for row in r["J"]]
    d.efc_type = np.array([MP.MJ_CNSTR_FRICTION_DOF if t[i] == 0 else (MP.MJ_CNSTR_LIMIT_JOINT if single[i] else MP.MJ_CNSTR_CONTACT_PYRAMIDAL)
                           for i in range(r["nefc"])], dtype=np.int32)
    _collide(m, d)


def mj_fullM(m, dst, qM):
    dst[:] = qM


def mj_step1(m, d):
    d._s1 = d.qpos.copy()       # the state whose products the next mj_step2 consumes
    _collide(m, d)


def mj_step2(m, d):
    q, v, w, bad, _, _, _ = m.o.physics_step(d.qpos, d.qvel, d.ctrl, d.qacc_warmstart, d._s1, 1)
    assert not bad
    d.qpos, d.qvel, d.qacc_warmstart = q, v, w


def mj_step(m, d):
    mj_step1(m, d)
    mj_step2(m, d)
