#!/usr/bin/env python3
"""Extract a build-owned, mesh-free model spec from the reference MJCF (read as DATA).

Runs only in the build container (needs /root/reference).  The JSON it writes under
gym_kmanip_amd/assets/ is what ships: kinematic tree, joint ranges, actuators, sites, cube,
cameras (numbers from the reference XML, cited below) PLUS the build's documented surrogates
for what the reference checkout lacks (link inertials and collision shapes live in STL meshes
that are git-ignored upstream: SURVEY.md finding 1).

Reference data read (file:line are the places the numbers come from):
  gym_kmanip/assets/_env_solo_arm.xml:1-18, _env_dual_arm.xml:1-27, _env_torso.xml:1-21
  gym_kmanip/assets/arm_r_body.xml:1-76, arm_l_body.xml:1-76, torso_body.xml:1-182
  gym_kmanip/assets/arm_r.xml:45-56, arm_l.xml:45-56, torso.xml:112-135 (actuators)
  gym_kmanip/assets/scene.xml:14-21 (table, cube)

Fixed (joint-less) bodies are folded into their moving parent, so every "link" in the output
has exactly one 1-DoF joint and link index == dof index == qpos index (the cube's free joint
comes last, as in the reference where scene.xml is included last).
"""
import json
import math
import os
import sys
import xml.etree.ElementTree as ET

import numpy as np

REF_ASSETS = "/root/reference/gym_kmanip/assets"
OUT_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gym_kmanip_amd", "assets")


# ---------------------------------------------------------------- quaternion helpers (wxyz)
def qmul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array([
        aw * bw - ax * bx - ay * by - az * bz,
        aw * bx + ax * bw + ay * bz - az * by,
        aw * by - ax * bz + ay * bw + az * bx,
        aw * bz + ax * by - ay * bx + az * bw,
    ])


def qrot(q, v):
    w, x, y, z = q
    R = np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
        [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
        [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)],
    ])
    return R @ np.asarray(v, dtype=float)


def qconj(q):
    return np.array([q[0], -q[1], -q[2], -q[3]])


def euler_xyz_intrinsic(e):
    """MuJoCo default eulerseq="xyz" (intrinsic): R = Rx(a) Ry(b) Rz(c)."""
    a, b, c = e
    qx = np.array([math.cos(a / 2), math.sin(a / 2), 0, 0])
    qy = np.array([math.cos(b / 2), 0, math.sin(b / 2), 0])
    qz = np.array([math.cos(c / 2), 0, 0, math.sin(c / 2)])
    return qmul(qmul(qx, qy), qz)


def fvec(s, n=None):
    v = [float(t) for t in s.split()]
    if n is not None:
        assert len(v) == n, (s, n)
    return np.array(v)


def body_frame(el):
    pos = fvec(el.get("pos", "0 0 0"), 3)
    if el.get("quat") is not None:
        q = fvec(el.get("quat"), 4)
        q = q / np.linalg.norm(q)
    elif el.get("euler") is not None:
        q = euler_xyz_intrinsic(fvec(el.get("euler"), 3))
    else:
        q = np.array([1.0, 0, 0, 0])
    return pos, q


# ---------------------------------------------------------------- include expansion
def load_expanded(path):
    root = ET.parse(path).getroot()
    _expand(root, os.path.dirname(path))
    return root


def _expand(el, base):
    i = 0
    children = list(el)
    for ch in children:
        if ch.tag == "include":
            inc = ET.parse(os.path.join(base, ch.get("file"))).getroot()
            _expand(inc, base)
            idx = list(el).index(ch)
            el.remove(ch)
            for k, sub in enumerate(list(inc)):
                el.insert(idx + k, sub)
        else:
            _expand(ch, base)
        i += 1


# ---------------------------------------------------------------- surrogate rules (build-owned)
# The reference's link masses/inertias come from absent STL meshes.  Surrogate: mass by servo
# class encoded in the joint name (MyActuator-class x8/x6/x4), a thin solid cylinder (r = 3 cm)
# spanning from the joint origin to the farthest child attachment, isotropised so that the
# tensor is frame independent.  These numbers are the build's, not the reference's.
MASS_BY_CLASS = [("slider", 0.05), ("x8", 1.0), ("x6", 0.7), ("x4", 0.4)]
LINK_RADIUS = 0.03
MIN_LINK_LEN = 0.04
# The reference XML has no armature/damping and kp = 1000 position servos integrated with explicit
# Euler at dt = 2 ms: that is only stable if every actuated direction has inertia > kp*dt^2/4 = 1e-3.
# The thin-cylinder tensor alone is below that for the wrist links, so each hinge link gets an extra
# isotropic term standing in for the (unknown) mesh bulk + reflected rotor inertia of a geared servo.
HINGE_EXTRA_INERTIA = 0.01
FINGER_RADIUS = 0.010       # one sphere collider per gripper finger
FINGER_OPEN_OFFSET = 0.035  # finger centre sits this far (along +slide axis) from the EE site at q=0
PALM_RADIUS = 0.030         # link colliders (see build()): the palm sphere; the joint-housing spheres use LINK_RADIUS -- the
                            # reference's link meshes all collide with table and cube (contype = conaffinity = 1)
TABLE_TOP_Z = 0.5           # table body origin z (scene.xml:14); surrogate = plane z = 0.5
# The table top's extent: the reference's own stand-in for tabletop.stl (examples/4_teleop.py:82-84, a 0.4 x 0.8 plane primitive at the
# table body's position).  The long side lies along x: the cube spawns at x in [0.1, 0.3], y in [0.5, 0.7] (__init__.py:164-170) and
# the table body sits at x = 0, y = 0.6, so only 0.8 along x puts every spawn on the table.
TABLE_SIZE_XY = (0.8, 0.4)


def link_mass(joint_name):
    for key, m in MASS_BY_CLASS:
        if key in joint_name:
            return m
    raise ValueError(joint_name)


def build(env_xml, name, assets_dir=None):
    """env_xml under assets_dir (default: the reference's assets).  The reference's files carry meshes only, so inertials and
    colliders come from the surrogate rules below; a PRIMITIVE-ONLY file -- what tools/mjcf_export.py writes from the JSON this
    function produced -- states them itself (<inertial> per link, sphere / capsule geoms with user="<order>", a box table, an
    <option>), and then they are READ instead: export -> build is the identity on the model (tests/test_mjcf_export.py)."""
    root = load_expanded(os.path.join(assets_dir or REF_ASSETS, env_xml))
    wb_list = root.findall("worldbody")
    links = []
    sites = {}
    cameras = []
    bodies_by_name = {}
    cube = None
    table = None
    stated_spheres, stated_segs = {}, {}

    def walk(el, parent_link, rel_pos, rel_quat):
        """el: <body>; (rel_pos, rel_quat): pose of el's PARENT frame in parent_link's frame."""
        nonlocal cube, table
        pos, quat = body_frame(el)
        p = rel_pos + qrot(rel_quat, pos)
        q = qmul(rel_quat, quat)
        joints = el.findall("joint")
        bname = el.get("name")
        if bname == "cube":
            inert = el.find("inertial")
            g = el.find("geom")
            j = joints[0]
            cube = {
                "pos0": p.tolist(), "quat0": q.tolist(),
                "mass": float(inert.get("mass")),
                "diaginertia": fvec(inert.get("diaginertia"), 3).tolist(),
                "half_size": fvec(g.get("size"), 3).tolist(),
                "frictionloss": float(j.get("frictionloss", "0")),
                "condim": int(g.get("condim", "3")),
                "friction": fvec(g.get("friction"), 3).tolist(),
                "solref": fvec(g.get("solref"), 2).tolist(),
                "solimp": fvec(g.get("solimp")).tolist(),
            }
            return
        if bname == "table":
            table = {"pos": p.tolist()}
            g = el.find("geom")
            if g is not None and g.get("type") == "box":          # primitive-only file: the top face is the table rectangle
                hs, gp = fvec(g.get("size"), 3), fvec(g.get("pos", "0 0 0"), 3)
                table["rect"] = [p[0] + gp[0] - hs[0], p[0] + gp[0] + hs[0], p[1] + gp[1] - hs[1], p[1] + gp[1] + hs[1]]
                table["plane_z"] = p[2] + gp[2] + hs[2]
            return
        if len(joints) == 0:
            # fixed body: fold into parent link
            me_link, me_p, me_q = parent_link, p, q
        else:
            assert len(joints) == 1
            j = joints[0]
            assert fvec(j.get("pos", "0 0 0"), 3).tolist() == [0, 0, 0]
            idx = len(links)
            links.append({
                "name": bname,
                "parent": parent_link,
                "pos": p.tolist(), "quat": q.tolist(),
                "joint": {
                    "name": j.get("name"),
                    "type": j.get("type", "hinge"),
                    "axis": fvec(j.get("axis", "0 0 1"), 3).tolist(),
                    "range": fvec(j.get("range"), 2).tolist(),
                    "limited": j.get("limited", "false") == "true",
                    "frictionloss": float(j.get("frictionloss", "0")),
                },
                "_attach": [],
            })
            if parent_link >= 0:
                links[parent_link]["_attach"].append(p.tolist())
            me_link, me_p, me_q = idx, np.zeros(3), np.array([1.0, 0, 0, 0])
            inert = el.find("inertial")
            if inert is not None:                                  # primitive-only file: stated, not derived
                links[idx]["inertial"] = {"mass": float(inert.get("mass")), "com": fvec(inert.get("pos", "0 0 0"), 3).tolist(),
                                          "diaginertia": fvec(inert.get("diaginertia"), 3).tolist()}
            for g in el.findall("geom"):
                if g.get("type") == "sphere" and g.get("user") is not None:
                    nm = g.get("name")
                    rec = {"name": nm[:-5] if nm.endswith("__seg") else nm, "link": idx, "pos": fvec(g.get("pos", "0 0 0"), 3).tolist(),
                           "radius": float(g.get("size")), "visible": int(float(g.get("rgba", "0 0 0 1").split()[3]) > 0)}
                    if nm.endswith("__seg"):
                        rec["seg"] = [0.0, 0.0, 0.0]
                    stated_spheres[int(float(g.get("user")))] = rec
                elif g.get("type") == "capsule" and g.get("user") is not None:
                    ft = fvec(g.get("fromto"), 6)
                    stated_segs[int(float(g.get("user")))] = (ft[3:] - ft[:3]).tolist()
        bodies_by_name[bname] = (me_link, me_p.copy(), me_q.copy())
        for s in el.findall("site"):
            sp, sq = body_frame(s)
            sites[s.get("name")] = {
                "link": me_link,
                "pos": (me_p + qrot(me_q, sp)).tolist(),
                "quat": qmul(me_q, sq).tolist(),
            }
            if me_link >= 0 and len(joints) == 0:
                links[me_link]["_attach"].append((me_p + qrot(me_q, sp)).tolist())
        for c in el.findall("camera"):
            cp, cq = body_frame(c)
            cameras.append({
                "name": c.get("name"), "link": me_link,
                "pos": (me_p + qrot(me_q, cp)).tolist(),
                "fovy": float(c.get("fovy", "45")),
                "mode": c.get("mode", "fixed"), "target": c.get("target"),
            })
        for ch in el.findall("body"):
            walk(ch, me_link, me_p, me_q)

    for wb in wb_list:
        for c in wb.findall("camera"):
            cp, _ = body_frame(c)
            cameras.append({"name": c.get("name"), "link": -1, "pos": cp.tolist(),
                            "fovy": float(c.get("fovy", "45")), "mode": c.get("mode", "fixed"),
                            "target": c.get("target")})
        for b in wb.findall("body"):
            walk(b, -1, np.zeros(3), np.array([1.0, 0, 0, 0]))

    # actuators: actuator i must drive joint i (the reference indexes qpos with ctrl ids,
    # env_sim.py:45,55)
    acts = root.find("actuator").findall("position") if root.find("actuator") is not None else []
    all_acts = []
    for a_el in root.findall("actuator"):
        all_acts += a_el.findall("position")
    assert len(all_acts) == len(links), (len(all_acts), len(links))
    for i, (a, l) in enumerate(zip(all_acts, links)):
        assert a.get("joint") == l["joint"]["name"], (i, a.get("joint"), l["joint"]["name"])
        l["actuator"] = {
            "kp": float(a.get("kp")),
            "ctrlrange": fvec(a.get("ctrlrange"), 2).tolist(),
            "forcerange": fvec(a.get("forcerange"), 2).tolist() if a.get("forcelimited") == "true" else None,
        }

    # target bodies of cameras / sites folded to (link, pos)
    targets = {}
    for k, (lk, p, q) in bodies_by_name.items():
        targets[k] = {"link": lk, "pos": p.tolist(), "quat": q.tolist()}
    targets["table"] = {"link": -1, "pos": table["pos"], "quat": [1, 0, 0, 0]}

    # ---- surrogate inertials
    for l in links:
        att = l.pop("_attach")
        if "inertial" in l:
            continue
        far = np.zeros(3)
        for a_ in att:
            if np.linalg.norm(a_) > np.linalg.norm(far):
                far = np.array(a_)
        length = max(np.linalg.norm(far), MIN_LINK_LEN)
        if np.linalg.norm(far) < 1e-9:
            com = np.array([0.0, 0.0, -0.5 * MIN_LINK_LEN])
        else:
            com = 0.5 * far
        m = link_mass(l["joint"]["name"])
        inertia = m * (3 * LINK_RADIUS ** 2 + length ** 2) / 12.0
        if l["joint"]["type"] != "slide":
            inertia += HINGE_EXTRA_INERTIA
        l["inertial"] = {"mass": m, "com": com.tolist(), "diaginertia": [inertia] * 3}

    # ---- surrogate finger colliders: one sphere per slider link, placed relative to the EE site
    spheres = []
    if stated_spheres:                                             # primitive-only file: the colliders are stated, in `user` order
        for k in sorted(stated_spheres):
            rec = stated_spheres[k]
            if k in stated_segs:
                rec["seg"] = stated_segs[k]
            spheres.append(rec)
    for i, l in enumerate(links):
        if stated_spheres or l["joint"]["type"] != "slide":
            continue
        par = l["parent"]
        # EE site sharing this hand: the site whose link is the slider's parent or a sibling
        # subtree of that parent (torso: site hangs off the x4_2 sibling link).
        best = None
        for sname, s in sites.items():
            if not sname.endswith("_site_pos"):
                continue
            lk = s["link"]
            sp = np.array(s["pos"])
            # express site in slider-parent frame at zero configuration
            while lk != par and lk >= 0:
                sp = np.array(links[lk]["pos"]) + qrot(np.array(links[lk]["quat"]), sp)
                lk = links[lk]["parent"]
            if lk == par:
                best = (sname, sp)
        assert best is not None, l["name"]
        sname, sp = best
        rel = qrot(qconj(np.array(l["quat"])), sp - np.array(l["pos"]))  # site in slider frame @ q=0
        centre = np.array([rel[0], rel[1], rel[2] + FINGER_OPEN_OFFSET])
        spheres.append({"name": "finger_" + l["joint"]["name"].split("_hand_")[-1] + ("_r" if "right" in l["joint"]["name"] else "_l"),
                        "link": i, "pos": centre.tolist(), "radius": FINGER_RADIUS,
                        "site": sname})

    # ---- surrogate link colliders, after the fingers (sphere order = priority for the solver's contact slots, so the
    # distal ones come first): per hand link H (the parent of a pair of finger sliders) the palm, half way between H's
    # origin and the EE site, then the joint housings at the origins of H and of its two ancestors (wrist, forearm,
    # elbow).  Links nearer the shoulder stay collider-free: the arms are mounted at (solo / dual) or next to (torso) the
    # table top's height and those links could only meet the table at its edge.  Collision only (visible = 0):
    # the wrist cameras sit inside / behind these spheres, where the reference's camera sees past its gripper mesh.
    for s in spheres:
        s.setdefault("visible", 1)
    hands = []
    for s in ([] if stated_spheres else list(spheres)):
        par = links[s["link"]]["parent"]
        if par in [h[0] for h in hands]:
            continue
        st = sites[s["site"]]
        lk, sp = st["link"], np.array(st["pos"])
        while lk != par and lk >= 0:                      # EE site in the hand link's frame at zero configuration
            sp = np.array(links[lk]["pos"]) + qrot(np.array(links[lk]["quat"]), sp)
            lk = links[lk]["parent"]
        assert lk == par
        hands.append((par, "r" if s["name"].endswith("_r") else "l", sp, s["site"]))
    for par, side, sp, site in hands:
        spheres.append({"name": "palm_" + side, "link": par, "pos": (0.5 * sp).tolist(), "radius": PALM_RADIUS,
                        "site": site, "visible": 0})
    # The forearm and elbow housings are the ends of CAPSULES: "seg" is the vector, in the link's own frame, from the link's
    # origin to the origin of its child on the way to the hand (= that child's body position).  Against the table a capsule
    # touches in its two end spheres (the housings); against the cube the collider is the closest point of the segment (a
    # sphere sliding along the link), so the cylinder section between two housings is covered too.  wrist: a plain sphere.
    for tier, tname in enumerate(["wrist", "forearm", "elbow"]):
        for par, side, sp, site in hands:
            lk, child = par, None
            for _ in range(tier):
                child = lk
                lk = links[lk]["parent"]
                assert lk >= 0
            seg = [0.0, 0.0, 0.0] if child is None else list(links[child]["pos"])
            spheres.append({"name": tname + "_" + side, "link": lk, "pos": [0.0, 0.0, 0.0], "radius": LINK_RADIUS,
                            "site": site, "visible": 0, "seg": seg})

    spec = {
        "name": name,
        "source": env_xml,
        "nlink": len(links),
        "links": links,
        "sites": sites,
        "targets": targets,
        "cameras": cameras,
        "cube": cube,
        "table": {"pos": table["pos"], "plane_z": table.get("plane_z", TABLE_TOP_Z),
                  "rect": table.get("rect", [table["pos"][0] - 0.5 * TABLE_SIZE_XY[0], table["pos"][0] + 0.5 * TABLE_SIZE_XY[0],
                                             table["pos"][1] - 0.5 * TABLE_SIZE_XY[1], table["pos"][1] + 0.5 * TABLE_SIZE_XY[1]])},
        "spheres": spheres,
        "option": ({"timestep": float(root.find("option").get("timestep")), "gravity": fvec(root.find("option").get("gravity"), 3).tolist()}
                   if root.find("option") is not None and root.find("option").get("timestep") else {"timestep": 0.002, "gravity": [0, 0, -9.81]}),
        "surrogate_note": "link inertials, finger / link spheres and the table plane are build-owned "
                          "surrogates (reference meshes absent); see tools/mjcf_extract.py",
    }
    return spec


def main():
    os.makedirs(OUT_DIR, exist_ok=True)
    for xml, name in [("_env_solo_arm.xml", "solo_arm"), ("_env_dual_arm.xml", "dual_arm"),
                      ("_env_torso.xml", "torso")]:
        spec = build(xml, name)
        out = os.path.join(OUT_DIR, name + ".json")
        with open(out, "w") as f:
            json.dump(spec, f, indent=1)
        print(name, "links", spec["nlink"], "sites", list(spec["sites"]), "spheres",
              [(s["name"], s["link"], np.round(s["pos"], 4).tolist()) for s in spec["spheres"]])


if __name__ == "__main__":
    sys.exit(main())
