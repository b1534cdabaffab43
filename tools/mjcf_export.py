#!/usr/bin/env python3
"""Write the build's surrogate model (gym_kmanip_amd/assets/<name>.json) as primitive-only MJCF text.

Why: MuJoCo's mj_step under the path is the one part no fixture pins -- `mujoco` is absent from the
build image and from the GPU box, and the reference's own MJCF needs 40-107 STL meshes that are not in its checkout
(SURVEY 8c).  This exporter hands a maintainer WITH MuJoCo a mesh-free model of exactly what the HIP engine and the oracle
simulate -- same kinematic tree, joint ranges, position actuators, surrogate inertials, collider spheres / capsules, table
rectangle, cube with the reference's contact parameters, sites, mocap bodies and cameras -- so that

    python tools/mjcf_export.py --out /tmp/km && python -c "import mujoco; m = mujoco.MjModel.from_xml_path('/tmp/km/solo_arm.xml')"

is a one-command start for pinning them (bench.py's cpu_baseline probes `mujoco` at run time and, when it is there, times it
on this file).  Nothing here can execute MuJoCo on this pool; what IS tested (tests/test_mjcf_export.py) is that the text
round-trips through tools/mjcf_extract.py -- the reader that made the JSON from the reference's MJCF -- to the same model.

Conventions of the emitted file (all plain MJCF):
  * one <body> per link, nested by parent, in link order (link i <-> joint i <-> qpos i <-> actuator i; the cube's free joint
    last, as in the reference where scene.xml is included last); fixed reference bodies (robot_root, arm_r, eer_site, ...)
    are emitted as jointless child bodies at their folded pose so that `target=` / mocap ids keep their names;
  * <inertial> on every link = the surrogate inertials (tools/mjcf_extract.py: mass by servo class, thin cylinder + 0.01);
  * colliders: geom `user="<index>"` keeps the sphere order (= the solver's slot priority).  Contact bits reproduce the
    surrogate's pair set -- cube x table, sphere x cube, sphere x table, nothing robot x robot (DESIGN.md 4 dev. 1):
        table   contype 0 conaffinity 3      cube     contype 1 conaffinity 4
        sphere  contype 6 conaffinity 0      (fingers, palm, wrist: meet table and cube)
        capsule end sphere  contype 2 conaffinity 0   (forearm / elbow housings: meet the TABLE as spheres)
        capsule contype 4 conaffinity 0      (the same housings' link segment: meets the CUBE only)
    (MuJoCo's capsule-box routine returns up to two contacts where the surrogate keeps one: the stated deviation);
  * the table top is a box whose upper face is the surrogate's rectangle at plane_z (thickness 2 cm, downwards).
"""
import argparse
import json
import os
import sys

ASSET_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gym_kmanip_amd", "assets")
TABLE_THICKNESS = 0.02
BITS = {"table": (0, 3), "cube": (1, 4), "sphere": (6, 0), "seg_sphere": (2, 0), "capsule": (4, 0)}


def _f(v):
    return " ".join(repr(float(x)) for x in v)


def export(spec):
    links, sites, targets, cams, spheres = spec["links"], spec["sites"], spec["targets"], spec["cameras"], spec["spheres"]
    link_names = {l["name"] for l in links}
    out = ['<mujoco model="%s_surrogate">' % spec["name"],
           '  <!-- written by tools/mjcf_export.py from gym_kmanip_amd/assets/%s.json (source: %s); build-owned surrogates, see that file -->' % (spec["name"], spec["source"]),
           '  <compiler angle="radian" autolimits="false" inertiafromgeom="false"/>',
           '  <option timestep="%s" gravity="%s"/>' % (repr(float(spec["option"]["timestep"])), _f(spec["option"]["gravity"])),
           '  <size nuser_geom="1"/>',
           '  <worldbody>']

    def emit_frame_children(link, ind):
        pad = "  " * ind
        # fixed bodies folded into this link (targets), each with the sites that sit at its origin
        used_sites = set()
        for tname, t in targets.items():
            if t["link"] != link or tname in link_names or tname in ("table", "cube"):
                continue
            mocap = ' mocap="true"' if link == -1 and any(s["link"] == -1 and s["pos"] == t["pos"] for s in sites.values()) and tname.startswith("hand_") else ""
            out.append('%s<body name="%s" pos="%s" quat="%s"%s>' % (pad, tname, _f(t["pos"]), _f(t["quat"]), mocap))
            for sname, s in sites.items():
                if sname not in used_sites and s["link"] == link and s["pos"] == t["pos"] and _same_frame(s, t, sname, tname):
                    out.append('%s  <site name="%s" pos="0 0 0" quat="%s" size="0.01"/>' % (pad, sname, _f(_rel_quat(t["quat"], s["quat"]))))
                    used_sites.add(sname)
            out.append('%s</body>' % pad)
        for sname, s in sites.items():
            if s["link"] == link and sname not in used_sites:
                out.append('%s<site name="%s" pos="%s" quat="%s" size="0.01"/>' % (pad, sname, _f(s["pos"]), _f(s["quat"])))
        for c in cams:
            if c["link"] == link:
                tgt = ' target="%s"' % c["target"] if c.get("target") else ""
                out.append('%s<camera name="%s" pos="%s" fovy="%s" mode="%s"%s/>' % (pad, c["name"], _f(c["pos"]), repr(float(c["fovy"])), c["mode"], tgt))

    def emit_link(i, ind):
        l = links[i]
        pad = "  " * ind
        j, inr = l["joint"], l["inertial"]
        out.append('%s<body name="%s" pos="%s" quat="%s">' % (pad, l["name"], _f(l["pos"]), _f(l["quat"])))
        out.append('%s  <joint name="%s" type="%s" axis="%s" pos="0 0 0" range="%s" limited="%s" frictionloss="%s"/>' % (
            pad, j["name"], j["type"], _f(j["axis"]), _f(j["range"]), "true" if j["limited"] else "false", repr(float(j["frictionloss"]))))
        out.append('%s  <inertial pos="%s" mass="%s" diaginertia="%s"/>' % (pad, _f(inr["com"]), repr(float(inr["mass"])), _f(inr["diaginertia"])))
        for k, s in enumerate(spheres):
            if s["link"] != i:
                continue
            seg = s.get("seg")
            has_seg = seg is not None and any(float(x) != 0.0 for x in seg)
            ct, ca = BITS["seg_sphere" if has_seg else "sphere"]
            alpha = 1 if s.get("visible", 1) else 0
            name = s["name"] + ("__seg" if seg is not None else "")      # (a link-capsule end: the reader restores `seg`, zero or not)
            out.append('%s  <geom name="%s" type="sphere" size="%s" pos="%s" contype="%d" conaffinity="%d" user="%d" rgba="0.6 0.6 0.6 %d"/>' % (
                pad, name, repr(float(s["radius"])), _f(s["pos"]), ct, ca, k, alpha))
            if has_seg:
                ct, ca = BITS["capsule"]
                end = [float(s["pos"][c]) + float(seg[c]) for c in range(3)]
                out.append('%s  <geom name="%s__capsule" type="capsule" size="%s" fromto="%s %s" contype="%d" conaffinity="%d" user="%d" rgba="0.6 0.6 0.6 0"/>' % (
                    pad, s["name"], repr(float(s["radius"])), _f(s["pos"]), _f(end), ct, ca, k))
        emit_frame_children(i, ind + 1)
        for c in range(len(links)):
            if links[c]["parent"] == i:
                emit_link(c, ind + 1)
        out.append('%s</body>' % pad)

    emit_frame_children(-1, 2)
    for i, l in enumerate(links):
        if l["parent"] == -1:
            emit_link(i, 2)
    tb, cube = spec["table"], spec["cube"]
    x0, x1, y0, y1 = [float(v) for v in tb["rect"]]
    cx, cy = 0.5 * (x0 + x1) - float(tb["pos"][0]), 0.5 * (y0 + y1) - float(tb["pos"][1])
    out.append('    <body name="table" pos="%s">' % _f(tb["pos"]))
    out.append('      <geom name="table" type="box" size="%s" pos="%s" contype="%d" conaffinity="%d" rgba="0.2 0.2 0.2 1"/>' % (
        _f([0.5 * (x1 - x0), 0.5 * (y1 - y0), 0.5 * TABLE_THICKNESS]),
        _f([cx, cy, float(tb["plane_z"]) - float(tb["pos"][2]) - 0.5 * TABLE_THICKNESS]), *BITS["table"]))
    out.append('    </body>')
    out.append('    <body name="cube" pos="%s" quat="%s">' % (_f(cube["pos0"]), _f(cube["quat0"])))
    out.append('      <joint name="cube_joint" type="free" frictionloss="%s"/>' % repr(float(cube["frictionloss"])))
    out.append('      <inertial pos="0 0 0" mass="%s" diaginertia="%s"/>' % (repr(float(cube["mass"])), _f(cube["diaginertia"])))
    out.append('      <geom name="cube" type="box" size="%s" pos="0 0 0" condim="%d" solimp="%s" solref="%s" friction="%s" contype="%d" conaffinity="%d" rgba="1 0 0 1"/>' % (
        _f(cube["half_size"]), int(cube["condim"]), _f(cube["solimp"]), _f(cube["solref"]), _f(cube["friction"]), *BITS["cube"]))
    out.append('    </body>')
    out.append('  </worldbody>')
    out.append('  <actuator>')
    for l in links:
        a = l["actuator"]
        fr = ' forcelimited="true" forcerange="%s"' % _f(a["forcerange"]) if a["forcerange"] is not None else ""
        out.append('    <position name="act_%s" joint="%s" kp="%s" ctrllimited="true" ctrlrange="%s"%s/>' % (
            l["joint"]["name"], l["joint"]["name"], repr(float(a["kp"])), _f(a["ctrlrange"]), fr))
    out.append('  </actuator>')
    out.append('</mujoco>')
    return "\n".join(out) + "\n"


def _qconj(q):
    return [q[0], -q[1], -q[2], -q[3]]


def _qmul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return [aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
            aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw]


def _rel_quat(qt, qs):
    """site orientation in the target body's frame: conj(qt) * qs."""
    return _qmul(_qconj([float(x) for x in qt]), [float(x) for x in qs])


def _same_frame(s, t, sname, tname):
    """a site belongs to the fixed body whose name it extends (eer_site -> eer_site_pos, hand_r -> hand_r_pos, ...)."""
    return sname.startswith(tname + "_")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=".")
    ap.add_argument("names", nargs="*", default=["solo_arm", "dual_arm", "torso"])
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    for n in a.names:
        spec = json.load(open(os.path.join(ASSET_DIR, n + ".json")))
        p = os.path.join(a.out, n + ".xml")
        with open(p, "w") as f:
            f.write(export(spec))
        print(p)


if __name__ == "__main__":
    sys.exit(main())
