"""tools/mjcf_export.py: the surrogate model as primitive-only MJCF text (SURVEY 8d plan (1)).

MuJoCo cannot run on this pool (`mujoco` absent here and on the GPU box), so what is tested is the text: it parses, it states
every number of the model, its contact bits give exactly the surrogate's pair set, and it ROUND-TRIPS through
tools/mjcf_extract.py -- the reader that produced gym_kmanip_amd/assets/*.json from the reference's MJCF -- to the same JSON."""
import json
import os
import sys
import xml.etree.ElementTree as ET

import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import mjcf_export as E  # noqa: E402
import mjcf_extract as X  # noqa: E402

NAMES = ["solo_arm", "dual_arm", "torso"]


def _asset(name):
    return json.load(open(os.path.join(ROOT, "gym_kmanip_amd", "assets", name + ".json")))


def _diff(a, b, path="", out=None):
    out = [] if out is None else out
    if isinstance(a, dict) and isinstance(b, dict):
        for k in set(a) | set(b):
            if k not in a or k not in b:
                out.append((path + "/" + k, "missing on one side"))
            else:
                _diff(a[k], b[k], path + "/" + k, out)
    elif isinstance(a, list) and isinstance(b, list):
        if len(a) != len(b):
            out.append((path, "length %d vs %d" % (len(a), len(b))))
        else:
            for i, (x, y) in enumerate(zip(a, b)):
                _diff(x, y, "%s/%d" % (path, i), out)
    elif isinstance(a, (int, float)) and isinstance(b, (int, float)) and not isinstance(a, bool):
        if abs(a - b) > 1e-15:
            out.append((path, (a, b)))
    elif a != b:
        out.append((path, (a, b)))
    return out


@pytest.mark.parametrize("name", NAMES)
def test_export_round_trips_through_the_extractor(name, tmp_path):
    old = _asset(name)
    (tmp_path / (name + ".xml")).write_text(E.export(old))
    back = json.loads(json.dumps(X.build(name + ".xml", name, assets_dir=str(tmp_path))))
    # `source` names the file read; a sphere's `site` is an informational field of the generator (no consumer reads it)
    for s in old["spheres"]:
        s.pop("site", None)
    old["source"] = back["source"]
    assert _diff(old, back) == []


@pytest.mark.parametrize("name", NAMES)
def test_exported_text_is_primitive_only_and_complete(name):
    spec = _asset(name)
    root = ET.fromstring(E.export(spec))
    assert root.tag == "mujoco" and root.find("asset") is None and not root.findall(".//geom[@type='mesh']")
    bodies = {b.get("name"): b for b in root.iter("body")}
    for l in spec["links"]:                                     # every link: one joint, stated inertial
        b = bodies[l["name"]]
        (j,) = b.findall("joint")
        assert j.get("name") == l["joint"]["name"] and b.find("inertial") is not None
    acts = root.find("actuator").findall("position")
    assert [a.get("joint") for a in acts] == [l["joint"]["name"] for l in spec["links"]]      # actuator i drives joint i (env_sim.py:45,55)
    assert all(a.get("ctrllimited") == "true" for a in acts)
    assert bodies["cube"].find("joint").get("type") == "free" and bodies["cube"].find("geom").get("condim") == "4"
    for m in ("hand_r", "hand_l"):
        if m in spec["targets"]:
            assert bodies[m].get("mocap") == "true"              # ik / before_step write data.mocap_pos (env_sim.py:70-72)
    assert {c.get("name") for c in root.iter("camera")} == {c["name"] for c in spec["cameras"]}
    assert {s.get("name") for s in root.iter("site")} == set(spec["sites"])


def test_contact_bits_give_the_surrogate_pair_set():
    """MuJoCo collides two geoms iff (contype1 & conaffinity2) | (contype2 & conaffinity1)."""
    hit = lambda a, b: bool((E.BITS[a][0] & E.BITS[b][1]) | (E.BITS[b][0] & E.BITS[a][1]))
    want = {("cube", "table"), ("sphere", "cube"), ("sphere", "table"), ("seg_sphere", "table"), ("capsule", "cube")}
    kinds = list(E.BITS)
    for i, a in enumerate(kinds):
        for b in kinds[i:]:
            assert hit(a, b) == ((a, b) in want or (b, a) in want), (a, b)


def test_bench_probes_mujoco_at_run_time():
    import importlib.util
    sys.path.insert(0, ROOT)
    import bench
    rec = bench.probe_mujoco()
    assert rec["present"] == (importlib.util.find_spec("mujoco") is not None)
    assert "probed" in rec["note"]
