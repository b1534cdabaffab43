"""Host logic of the batched episode logger on CPU tensors: tree layout of reference
gym_kmanip/log_h5py.py:13-61, float32 [64, width] datasets, one file per selected env, ring reuse across episodes."""
import json

import numpy as np
import pytest
import torch

from gym_kmanip_amd.episode_log import EpisodeLogger
from gym_kmanip_amd.model import MAX_EPISODE_STEPS, compile_model


def _load_meta(z):
    return json.loads(bytes(z["metadata"]).decode())


import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
import h5_recorder as H5  # noqa: E402  (the recording h5py stand-in the reference's own logger was run against)

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _episode(lg, n, q_len, a_len, cams=()):
    rng = np.random.default_rng(1)
    for t in range(MAX_EPISODE_STEPS):
        imgs = {c.name: torch.from_numpy(rng.integers(0, 256, (n, c.h, c.w, 3), dtype=np.uint8)) for c in cams}
        lg.step(torch.from_numpy(rng.uniform(-1, 1, (n, a_len)).astype(np.float32)), torch.from_numpy(rng.uniform(0, 1, (n, q_len))),
                torch.from_numpy(rng.uniform(-1, 1, (n, q_len))), images=imgs or None)
        last = imgs
    return last


def ref_tree(env_id):
    return json.load(open(os.path.join(GOLDEN, "ref_h5_tree_%s.json" % env_id)))


def shell_info(env_id):
    """The constructor part of KManipEnv.info (env_base.py:201-212) as gym_shell builds it, without a device."""
    from gym_kmanip_amd import gym_shell
    from gym_kmanip_amd.model import CAMERAS, ENV_SPECS
    spec = ENV_SPECS[env_id]
    sp = gym_shell.spaces_for(env_id)
    return {"step": 0, "episode": 0, "is_success": False, "q_keys": gym_shell.q_keys_for(env_id), "q_len": len(spec.q_pos_home),
            "a_len": len(sp["action"]), "obs_list": list(spec.obs_list), "act_list": list(spec.act_list),
            "cameras": [CAMERAS[o.split("/")[-1]] for o in spec.obs_list if "camera" in o], "sim": True}


def assert_same_tree(ours, ref, path="", extra_attrs=()):
    """Node for node: the same attrs (dtype, shape, value where the fixture keeps one), groups and datasets (shape, dtype, chunks).
    extra_attrs: attrs ours may have on top (the batch's `env` / `steps` in `metadata`)."""
    extra = set(extra_attrs) if path == "/metadata" else set()
    assert set(ours["attrs"]) - extra == set(ref["attrs"]), (path, set(ours["attrs"]) ^ set(ref["attrs"]))
    for k, r in ref["attrs"].items():
        o = ours["attrs"][k]
        assert o["dtype"] == r["dtype"] and o["shape"] == r["shape"], (path, k, o, r)
        if "value" in r:
            assert o["value"] == r["value"], (path, k, o["value"], r["value"])
    assert set(ours["datasets"]) == set(ref["datasets"]), (path, set(ours["datasets"]) ^ set(ref["datasets"]))
    for k, r in ref["datasets"].items():
        assert ours["datasets"][k] == r, (path, k, ours["datasets"][k], r)
    assert set(ours["groups"]) == set(ref["groups"]), (path, set(ours["groups"]) ^ set(ref["groups"]))
    for k in ref["groups"]:
        assert_same_tree(ours["groups"][k], ref["groups"][k], path + "/" + k, extra_attrs)


@pytest.mark.parametrize("env_id", ["KManipSoloArm", "KManipDualArm", "KManipTorso", "KManipSoloArmVision"])
def test_reference_layout_is_the_reference_loggers_tree(tmp_path, env_id):
    """EpisodeLogger(h5py branch, reference_action_quirk=True) against tests/golden/ref_h5_tree_<id>.json -- the tree the
    reference's own log_h5py.new / cam / step / end built inside its KManipEnv(log_h5py=True) (tests/tools/make_golden_ref.py)."""
    ref = ref_tree(env_id)
    cm = compile_model(env_id)
    info = shell_info(env_id)
    n, q_len = 3, info["q_len"]
    assert ref["tree"]["groups"]["metadata"]["attrs"]["a_len"]["value"] == info["a_len"] != cm.act_dim
    H5.FILES.clear()
    lg = EpisodeLogger(str(tmp_path), n, q_len, cm.act_dim, env_ids=[2], info=info, grip_r_col=cm.act_slices["grip_r"].start,
                       reference_action_quirk=True, backend="h5py", h5py_module=H5)
    for cam in info["cameras"]:
        lg.cam(cam)
    _episode(lg, n, q_len, cm.act_dim, info["cameras"])
    (path,) = lg.end_episode()
    f = H5.FILES[path]
    assert f.closed and f.rdcc_nbytes == ref["rdcc_nbytes"] and os.path.basename(path) == "episode_1_env2.hdf5"
    ours = H5.tree(f, skip_attr_values=("cpu_time",))
    assert_same_tree(ours, ref["tree"], extra_attrs=("env", "steps"))
    meta = ours["groups"]["metadata"]["attrs"]
    assert meta["env"]["value"] == 2 and meta["steps"]["value"] == MAX_EPISODE_STEPS
    assert "reward" not in meta and ("cameras" in meta) == (not info["cameras"])      # None / dataclass lists have no HDF5 type


def test_default_layout_differs_from_the_reference_tree_only_where_documented(tmp_path):
    """The default layout against the same fixture: `action` is [64, act_dim] (the flat row) instead of the reference's
    [64, number of action keys] of broadcast grip_r, and `metadata` holds the caller's info + env / steps."""
    ref = ref_tree("KManipSoloArm")
    cm = compile_model("KManipSoloArm")
    H5.FILES.clear()
    lg = EpisodeLogger(str(tmp_path), 2, 10, cm.act_dim, info={"sim": True}, backend="h5py", h5py_module=H5)
    _episode(lg, 2, 10, cm.act_dim)
    ours = H5.tree(H5.FILES[lg.end_episode()[0]])
    assert ours["datasets"]["action"] == {"shape": [MAX_EPISODE_STEPS, cm.act_dim], "dtype": "float32", "chunks": None}
    assert ref["tree"]["datasets"]["action"]["shape"] == [MAX_EPISODE_STEPS, 3]
    assert ours["groups"]["observations"] == ref["tree"]["groups"]["observations"]
    assert ours["attrs"] == ref["tree"]["attrs"]


def test_reference_action_rows_hold_grip_r_only():
    """What the fixture says about log_h5py.py:55: every logged action row is action["grip_r"] broadcast over a_len columns."""
    z = np.load(os.path.join(GOLDEN, "ref_scripted_KManipSoloArm.npz"))
    cm = compile_model("KManipSoloArm")
    col = cm.act_slices["grip_r"].start
    assert z["h5/action"].shape == (MAX_EPISODE_STEPS, 3) and z["h5/action"].dtype == np.float32
    assert np.array_equal(z["h5/action"], np.repeat(z["action"][:, col:col + 1], 3, axis=1))
    assert np.array_equal(z["h5/observations/qpos"], z["obs"][:, :10].astype(np.float32))
    assert np.array_equal(z["h5/observations/qvel"], z["obs"][:, 10:20].astype(np.float32))


def test_h5py_file_when_available(tmp_path):
    h5py = pytest.importorskip("h5py")
    from gym_kmanip_amd.model import CAMERAS
    lg = EpisodeLogger(str(tmp_path), 2, 10, 7, info={"sim": True}, backend="h5py")
    lg.cam(CAMERAS["grip_r"])
    _episode(lg, 2, 10, 7, [CAMERAS["grip_r"]])
    with h5py.File(lg.end_episode()[0], "r") as f:
        assert f["observations/qpos"].shape == (MAX_EPISODE_STEPS, 10) and f["observations/images/grip_r"].dtype == np.uint8
        assert f["metadata/camera/grip_r"].attrs["focal_length"] == 45 and f.attrs["sim"]


def test_npz_images_and_camera_metadata(tmp_path):
    from gym_kmanip_amd.model import CAMERAS
    lg = EpisodeLogger(str(tmp_path), 4, 10, 7, env_ids=[1, 3], info={"sim": True}, backend="npz")
    lg.cam(CAMERAS["grip_r"])
    with pytest.raises(KeyError):
        lg.step(torch.zeros((4, 7)), torch.zeros((4, 10)), torch.zeros((4, 10)))       # a registered camera needs its frame
    last = _episode(lg, 4, 10, 7, [CAMERAS["grip_r"]])
    z = np.load(lg.end_episode()[1])
    assert z["observations/images/grip_r"].shape == (MAX_EPISODE_STEPS, 40, 60, 3)
    assert np.array_equal(z["observations/images/grip_r"][-1], last["grip_r"][3].numpy())
    assert _load_meta(z)["_groups"]["metadata/camera/grip_r"]["resolution"] == [60, 40]


def test_npz_tree_matches_reference_layout(tmp_path):
    cm = compile_model("KManipSoloArm")
    n, q_len, a_len = 5, 10, cm.act_dim
    lg = EpisodeLogger(str(tmp_path), n, q_len, a_len, env_ids=[0, 3], info={"sim": True, "obs_list": ["q_pos", "q_vel"]},
                       backend="npz")
    rng = np.random.default_rng(0)
    acts, qps, qvs = [], [], []
    for t in range(MAX_EPISODE_STEPS):
        a = torch.from_numpy(rng.uniform(-1, 1, (n, a_len)).astype(np.float32))
        qp = torch.from_numpy(rng.uniform(0, 1, (n, q_len))); qv = torch.from_numpy(rng.uniform(-1, 1, (n, q_len)))
        lg.step(a, qp, qv); acts.append(a.numpy()); qps.append(qp.numpy()); qvs.append(qv.numpy())
    with pytest.raises(RuntimeError):
        lg.step(a, qp, qv)                                  # the TimeLimit boundary must close the episode
    paths = lg.end_episode()
    assert [p.split("/")[-1] for p in paths] == ["episode_1_env0.npz", "episode_1_env3.npz"]
    z = np.load(paths[1])
    assert sorted(z.files) == ["action", "metadata", "observations/qpos", "observations/qvel"]
    assert z["observations/qpos"].shape == (MAX_EPISODE_STEPS, q_len) and z["observations/qpos"].dtype == np.float32
    assert z["action"].shape == (MAX_EPISODE_STEPS, a_len) and z["action"].dtype == np.float32
    assert np.array_equal(z["action"], np.stack(acts)[:, 3])
    assert np.allclose(z["observations/qpos"], np.stack(qps)[:, 3].astype(np.float32), rtol=0, atol=0)
    assert np.allclose(z["observations/qvel"], np.stack(qvs)[:, 3].astype(np.float32), rtol=0, atol=0)
    meta = _load_meta(z)
    assert meta["episode"] == 1 and meta["env"] == 3 and meta["steps"] == MAX_EPISODE_STEPS and meta["sim"] is True
    # next episode reuses the ring from step 0
    lg.step(torch.ones((n, a_len)), torch.ones((n, q_len), dtype=torch.float64), torch.zeros((n, q_len)))
    p2 = lg.end_episode()
    z2 = np.load(p2[0])
    assert _load_meta(z2)["episode"] == 2 and _load_meta(z2)["steps"] == 1
    assert (z2["action"][0] == 1).all() and (z2["action"][1:] == 0).all()


def test_reference_action_quirk(tmp_path):
    """log_h5py.py:55 writes action["grip_r"] into the whole action row; reproduced only on request."""
    cm = compile_model("KManipSoloArm")
    col = cm.act_slices["grip_r"].start
    lg = EpisodeLogger(str(tmp_path), 2, 10, cm.act_dim, grip_r_col=col, reference_action_quirk=True, backend="npz")
    a = torch.arange(2 * cm.act_dim, dtype=torch.float32).reshape(2, cm.act_dim)
    lg.step(a, torch.zeros((2, 10)), torch.zeros((2, 10)))
    z = np.load(lg.end_episode()[1])
    assert (z["action"][0] == a[1, col].item()).all()
    with pytest.raises(ValueError):
        EpisodeLogger(str(tmp_path), 2, 10, cm.act_dim, reference_action_quirk=True, backend="npz")


def test_frames_that_arrive_a_step_late(tmp_path):
    """step(..., images_later=True) + late_images(t, frames): the shape of a loop that renders behind its steps
    (pipeline.RenderBehind) -- the file is the one in-step frames give; frames that never arrive stop end_episode()."""
    from gym_kmanip_amd.model import CAMERAS
    cam = CAMERAS["grip_r"]
    gen = torch.Generator(); gen.manual_seed(5)
    rows = [(torch.rand((4, 7), generator=gen), torch.rand((4, 10), generator=gen), torch.rand((4, 10), generator=gen),
             {"grip_r": torch.randint(0, 255, (4, cam.h, cam.w, 3), generator=gen, dtype=torch.uint8)}) for _ in range(6)]
    os.makedirs(tmp_path / "a"); os.makedirs(tmp_path / "b")
    a = EpisodeLogger(str(tmp_path / "a"), 4, 10, 7, env_ids=[2], info={"sim": True}, backend="npz")
    b = EpisodeLogger(str(tmp_path / "b"), 4, 10, 7, env_ids=[2], info={"sim": True}, backend="npz")
    a.cam(cam); b.cam(cam)
    due = None
    for act, qp, qv, img in rows:
        a.step(act, qp, qv, images=img)
        t = b.step(act, qp, qv, images_later=True)
        if due is not None:
            b.late_images(*due)
        due = (t, img)
    with pytest.raises(RuntimeError, match="never arrived"):
        b.end_episode()
    with pytest.raises(KeyError):
        b.late_images(0, rows[0][3])                  # row 0 has its frames already
    b.late_images(*due)
    za, zb = np.load(a.end_episode()[0]), np.load(b.end_episode()[0])
    for k in ("observations/images/grip_r", "observations/qpos", "action"):
        assert np.array_equal(za[k], zb[k]), k
    assert za["observations/images/grip_r"][:6].any()
