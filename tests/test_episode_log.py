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


class _FakeAttrs(dict):
    def __setitem__(self, k, v):
        if isinstance(v, (dict, set)) or (isinstance(v, list) and v and not isinstance(v[0], (int, float, str))):
            raise TypeError("h5py cannot store %r" % type(v))         # what h5py does for objects without an HDF5 type
        super().__setitem__(k, v)


class _FakeNode:
    def __init__(self):
        self.attrs, self.children = _FakeAttrs(), {}

    def create_group(self, path):
        node = self
        for part in path.strip("/").split("/"):
            node = node.children.setdefault(part, _FakeNode())
        return node

    def create_dataset(self, path, data=None, chunks=None):
        parts = path.strip("/").split("/")
        node = self.create_group("/".join(parts[:-1])) if len(parts) > 1 else self
        node.children[parts[-1]] = ("dataset", np.array(data), chunks)


class _FakeH5:
    """h5py's File / Group / attrs surface that log_h5py.py uses, in memory -- executes the logger's h5py branch where the real
    module cannot be installed (the real one is exercised by test_h5py_file_when_available wherever it exists)."""
    files = {}

    class File(_FakeNode):
        def __init__(self, path, mode, rdcc_nbytes=None):
            super().__init__()
            assert mode == "w" and rdcc_nbytes == 2 * 1024 ** 2
            self.path, self.closed = path, False

        def close(self):
            self.closed = True
            _FakeH5.files[self.path] = self


def _episode(lg, n, q_len, a_len, cams=()):
    rng = np.random.default_rng(1)
    for t in range(MAX_EPISODE_STEPS):
        imgs = {c.name: torch.from_numpy(rng.integers(0, 256, (n, c.h, c.w, 3), dtype=np.uint8)) for c in cams}
        lg.step(torch.from_numpy(rng.uniform(-1, 1, (n, a_len)).astype(np.float32)), torch.from_numpy(rng.uniform(0, 1, (n, q_len))),
                torch.from_numpy(rng.uniform(-1, 1, (n, q_len))), images=imgs or None)
        last = imgs
    return last


def test_h5py_branch_builds_the_reference_tree(tmp_path):
    """log_h5py.new / cam / step restated: root attr `sim`, `metadata` attrs (unstorable values skipped), `observations/images`
    group, float32 qpos / qvel / action datasets, and per camera `metadata/camera/<name>` attrs + a uint8
    `observations/images/<name>` dataset chunked one frame at a time."""
    from gym_kmanip_amd.model import CAMERAS
    cm = compile_model("KManipSoloArmVision")
    n, q_len, a_len = 3, 10, cm.act_dim
    lg = EpisodeLogger(str(tmp_path), n, q_len, a_len, env_ids=[2], info={"sim": True, "q_len": q_len, "cameras": {"not": "storable"}},
                       backend="h5py", h5py_module=_FakeH5)
    cam = CAMERAS["grip_r"]
    lg.cam(cam)
    last = _episode(lg, n, q_len, a_len, [cam])
    (path,) = lg.end_episode()
    f = _FakeH5.files[path]
    assert path.endswith("episode_1_env2.hdf5") and f.closed and f.attrs["sim"] is True
    meta = f.children["metadata"]
    assert meta.attrs["episode"] == 1 and meta.attrs["env"] == 2 and "cameras" not in meta.attrs
    cam_meta = meta.children["camera"].children["grip_r"].attrs
    assert cam_meta["resolution"] == [60, 40] and cam_meta["focal_length"] == 45 and cam_meta["principal_point"] == [30, 20]
    obs = f.children["observations"].children
    kind, qp, _ = obs["qpos"]
    assert kind == "dataset" and qp.shape == (MAX_EPISODE_STEPS, q_len) and qp.dtype == np.float32
    kind, img, chunks = obs["images"].children["grip_r"]
    assert img.shape == (MAX_EPISODE_STEPS, 40, 60, 3) and img.dtype == np.uint8 and chunks == (1, 40, 60, 3)
    assert np.array_equal(img[-1], last["grip_r"][2].numpy())
    assert f.children["action"][1].shape == (MAX_EPISODE_STEPS, a_len)


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
