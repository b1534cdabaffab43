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
