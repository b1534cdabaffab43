"""Episode logger fed from device buffers, in the reference's ACT / LeRobot HDF5 layout.

The reference writes one `episode_<n>.hdf5` per episode from inside `KManipEnv.step` (gym_kmanip/log_h5py.py:13-61,
called at env_base.py:231-263): attrs `sim`, group `metadata` (the `info` dict as attrs), float32 datasets
`observations/qpos [64, q_len]`, `observations/qvel [64, q_len]`, `action [64, a_len]`, group `observations/images`.
It flushes the file every step of its single env.  Here a whole batch is logged: every step appends the batch's
observation / action rows to a device-resident ring `[64, num_envs, width]` (no host traffic on the step path); at the
episode boundary the ring crosses PCIe once and one file per selected env is written with the same internal paths.

`h5py` is not installable in the build image, so the writer is chosen at run time: `h5py` when importable (exact
reference layout), otherwise `.npz` archives whose member names are the HDF5 dataset paths (`observations/qpos`, ...)
plus a `metadata` JSON member -- the same tree, loadable with numpy alone.  One reference quirk is NOT copied by
default: log_h5py.py:55 stores `action["grip_r"]` broadcast over the whole action row; `reference_action_quirk=True`
reproduces that, the default stores the flat action row (Dict-space insertion order, include/kmanip.h).
"""
from __future__ import annotations

import json
import os
from typing import Any, Dict, Iterable, Optional

import numpy as np

from .model import MAX_EPISODE_STEPS

try:  # pragma: no cover - absent in the build image
    import h5py as _h5py
except Exception:  # noqa: BLE001
    _h5py = None


class EpisodeLogger:
    def __init__(self, log_dir: str, num_envs: int, q_len: int, a_len: int, device="cpu", env_ids: Optional[Iterable[int]] = None,
                 info: Optional[Dict[str, Any]] = None, grip_r_col: Optional[int] = None,
                 reference_action_quirk: bool = False, backend: Optional[str] = None):
        import torch
        assert os.path.exists(log_dir), f"Directory {log_dir} does not exist"      # log_h5py.py:14
        self.torch = torch
        self.log_dir, self.n, self.q_len, self.a_len = log_dir, num_envs, q_len, a_len
        self.env_ids = list(range(num_envs)) if env_ids is None else list(env_ids)
        self.info = dict(info or {})
        self.grip_r_col = grip_r_col
        self.quirk = reference_action_quirk
        if self.quirk and grip_r_col is None:
            raise ValueError("reference_action_quirk needs the grip_r column of the flat action")
        self.backend = backend or ("h5py" if _h5py is not None else "npz")
        if self.backend == "h5py" and _h5py is None:
            raise RuntimeError("h5py is not importable here; use backend='npz'")
        T = MAX_EPISODE_STEPS
        self.qpos = torch.zeros((T, num_envs, q_len), dtype=torch.float32, device=device)
        self.qvel = torch.zeros((T, num_envs, q_len), dtype=torch.float32, device=device)
        self.action = torch.zeros((T, num_envs, a_len), dtype=torch.float32, device=device)
        self.t = 0
        self.episode = 0

    def step(self, act_flat, obs_q_pos, obs_q_vel) -> None:
        """Append one control step (log_h5py.step): device-to-device copies only."""
        if self.t >= MAX_EPISODE_STEPS:
            raise RuntimeError("episode longer than MAX_EPISODE_STEPS: call end_episode() at the TimeLimit boundary")
        self.qpos[self.t].copy_(obs_q_pos)          # float64 obs -> float32 datasets, as h5py's default dtype does
        self.qvel[self.t].copy_(obs_q_vel)
        if self.quirk:
            self.action[self.t].copy_(act_flat[:, self.grip_r_col:self.grip_r_col + 1].expand(-1, self.a_len))
        else:
            self.action[self.t].copy_(act_flat)
        self.t += 1

    def end_episode(self):
        """Write `episode_<n>_env<e>` files for the selected envs (one PCIe crossing for the whole batch)."""
        self.episode += 1
        qpos = self.qpos.cpu().numpy(); qvel = self.qvel.cpu().numpy(); action = self.action.cpu().numpy()
        paths = []
        for e in self.env_ids:
            meta = dict(self.info, episode=self.episode, env=e, steps=self.t, q_len=self.q_len, a_len=self.a_len)
            stem = os.path.join(self.log_dir, "episode_%d_env%d" % (self.episode, e))
            if self.backend == "h5py":  # pragma: no cover - exercised only where h5py exists
                f = _h5py.File(stem + ".hdf5", "w")
                f.attrs["sim"] = bool(meta.get("sim", True))
                g = f.create_group("metadata")
                for k, v in meta.items():
                    try:
                        g.attrs[k] = v
                    except TypeError:
                        pass
                f.create_group("observations/images")
                f.create_dataset("observations/qpos", data=qpos[:, e])
                f.create_dataset("observations/qvel", data=qvel[:, e])
                f.create_dataset("action", data=action[:, e])
                f.close()
                paths.append(stem + ".hdf5")
            else:
                np.savez(stem + ".npz", **{"observations/qpos": qpos[:, e], "observations/qvel": qvel[:, e],
                                           "action": action[:, e],
                                           "metadata": np.frombuffer(json.dumps(meta, default=str).encode(), dtype=np.uint8)})
                paths.append(stem + ".npz")
        self.t = 0
        self.qpos.zero_(); self.qvel.zero_(); self.action.zero_()
        return paths
