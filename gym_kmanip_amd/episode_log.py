"""Episode logger fed from device buffers, in the reference's ACT / LeRobot HDF5 layout.

The reference writes one `episode_<n>.hdf5` per episode from inside `KManipEnv.step` (gym_kmanip/log_h5py.py:13-61,
called at env_base.py:231-263): attrs `sim`, group `metadata` (the `info` dict as attrs), float32 datasets
`observations/qpos [64, q_len]`, `observations/qvel [64, q_len]`, `action [64, a_len]`, group `observations/images`
and, per camera (log_h5py.cam, :36-46), a group `metadata/camera/<name>` with attrs `resolution`, `focal_length`,
`principal_point` plus a uint8 dataset `observations/images/<name> [64, h, w, 3]` chunked one frame at a time.
It flushes the file every step of its single env.  Here a whole batch is logged: every step appends the batch's
observation / action rows (and the selected envs' camera frames) to device-resident rings (no host traffic on the step
path); at the episode boundary the rings cross PCIe once and one file per selected env is written with the same
internal paths.

`h5py` is not installable in the build image or on the GPU box (profiles/r02_probe_imports.txt), so the writer is chosen
at run time: `h5py` when importable (the reference's layout, through the same calls log_h5py.py makes), otherwise `.npz`
archives whose member names are the HDF5 dataset paths (`observations/qpos`, ...) plus a `metadata` JSON member -- the
same tree, loadable with numpy alone.

Against the reference's own logger.  tests/golden/ref_h5_tree_<id>.json is the tree log_h5py.new / cam / step / end build when they
run, unmodified, inside the reference's KManipEnv(log_h5py=True) against a recording h5py stand-in (tests/tools/h5_recorder.py);
tests/test_episode_log.py and tests/test_gpu_ref_fixtures.py compare this logger's h5py branch with it node for node.
`reference_action_quirk=True` is the reference's tree exactly, quirks included:
  * `action` is [64, a_len] where a_len = the NUMBER OF KEYS of the action Dict (info["a_len"] = len(action_space.spaces),
    env_base.py:190: 3 for SoloArm, 6 for DualArm / Torso), and every row holds action["grip_r"] broadcast over it
    (log_h5py.py:28,55: `f["action"][id] = action["grip_r"]`) -- the end-effector deltas are never logged;
  * `metadata` carries the info dict as it is at reset() (env_base.py:222-232): step 0, episode, is_success False, q_keys, q_len,
    a_len, obs_list, act_list, sim, sim_time 0.0, cpu_time, terminated False; `reward` (None at reset) and a non-empty
    `cameras` list (dataclass instances) have no HDF5 type and are skipped (log_h5py.py:20-24), an empty `cameras` list is stored.
The DEFAULT layout deviates from that fixture in two documented places: `action` is [64, act_dim] holding the flat float32
action row the step ran on (Dict-space insertion order, include/kmanip.h) -- what a learner needs --, and `metadata` has the
batch's extras.  In both layouts a file is one env of a batch: it is named `episode_<n>_env<e>` (reference: `episode_<n>`) and
`metadata` has two extra attrs, `env` (the env's index) and `steps` (control steps logged before the file was written).
"""
from __future__ import annotations

import json
import os
from typing import Any, Dict, Iterable, Optional

import numpy as np

from .model import MAX_EPISODE_STEPS

H5PY_CHUNK_SIZE_BYTES = 1024 ** 2 * 2          # gym_kmanip/__init__.py:216 (rdcc_nbytes of the reference's h5py.File)


def _import_h5py():
    try:
        import h5py
        return h5py
    except Exception:  # noqa: BLE001
        return None


class _H5Tree:
    """The calls log_h5py.py makes, on an h5py(-compatible) module: File, attrs, create_group, create_dataset."""
    suffix = ".hdf5"

    def __init__(self, stem, h5):
        self.path = stem + self.suffix
        self.f = h5.File(self.path, "w", rdcc_nbytes=H5PY_CHUNK_SIZE_BYTES)

    def root_attr(self, key, value):
        self.f.attrs[key] = value

    def group_attrs(self, path, attrs):
        g = self.f.create_group(path)
        for k, v in attrs.items():
            try:
                g.attrs[k] = v
            except TypeError:                      # log_h5py.py:22-23: values h5py cannot store are skipped
                pass

    def group(self, path):
        self.f.create_group(path)

    def dataset(self, path, data, chunks=None):
        kw = {"chunks": chunks} if chunks is not None else {}
        self.f.create_dataset(path, data=data, **kw)

    def close(self):
        self.f.close()
        return self.path


class _NpzTree:
    """Same tree as members of one .npz: dataset paths are member names, groups' attrs go into a `metadata` JSON member."""
    suffix = ".npz"

    def __init__(self, stem, _unused=None):
        self.path = stem + self.suffix
        self.members, self.meta = {}, {}

    def root_attr(self, key, value):
        self.meta.setdefault("/", {})[key] = value

    def group_attrs(self, path, attrs):
        self.meta[path] = {k: (list(v) if isinstance(v, tuple) else v) for k, v in attrs.items() if v is not None}

    def group(self, path):
        self.meta.setdefault(path, {})

    def dataset(self, path, data, chunks=None):
        self.members[path.lstrip("/")] = data

    def close(self):
        flat = dict(self.meta.get("metadata", {}))
        flat["_groups"] = {k: v for k, v in self.meta.items() if k != "metadata"}
        np.savez(self.path, **self.members, metadata=np.frombuffer(json.dumps(flat, default=str).encode(), dtype=np.uint8))
        return self.path


class EpisodeLogger:
    def __init__(self, log_dir: str, num_envs: int, q_len: int, a_len: int, device="cpu", env_ids: Optional[Iterable[int]] = None,
                 info: Optional[Dict[str, Any]] = None, grip_r_col: Optional[int] = None,
                 reference_action_quirk: bool = False, backend: Optional[str] = None, h5py_module=None,
                 ref_a_len: Optional[int] = None):
        """a_len: width of the flat action row (act_dim).  reference_action_quirk: the reference's `action` dataset and `metadata`
        attrs exactly (module docstring); then ref_a_len (default info["a_len"]) is the number of action keys."""
        import torch
        assert os.path.exists(log_dir), f"Directory {log_dir} does not exist"      # log_h5py.py:14
        self.torch = torch
        self.device = device
        self.log_dir, self.n, self.q_len, self.a_len = log_dir, num_envs, q_len, a_len
        self.env_ids = list(range(num_envs)) if env_ids is None else list(env_ids)
        self.info = dict(info or {})
        self.grip_r_col = grip_r_col
        self.quirk = reference_action_quirk
        if self.quirk and grip_r_col is None:
            raise ValueError("reference_action_quirk needs the grip_r column of the flat action")
        # width of the logged action rows: the flat action, or -- the reference's -- one column per action KEY
        self.log_a_len = a_len if not self.quirk else int(ref_a_len if ref_a_len is not None else self.info.get("a_len", a_len))
        self.h5 = h5py_module or _import_h5py()
        self.backend = backend or ("h5py" if self.h5 is not None else "npz")
        if self.backend == "h5py" and self.h5 is None:
            raise RuntimeError("h5py is not importable here; use backend='npz'")
        T = MAX_EPISODE_STEPS
        self.qpos = torch.zeros((T, num_envs, q_len), dtype=torch.float32, device=device)
        self.qvel = torch.zeros((T, num_envs, q_len), dtype=torch.float32, device=device)
        self.action = torch.zeros((T, num_envs, self.log_a_len), dtype=torch.float32, device=device)
        self.cams = {}                 # name -> (Cam, ring uint8 [T, len(env_ids), h, w, c])
        self._frames_due = set()       # rows logged with images_later=True whose frames have not arrived
        self._sel = torch.as_tensor(self.env_ids, dtype=torch.long, device=device)
        self.t = 0
        self.episode = 0
        self.cpu_time0 = None          # info["cpu_time"] of the episode's reset (env_base.py:226)

    def cam(self, cam) -> None:
        """log_h5py.cam (:36-46): register a camera -- its metadata group and a uint8 image dataset per episode.  Frames of
        the selected envs only are kept (a 480x640 head frame is 0.9 MB per env and step)."""
        torch = self.torch
        ring = torch.zeros((MAX_EPISODE_STEPS, len(self.env_ids), cam.h, cam.w, cam.c), dtype=torch.uint8, device=self.device)
        self.cams[cam.name] = (cam, ring)

    def step(self, act_flat, obs_q_pos, obs_q_vel, images: Optional[Dict[str, Any]] = None, images_later: bool = False) -> int:
        """Append one control step (log_h5py.step): device-to-device copies only.  `images`: camera name -> uint8
        [num_envs, h, w, 3] (e.g. KManipEnvHip.render_rgb) for every registered camera.  images_later: the frames of this step
        are still being rendered (pipeline.RenderBehind): hand them to `late_images(t, images)` with the returned row index t
        before end_episode().  Returns t."""
        if self.t >= MAX_EPISODE_STEPS:
            raise RuntimeError("episode longer than MAX_EPISODE_STEPS: call end_episode() at the TimeLimit boundary")
        if self.t == 0 and self.cpu_time0 is None:
            import time
            self.cpu_time0 = time.time()
        self.qpos[self.t].copy_(obs_q_pos)          # float64 obs -> float32 datasets, as h5py's default dtype does
        self.qvel[self.t].copy_(obs_q_vel)
        if self.quirk:
            self.action[self.t].copy_(act_flat[:, self.grip_r_col:self.grip_r_col + 1].expand(-1, self.log_a_len))
        else:
            self.action[self.t].copy_(act_flat)
        if images_later:
            self._frames_due.add(self.t)
        else:
            self._put_frames(self.t, images)
        self.t += 1
        return self.t - 1

    def _put_frames(self, t, images):
        for name, (cam, ring) in self.cams.items():
            if images is None or name not in images and cam.log_name not in images:
                raise KeyError("no frame for registered camera %r in this step" % name)
            img = images[name] if name in images else images[cam.log_name]
            ring[t].copy_(img.index_select(0, self._sel))

    def late_images(self, t: int, images: Dict[str, Any]) -> None:
        """The frames of row t, logged with `step(..., images_later=True)` (copies on the CURRENT stream: call it after
        RenderBehind.images(), which orders that stream behind the render)."""
        if t not in self._frames_due:
            raise KeyError("row %d is not waiting for frames" % t)
        self._put_frames(t, images)
        self._frames_due.discard(t)

    def _metadata(self, e):
        if not self.quirk:
            return dict(self.info, episode=self.episode, env=e, steps=self.t, q_len=self.q_len, a_len=self.a_len)
        # the info dict as the reference's reset() leaves it (env_base.py:201-212,222-229), in its insertion order; `reward` is None
        # there and a Vision id's `cameras` holds dataclass instances: both are offered to the writer, which skips what has no
        # HDF5 type (log_h5py.py:20-24)
        base = self.info
        meta = {"step": 0, "episode": self.episode, "is_success": False}
        for k in ("q_keys", "q_len", "a_len", "obs_list", "act_list", "cameras", "sim"):
            if k in base:
                meta[k] = base[k]
        meta.setdefault("q_len", self.q_len)
        meta["a_len"] = self.log_a_len
        meta.update(sim_time=0.0, cpu_time=float(self.cpu_time0 or 0.0), reward=None, terminated=False)
        meta.update(env=e, steps=self.t)            # the batch's two extras (module docstring)
        return meta

    def end_episode(self):
        """Write `episode_<n>_env<e>` files for the selected envs (one PCIe crossing for the whole batch)."""
        if self._frames_due:
            raise RuntimeError("end_episode(): rows %s were logged with images_later=True and their frames never arrived" % sorted(self._frames_due))
        self.episode += 1
        qpos = self.qpos.cpu().numpy(); qvel = self.qvel.cpu().numpy(); action = self.action.cpu().numpy()
        frames = {name: ring.cpu().numpy() for name, (cam, ring) in self.cams.items()}
        Tree = _H5Tree if self.backend == "h5py" else _NpzTree
        paths = []
        for k, e in enumerate(self.env_ids):
            meta = self._metadata(e)
            tree = Tree(os.path.join(self.log_dir, "episode_%d_env%d" % (self.episode, e)), self.h5)
            tree.root_attr("sim", bool(meta.get("sim", True)))                       # log_h5py.py:18
            tree.group_attrs("metadata", meta)                                       # :19-24
            tree.group("observations/images")                                        # :25
            tree.dataset("observations/qpos", qpos[:, e])                            # :26-28
            tree.dataset("observations/qvel", qvel[:, e])
            tree.dataset("action", action[:, e])
            for name, (cam, _) in self.cams.items():                                 # log_h5py.cam :36-46
                tree.group_attrs("metadata/" + cam.log_name, {"resolution": [cam.w, cam.h], "focal_length": cam.fl,
                                                              "principal_point": cam.pp})
                tree.dataset("/observations/images/" + cam.name, frames[name][:, k], chunks=(1, cam.h, cam.w, cam.c))
            paths.append(tree.close())
        self.t = 0
        self.cpu_time0 = None
        self.qpos.zero_(); self.qvel.zero_(); self.action.zero_()
        for _, ring in self.cams.values():
            ring.zero_()
        return paths
