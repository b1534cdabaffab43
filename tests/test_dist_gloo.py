"""N > 1 path on CPU: world_size-2 gloo ranks, each stepping its contiguous env shard (here with the oracle
standing in for the GPU engine -- the HIP library has no CPU path), the per-step (reward, done) all-gather,
and equality with a single-process run of all envs (results must not depend on the shard layout)."""
import os
import socket

import numpy as np
import pytest

from gym_kmanip_amd.dist import RewardDoneGather, shard_range
from gym_kmanip_amd.model import compile_model


def test_shard_range_partition():
    for total, world in [(4096, 8), (65536, 8), (10, 3), (7, 8)]:
        spans = [shard_range(total, world, r) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n_total, steps, out_dir):
    import torch
    import torch.distributed as dist
    from oracle.oracle import Oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cm = compile_model("KManipSoloArm", auto_reset=True)
    lo, hi = shard_range(n_total, world, rank)
    n = hi - lo
    eng = Oracle(cm, n, seed=9, env_id_offset=lo)
    eng.reset()
    g = RewardDoneGather(n, world, torch.device("cpu"), dist)
    rng = np.random.default_rng(77)
    rec = []
    for k in range(steps):
        act_all = rng.uniform(-1, 1, (n_total, cm.act_dim)).astype(np.float32)   # same stream on every rank
        obs, rew, done = eng.step(act_all[lo:hi])
        b = g.post(torch.from_numpy(rew), torch.from_numpy(done))
        r_all, d_all = g.result(b)
        rec.append((r_all.numpy().copy(), d_all.numpy().copy()))
    if rank == 0:
        np.savez(os.path.join(out_dir, "gathered.npz"), rew=np.array([r for r, _ in rec]), done=np.array([d for _, d in rec]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_matches_single_process(tmp_path):
    import torch.multiprocessing as mp
    from oracle.oracle import Oracle
    n_total, steps, world = 12, 5, 2
    mp.spawn(_worker, args=(world, _free_port(), n_total, steps, str(tmp_path)), nprocs=world, join=True)
    got = np.load(os.path.join(str(tmp_path), "gathered.npz"))
    cm = compile_model("KManipSoloArm", auto_reset=True)
    ref = Oracle(cm, n_total, seed=9, env_id_offset=0)
    ref.reset()
    rng = np.random.default_rng(77)
    for k in range(steps):
        act_all = rng.uniform(-1, 1, (n_total, cm.act_dim)).astype(np.float32)
        obs, rew, done = ref.step(act_all)
        assert np.array_equal(got["rew"][k], rew), k
        assert np.array_equal(got["done"][k], done), k
