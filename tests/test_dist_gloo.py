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


# ---------------------------------------------------------------------------------------------------------------------
# The bound-record ordering invariant (include/kmanip.h, dist.RewardDoneGather.before_step): a LAZY fake collective -- it
# reads its input only when somebody waits for it, i.e. as late as a lagging RCCL peer may make the real one -- and a fake
# bound engine that, like k_step, writes the selected record buffer inside its "launch".
class _LazyWork:
    def __init__(self, out, inp, world, log, tag):
        self.out, self.inp, self.world, self.log, self.tag, self.done = out, inp, world, log, tag, False

    def wait(self):
        if not self.done:
            self.log.append(("wait", self.tag))
            self.out.copy_(self.inp.repeat(self.world, 1))        # reads the record buffer NOW
            self.done = True


class _LazyDist:
    def __init__(self, world, log):
        self.world, self.log, self.n = world, log, 0

    def get_backend(self):
        return "fake"

    def all_gather_into_tensor(self, out, inp, async_op=False):
        w = _LazyWork(out, inp, self.world, self.log, self.n)
        self.n += 1
        if not async_op:
            w.wait()
        return w


class _BoundEngine:
    """Stands for KManipEnvHip: bind_reward_done_record / select_reward_done_record / a step that writes the selected buffer."""

    def __init__(self, n, log):
        self.n, self.log, self.rec, self.sel, self.k = n, log, None, 0, 0

    def bind_reward_done_record(self, r0, r1):
        self.rec, self.sel = (r0, r1), 0

    def select_reward_done_record(self, i):
        self.sel = i

    def step(self):
        import torch
        self.log.append(("write", self.sel, self.k))
        self.rec[self.sel][:, 0] = torch.arange(self.n, dtype=torch.float64) + 1000.0 * self.k      # reward of step k
        self.rec[self.sel][:, 1] = float(self.k % 2)
        self.k += 1


def test_bound_record_is_not_overwritten_before_its_gather_completed():
    import torch
    n, world, steps = 6, 2, 9
    log = []
    g = RewardDoneGather(n, world, torch.device("cpu"), _LazyDist(world, log))
    eng = _BoundEngine(n, log)
    g.bind(eng)
    got = []
    prev = None
    for k in range(steps):                      # bench.py's loop: pipelined, nobody consumes a result before the next step
        g.before_step()
        eng.step()
        b = g.post()
        if prev is not None and k % 3 == 0:     # (a learner that only looks now and then)
            got.append((k - 1, g.result(prev)[0].clone()))
        prev = b
    g.wait()
    # every collective read its record before the step that reuses the buffer wrote it: wait(tag j) precedes write(., j + 2)
    pos = {e: i for i, e in enumerate(log)}
    for j in range(steps - 2):
        assert pos[("wait", j)] < pos[("write", j & 1, j + 2)], (j, log)
    for k, r in got:                            # and what was gathered for step k IS step k's reward
        assert torch.equal(r, (torch.arange(n, dtype=torch.float64) + 1000.0 * k).repeat(world)), k
    # the final buffers hold the last two steps
    assert torch.equal(g.result((steps - 1) & 1)[0][:n], torch.arange(n, dtype=torch.float64) + 1000.0 * (steps - 1))
    assert torch.equal(g.result((steps - 2) & 1)[0][:n], torch.arange(n, dtype=torch.float64) + 1000.0 * (steps - 2))


@pytest.mark.parametrize("depth", [3, 8])
def test_deeper_ring_lets_a_rank_run_ahead_but_never_past_its_own_records(depth):
    """RewardDoneGather(depth=D): step k writes rec[k % D], which the exchange of step k - D read -- that one, and no later one,
    has to be complete before the launch; D - 1 exchanges stay in flight (a rank may run that far ahead of a crawling peer)."""
    import torch
    n, world, steps = 5, 2, 3 * depth + 2
    log = []
    g = RewardDoneGather(n, world, torch.device("cpu"), _LazyDist(world, log), depth=depth)
    eng = _BoundEngine(n, log)
    g.bind(eng)
    for k in range(steps):
        b = g.before_step()
        assert b == k % depth
        eng.step()
        assert g.post() == b
        assert sum(w is not None for w in g.pending) == min(k + 1, depth)      # nothing was waited for early
    writes = {e[2]: i for i, e in enumerate(log) if e[0] == "write"}
    waits = {e[1]: i for i, e in enumerate(log) if e[0] == "wait"}
    for j in range(steps - depth):
        assert waits[j] < writes[j + depth], (j, log)                           # the invariant
        assert waits[j] > writes[j + depth - 1], (j, log)                       # and not a step earlier than it must
    for k in range(steps - depth, steps):                                       # the ring holds the last D steps
        assert torch.equal(g.result(k % depth)[0], (torch.arange(n, dtype=torch.float64) + 1000.0 * k).repeat(world)), k


def test_a_step_without_an_exchange_keeps_the_two_sides_in_phase():
    """the engine used to flip its own buffer parity on every step, so one evaluation step without a post() made
    every later post() gather the stale buffer.  The engine keeps no counter now: before_step() selects the buffer."""
    import torch
    n, world = 4, 2
    log = []
    g = RewardDoneGather(n, world, torch.device("cpu"), _LazyDist(world, log))
    eng = _BoundEngine(n, log)
    g.bind(eng)
    for k in range(5):
        if k == 2:
            eng.step()                           # an extra step nobody exchanges (its record is overwritten by the next step)
        g.before_step()
        eng.step()
        b = g.post()
        r, _ = g.result(b)
        assert torch.equal(r[:n], torch.arange(n, dtype=torch.float64) + 1000.0 * (eng.k - 1)), k


def _block_worker(rank, world, port, n_total, steps, K, out_dir):
    import torch
    import torch.distributed as dist
    from gym_kmanip_amd.dist import BlockRewardDoneGather
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(n_total, world, rank)
    g = BlockRewardDoneGather(hi - lo, world, torch.device("cpu"), dist, block=K)
    rew_blocks, done_blocks = [], []
    for k in range(steps):
        # a stand-in engine: reward / done are functions of (global env id, step), so the gathered blocks are checkable
        rew = torch.arange(lo, hi, dtype=torch.float64) * 1000 + k
        done = torch.tensor([(e + k) % 5 == 0 for e in range(lo, hi)], dtype=torch.float64)
        b = g.post(rew, done)
        assert (b is not None) == (k % K == K - 1)
        if b is not None:
            r, d = g.result(b)
            rew_blocks.append(r.numpy().copy()); done_blocks.append(d.numpy().copy())
    if rank == 1:
        np.savez(os.path.join(out_dir, "blocks.npz"), rew=np.array(rew_blocks), done=np.array(done_blocks))
    dist.barrier()
    dist.destroy_process_group()


def test_block_gather_two_ranks(tmp_path):
    """dist.BlockRewardDoneGather: K steps per exchange, both ranks see every rank's records in global env order."""
    import torch.multiprocessing as mp
    n_total, steps, K, world = 10, 12, 3, 2
    mp.spawn(_block_worker, args=(world, _free_port(), n_total, steps, K, str(tmp_path)), nprocs=world, join=True)
    got = np.load(os.path.join(str(tmp_path), "blocks.npz"))
    assert got["rew"].shape == (steps // K, K, n_total)
    for blk in range(steps // K):
        for j in range(K):
            k = blk * K + j
            assert np.array_equal(got["rew"][blk, j], np.arange(n_total) * 1000.0 + k)
            assert np.array_equal(got["done"][blk, j], np.array([(e + k) % 5 == 0 for e in range(n_total)], dtype=np.uint8))


def _share_worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    from gym_kmanip_amd.dist import share_bytes
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    raw = bytes((7 * i) % 256 if i % 5 else 0 for i in range(128))       # NUL bytes inside, like a real ncclUniqueId
    got = share_bytes(raw if rank == 0 else None, 128, dist, torch)
    with open(os.path.join(out_dir, "id%d.bin" % rank), "wb") as f:
        f.write(got)
    dist.barrier()
    dist.destroy_process_group()


def test_unique_id_bytes_reach_every_rank(tmp_path):
    """dist.share_bytes: what carries rank 0's 128-byte ncclUniqueId to the other ranks before dist.RcclDirect's
    ncclCommInitRank -- every rank ends up with rank 0's bytes, embedded NULs included."""
    import torch.multiprocessing as mp
    mp.spawn(_share_worker, args=(3, _free_port(), str(tmp_path)), nprocs=3, join=True)
    want = bytes((7 * i) % 256 if i % 5 else 0 for i in range(128))
    for r in range(3):
        assert open(os.path.join(str(tmp_path), "id%d.bin" % r), "rb").read() == want


def test_direct_exchange_is_a_device_path_only():
    """direct=... asks for RCCL on device records: without anything to exchange it is inert; on host records it refuses."""
    import torch
    from gym_kmanip_amd.dist import BlockRewardDoneGather, RewardDoneGather
    g = RewardDoneGather(4, 1, "cpu", None, direct="side")
    assert g.direct is None and g.side is None
    g.post(torch.arange(4.0, dtype=torch.float64), torch.zeros(4, dtype=torch.float64))
    assert torch.equal(g.result(0)[0], torch.arange(4.0, dtype=torch.float64))
    for cls in (RewardDoneGather, BlockRewardDoneGather):
        with pytest.raises(RuntimeError, match="device path"):
            cls(4, 2, "cpu", None, direct="stream")


def _agree_worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    from gym_kmanip_amd.dist import RcclDirect, _all_agree
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # the local half of the direct communicator: rank 1 cannot bind its library (a path that does not exist)
    try:
        RcclDirect.bind_library(world, rank, lib_path=None if rank != 1 else "/nonexistent/librccl.so")
        mine = True
    except OSError:
        mine = False
    verdicts = [_all_agree(mine, dist, world, torch, "cpu"), _all_agree(True, dist, world, torch, "cpu")]
    with open(os.path.join(out_dir, "agree%d.txt" % rank), "w") as f:
        f.write("%s %s %s" % (mine, verdicts[0], verdicts[1]))
    dist.barrier()
    dist.destroy_process_group()


def test_ranks_agree_before_the_direct_communicator_is_made(tmp_path):
    """dist._make_direct's lock-step: a rank that cannot bind librccl.so still takes part in the agreement, every
    rank learns that ONE of them failed (so all raise together, before ncclCommInitRank could leave the others waiting), and
    the next agreement is unaffected."""
    import torch.multiprocessing as mp
    mp.spawn(_agree_worker, args=(3, _free_port(), str(tmp_path)), nprocs=3, join=True)
    got = [open(os.path.join(str(tmp_path), "agree%d.txt" % r)).read().split() for r in range(3)]
    assert [g[0] for g in got] == ["True", "False", "True"]          # only rank 1 failed locally ...
    assert all(g[1] == "False" and g[2] == "True" for g in got)      # ... and every rank knows
