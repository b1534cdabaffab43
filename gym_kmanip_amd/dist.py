"""Multi-GPU plumbing: env-index sharding + the per-step (reward, done) all-gather.

Envs share nothing but the read-only model (the reference holds one Physics per env instance,
env_sim.py:206-211), so the hot path shards by env index with NO data-path collective.  The only exchange
the north star names is the tiny per-step reward/done all-gather for a single learner process: one packed
[reward f64 | done f64] record per env, issued asynchronously (RCCL over xGMI on GPUs, gloo in the CPU
tests) and double-buffered so that it overlaps the next control step.
"""
from __future__ import annotations

from typing import Optional, Tuple


def shard_range(total_envs: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous env-index block [lo, hi) of `rank`; sizes differ by at most one."""
    base, rem = divmod(total_envs, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class RewardDoneGather:
    """Asynchronous, double-buffered all-gather of (reward, done) across ranks (equal shard sizes)."""

    def __init__(self, n_local: int, world: int, device, dist=None, force_collective: bool = False, overlap: bool = True):
        """force_collective: issue the real all_gather_into_tensor(async_op=True) even when world == 1 (which otherwise
        short-circuits to a device copy) -- a one-GPU box can then execute the RCCL device-collective path the N > 1 job
        will run (tests/test_multi_rank_gpu.py::test_rccl_device_collective_world1)."""
        import torch
        self.torch = torch
        # overlap=True: step k's exchange runs beside step k + 1 (the collective kernel then competes with k_step, whose waves
        # fill every SIMD slot, for compute units); overlap=False: the step's stream waits for its own exchange before the next
        # step is enqueued (no competition; the exchange's latency is paid every step).  Measured on one GPU (bench.py --rccl-world1
        # [--gather-serial] [--rccl-one-channel]): profiles/r05_rccl_world1.txt.
        self.overlap = overlap
        self.dist = dist
        self.world = world
        self.n = n_local
        self.rec = [torch.zeros((n_local, 2), dtype=torch.float64, device=device) for _ in range(2)]
        self.all = [torch.zeros((world * n_local, 2), dtype=torch.float64, device=device) for _ in range(2)]
        self.pending = [None, None]
        self.k = 0                # THE step counter: step k uses buffer k & 1 (the engine keeps none of its own)
        self.env = None           # bound engine (bind): it writes rec[k & 1] inside its step launch
        self._armed = -1          # the step before_step() last prepared: post() of a bound engine insists on it
        self.force_collective = bool(force_collective and dist is not None)
        # gloo has no device collectives: when a one-GPU box rehearses the multi-rank path over gloo, the records are
        # staged through host memory (synchronously); RCCL (backend "nccl") gathers the device buffers directly
        self.host_stage = bool(dist is not None and world > 1 and dist.get_backend() == "gloo"
                               and torch.device(device).type == "cuda")
        if self.host_stage:
            self.h_rec = torch.zeros((n_local, 2), dtype=torch.float64)
            self.h_all = torch.zeros((world * n_local, 2), dtype=torch.float64)

    def bind(self, env):
        """Let the engine write the packed records itself (KManipEnvHip.bind_reward_done_record -> kmanip_bind_reward_done_record):
        post() then skips its two packing kernels.  From then on every step must be bracketed `before_step(); env.step...;
        post()` -- before_step() is what keeps the step from overwriting a record a collective is still reading."""
        env.bind_reward_done_record(self.rec[0], self.rec[1])
        self.env = env

    def _retire(self, b):
        # Work.wait() makes the CURRENT stream wait for the collective (RCCL) or blocks the host (gloo): either way everything
        # enqueued on this stream afterwards runs after the collective has read rec[b] and written all[b]
        if self.pending[b] is not None:
            self.pending[b].wait()
            self.pending[b] = None

    def before_step(self):
        """Call BEFORE enqueuing step k.  ORDERING INVARIANT: the all-gather of step k-2 reads rec[k & 1] and writes
        all[k & 1]; a bound engine's step k writes rec[k & 1] inside its launch, so the step's stream has to wait for that
        collective before the launch is enqueued (a peer that lags two steps -- one IK crawl is seven -- would otherwise let
        this rank ship step k's rewards as step k-2's).  Also points the engine at buffer k & 1: one counter, here."""
        b = self.k & 1
        self._retire(b)
        if self.env is not None:
            self.env.select_reward_done_record(b)
        self._armed = self.k
        return b

    def post(self, reward=None, done=None):
        """Start the collective on step k's local results (packed here unless the engine is bound); returns the buffer index."""
        b = self.k & 1
        if self.env is not None and self._armed != self.k:
            # a bound engine fills the buffer before_step() selected: without it this step wrote the OTHER buffer (or the one a
            # collective was still reading) and the gather below would ship a stale record every other step
            raise RuntimeError("RewardDoneGather.post(): before_step() was not called for step %d of a bound engine "
                               "(contract: before_step(); env.step...; post())" % self.k)
        self.k += 1
        self._retire(b)           # (no-op after before_step(); the un-bound path may skip before_step)
        if self.env is None:
            self.rec[b][:, 0].copy_(reward)
            self.rec[b][:, 1].copy_(done)
        if self.host_stage:
            self.h_rec.copy_(self.rec[b])
            self.dist.all_gather_into_tensor(self.h_all, self.h_rec)
            self.all[b].copy_(self.h_all)
        elif self.dist is not None and (self.world > 1 or self.force_collective):
            self.pending[b] = self.dist.all_gather_into_tensor(self.all[b], self.rec[b], async_op=True)
            if not self.overlap:
                self._retire(b)       # Work.wait(): the current stream waits for the collective (the host does not block)
        else:
            self.all[b].copy_(self.rec[b])
        return b

    def wait(self, b: Optional[int] = None):
        for i in ([b] if b is not None else [0, 1]):
            self._retire(i)

    def result(self, b: int):
        """(reward[world*n], done[world*n] uint8) of buffer b, global env order."""
        self.wait(b)
        return self.all[b][:, 0], self.all[b][:, 1].to(self.torch.uint8)


class BlockRewardDoneGather:
    """(reward, done) of K consecutive steps per all-gather ("also offer gather every K steps").

    On one MI355X the per-step exchange costs the step's stream about 30 us -- event records, stream waits and the collective
    kernel's launch around a 0.6 ms step (bench.py --rccl-world1: 6.81 -> 6.48 M env steps/s, profiles/r05_rccl_world1.txt) --
    whatever the channel count and whether or not it overlaps the next step.  A learner that consumes rewards in blocks (n-step
    returns, GAE over a rollout) can take them K steps at a time: the bound engine writes step j of a block straight into row j
    of a [K, n, 2] record block (re-bound every step: two pointers), and the block is gathered once, asynchronously, while the
    next block fills the other buffer.  Same ordering invariant as RewardDoneGather, per block."""

    def __init__(self, n_local: int, world: int, device, dist=None, block: int = 8, force_collective: bool = False):
        import torch
        assert block >= 1
        self.torch, self.dist, self.world, self.n, self.K = torch, dist, world, n_local, block
        self.rec = [torch.zeros((block, n_local, 2), dtype=torch.float64, device=device) for _ in range(2)]
        # (the output of all_gather_into_tensor is the concatenation of the ranks' inputs along dim 0: [world * K, n, 2])
        self.all = [torch.zeros((world * block, n_local, 2), dtype=torch.float64, device=device) for _ in range(2)]
        self.pending = [None, None]
        self.k = 0
        self.env = None
        self._armed = -1
        self.force_collective = bool(force_collective and dist is not None)
        self.host_stage = bool(dist is not None and world > 1 and dist.get_backend() == "gloo" and torch.device(device).type == "cuda")

    def bind(self, env):
        self.env = env

    def _retire(self, b):
        if self.pending[b] is not None:
            self.pending[b].wait()
            self.pending[b] = None

    def before_step(self):
        j, b = self.k % self.K, (self.k // self.K) & 1
        if j == 0:
            self._retire(b)            # the exchange of two blocks ago read rec[b]: it must be done before the block is refilled
        if self.env is not None:
            row = self.rec[b][j]
            self.env.bind_reward_done_record(row, row)
        self._armed = self.k
        return b

    def post(self, reward=None, done=None):
        """After step k.  Returns the block index when this step completed a block (its exchange is now in flight), else None."""
        j, b = self.k % self.K, (self.k // self.K) & 1
        if self.env is not None and self._armed != self.k:
            raise RuntimeError("BlockRewardDoneGather.post(): before_step() was not called for step %d of a bound engine" % self.k)
        if self.env is None:
            if j == 0:
                self._retire(b)
            self.rec[b][j, :, 0].copy_(reward)
            self.rec[b][j, :, 1].copy_(done)
        self.k += 1
        if j != self.K - 1:
            return None
        if self.host_stage:
            h_all = self.torch.zeros(self.all[b].shape, dtype=self.torch.float64)
            self.dist.all_gather_into_tensor(h_all, self.rec[b].cpu())
            self.all[b].copy_(h_all)
        elif self.dist is not None and (self.world > 1 or self.force_collective):
            self.pending[b] = self.dist.all_gather_into_tensor(self.all[b], self.rec[b], async_op=True)
        else:
            self.all[b].copy_(self.rec[b])
        return b

    def wait(self, b: Optional[int] = None):
        for i in ([b] if b is not None else [0, 1]):
            self._retire(i)

    def result(self, b: int):
        """(reward [K, world*n], done [K, world*n] uint8) of block b, global env order."""
        self.wait(b)
        a = self.all[b].view(self.world, self.K, self.n, 2).permute(1, 0, 2, 3).reshape(self.K, self.world * self.n, 2)
        return a[:, :, 0], a[:, :, 1].to(self.torch.uint8)
