"""Multi-GPU plumbing: env-index sharding + the per-step (reward, done) all-gather.

Envs share nothing but the read-only model (the reference holds one Physics per env instance,
env_sim.py:206-211), so the hot path shards by env index with NO data-path collective.  The only exchange
the north star names is the tiny per-step reward/done all-gather for a single learner process: one packed
[reward f64 | done f64] record per env, issued asynchronously (RCCL over xGMI on GPUs, gloo in the CPU
tests) and double-buffered so that it overlaps the next control step.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple


def shard_range(total_envs: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous env-index block [lo, hi) of `rank`; sizes differ by at most one."""
    base, rem = divmod(total_envs, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class RewardDoneGather:
    """Asynchronous all-gather of (reward, done) across ranks (equal shard sizes) through a ring of `depth` buffers.

    depth = 2 (the default) double-buffers: step k's exchange runs beside step k + 1, and step k + 2 waits for it.  An
    all-gather is a rendezvous, and a rank's step lasts as long as its slowest wave -- an IK problem crawling at the reference's
    evaluation limit turns a 0.6 ms step into a 1.5-4 ms one about once in forty -- so with two buffers every rank soon waits for
    whichever rank crawled last (tools/scaling_model.py on measured launch times: 76 % of linear at 8 ranks).  A deeper ring lets
    a rank run up to `depth` steps ahead of the slowest one before it has to wait, so that the ranks' outliers average out
    instead of adding up (the model's K column); the exchanges queue up, in order, on the side stream."""

    def __init__(self, n_local: int, world: int, device, dist=None, force_collective: bool = False, overlap: bool = True,
                 direct: bool = False, depth: int = 2):
        """force_collective: issue the real all_gather_into_tensor(async_op=True) even when world == 1 (which otherwise
        short-circuits to a device copy) -- a one-GPU box can then execute the RCCL device-collective path the N > 1 job
        will run (tests/test_multi_rank_gpu.py::test_rccl_device_collective_world1).
        direct: ncclAllGather through RcclDirect (below) instead of torch.distributed's wrapper, which costs the step's stream
        about 30 us per exchange on one MI355X (profiles/r05_rccl_direct.txt).  "stream": on the STEP's stream -- ordered after
        its step and before the next one by the stream itself, no second stream, no events (+2 us on one GPU; across GPUs the
        exchange's own latency is then paid every step).  "side" (or True): on a side stream of the gather's own, tied to the
        step's stream by one event after the step and one wait two steps later, so that the exchange runs beside the next step
        like the wrapper's does.  Needs a "nccl" process group (or world == 1) for the 128-byte id broadcast; `overlap` has no
        meaning with it."""
        import torch
        self.torch = torch
        self.direct = _make_direct(direct, world, dist, device, force_collective)
        self.side = _SideStream(torch, device, depth) if (self.direct is not None and direct != "stream") else None
        # overlap=True: step k's exchange runs beside step k + 1 (the collective kernel then competes with k_step, whose waves
        # fill every SIMD slot, for compute units); overlap=False: the step's stream waits for its own exchange before the next
        # step is enqueued (no competition; the exchange's latency is paid every step).  Measured on one GPU (bench.py --rccl-world1
        # [--gather-serial] [--rccl-one-channel]): profiles/r05_rccl_world1.txt.
        self.overlap = overlap
        self.dist = dist
        self.world = world
        self.n = n_local
        assert depth >= 2
        self.depth = depth
        self.rec = [torch.zeros((n_local, 2), dtype=torch.float64, device=device) for _ in range(depth)]
        self.all = [torch.zeros((world * n_local, 2), dtype=torch.float64, device=device) for _ in range(depth)]
        self.pending = [None] * depth
        self.k = 0                # THE step counter: step k uses buffer k % depth (the engine keeps none of its own)
        self.env = None           # bound engine (bind): it writes rec[k % depth] inside its step launch
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
        env.bind_reward_done_record(self.rec[0], self.rec[1])      # (a ring deeper than two re-binds the record before every step)
        self.env = env

    def _retire(self, b):
        # Work.wait() makes the CURRENT stream wait for the collective (RCCL) or blocks the host (gloo): either way everything
        # enqueued on this stream afterwards runs after the collective has read rec[b] and written all[b]
        if self.pending[b] is not None:
            self.pending[b].wait()
            self.pending[b] = None

    def before_step(self):
        """Call BEFORE enqueuing step k.  ORDERING INVARIANT: the all-gather of step k-depth reads rec[k % depth] and writes
        all[k % depth]; a bound engine's step k writes rec[k % depth] inside its launch, so the step's stream has to wait for
        that collective before the launch is enqueued (a peer that lags `depth` steps -- one IK crawl is seven -- would otherwise
        let this rank ship step k's rewards as step k-depth's).  Also points the engine at that buffer: one counter, here."""
        b = self.k % self.depth
        self._retire(b)
        if self.env is not None:
            if self.depth == 2:
                self.env.select_reward_done_record(b)
            else:
                self.env.bind_reward_done_record(self.rec[b], self.rec[b])
        self._armed = self.k
        return b

    def post(self, reward=None, done=None):
        """Start the collective on step k's local results (packed here unless the engine is bound); returns the buffer index."""
        b = self.k % self.depth
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
        elif self.direct is not None:
            self.pending[b] = _direct_gather(self, b)
        elif self.dist is not None and (self.world > 1 or self.force_collective):
            self.pending[b] = self.dist.all_gather_into_tensor(self.all[b], self.rec[b], async_op=True)
            if not self.overlap:
                self._retire(b)       # Work.wait(): the current stream waits for the collective (the host does not block)
        else:
            self.all[b].copy_(self.rec[b])
        return b

    def wait(self, b: Optional[int] = None):
        for i in ([b] if b is not None else range(self.depth)):
            self._retire(i)

    def close(self):
        """Wait for what is in flight and give the direct communicator (if any) back to RCCL."""
        self.wait()
        if self.direct is not None:
            self.torch.cuda.synchronize()
            self.direct.destroy()
            self.direct = None

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

    def __init__(self, n_local: int, world: int, device, dist=None, block: int = 8, force_collective: bool = False,
                 direct: bool = False):
        import torch
        assert block >= 1
        self.direct = _make_direct(direct, world, dist, device, force_collective)
        self.side = _SideStream(torch, device) if (self.direct is not None and direct != "stream") else None
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
        elif self.direct is not None:
            self.pending[b] = _direct_gather(self, b)
        elif self.dist is not None and (self.world > 1 or self.force_collective):
            self.pending[b] = self.dist.all_gather_into_tensor(self.all[b], self.rec[b], async_op=True)
        else:
            self.all[b].copy_(self.rec[b])
        return b

    def wait(self, b: Optional[int] = None):
        for i in ([b] if b is not None else [0, 1]):
            self._retire(i)

    def close(self):
        """Wait for what is in flight and give the direct communicator (if any) back to RCCL."""
        self.wait()
        if self.direct is not None:
            self.torch.cuda.synchronize()
            self.direct.destroy()
            self.direct = None

    def result(self, b: int):
        """(reward [K, world*n], done [K, world*n] uint8) of block b, global env order."""
        self.wait(b)
        a = self.all[b].view(self.world, self.K, self.n, 2).permute(1, 0, 2, 3).reshape(self.K, self.world * self.n, 2)
        return a[:, :, 0], a[:, :, 1].to(self.torch.uint8)


class _SideStream:
    """The gather's own stream for direct exchanges, with the two events per buffer that tie it to the step's stream."""

    def __init__(self, torch, device, depth=2):
        self.stream = torch.cuda.Stream(device=device)
        self.after_step = [torch.cuda.Event() for _ in range(depth)]
        self.after_gather = [torch.cuda.Event() for _ in range(depth)]


class _SideWork:
    """What `pending[b]` holds for a direct exchange on the side stream: wait() makes the current stream wait for it (the host
    does not block) -- the contract of torch.distributed's Work.wait() on a device collective."""

    def __init__(self, torch, event):
        self.torch, self.event = torch, event

    def wait(self):
        self.torch.cuda.current_stream().wait_event(self.event)


def _direct_gather(g, b):
    cur = g.torch.cuda.current_stream()
    if g.side is None:
        g.direct.all_gather(g.rec[b], g.all[b], cur)       # stream order is the only ordering there is: nothing to wait for later
        return None
    sd = g.side
    sd.after_step[b].record(cur)
    sd.stream.wait_event(sd.after_step[b])
    g.direct.all_gather(g.rec[b], g.all[b], sd.stream)
    sd.after_gather[b].record(sd.stream)
    return _SideWork(g.torch, sd.after_gather[b])


def _all_agree(ok: bool, dist, world: int, torch, device) -> bool:
    """True iff `ok` on EVERY rank (an all-reduce MIN through the existing process group; world 1: `ok`).  Every rank takes
    part whatever its own `ok` is, so a rank whose local set-up failed never leaves its peers alone in the next collective."""
    if world <= 1 or dist is None:
        return bool(ok)
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()))


def _make_direct(direct, world, dist, device, force_collective):
    """The RcclDirect communicator of a gather (None unless asked for and there is something to exchange).

    The ranks move in lock-step: (1) each binds librccl.so and its symbols locally -- nothing that can wait for a peer --, then
    all agree (all-reduce MIN) that every rank could; (2) only then the id broadcast and ncclCommInitRank, which need every
    rank; (3) the self-test's verdict is agreed the same way, and a communicator that failed it on ANY rank is destroyed on
    every rank.  So a failure raises on all ranks together and no rank is left waiting in a collective its peers never issue."""
    if not direct or not (world > 1 or force_collective):
        return None
    import torch
    if torch.device(device).type != "cuda":
        raise RuntimeError("direct=... is the RCCL device path: the records must live on a GPU")
    if world > 1 and (dist is None or dist.get_backend() != "nccl"):
        raise RuntimeError("direct=True needs the ranks on distinct GPUs (RCCL refuses two ranks on one device): backend 'nccl'")
    rank = dist.get_rank() if (dist is not None and dist.is_initialized()) else 0
    with torch.cuda.device(device):
        err = None
        try:
            comm = RcclDirect.bind_library(world, rank)
        except Exception as ex:  # noqa: BLE001   (library not where torch keeps it, a symbol missing)
            comm, err = None, ex
        if not _all_agree(comm is not None, dist, world, torch, device):
            raise RuntimeError("RcclDirect: librccl.so could not be bound on %s (%s)" % ("rank %d" % rank if err else "a peer rank", err))
        comm.init_rank(dist)
        ok = comm.self_test(rank, device)
        if not _all_agree(ok, dist, world, torch, device):
            torch.cuda.synchronize()
            comm.destroy()
            raise RuntimeError("RcclDirect: the test all-gather of the rank numbers came back wrong on %s"
                               % ("rank %d" % rank if not ok else "a peer rank"))
        return comm


def share_bytes(raw, nbytes: int, dist, torch) -> bytes:
    """Rank 0's `raw` (nbytes long) on every rank, through the process group `dist` (a device tensor on "nccl", host on gloo)."""
    t = torch.zeros(nbytes, dtype=torch.uint8)
    if raw is not None:
        assert len(raw) == nbytes
        t.copy_(torch.frombuffer(bytearray(raw), dtype=torch.uint8))
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.broadcast(t, src=0)
    return bytes(t.cpu().numpy().tobytes())


class RcclDirect:
    """RCCL's C API bound directly (ctypes on the librccl.so torch ships): `ncclAllGather` on a stream of the CALLER's choice.

    torch.distributed issues a collective on a stream of its own and ties it to the caller's stream with an event record and a
    stream wait on either side; on one MI355X that costs the step's stream about 30 us per exchange (profiles/r05_rccl_world1.txt)
    -- around a 64 KB all-gather whose kernel runs for a few microseconds.  Put on the step's own stream the exchange needs no
    cross-stream synchronisation at all: it simply runs after k_step, before the next one (2 us); on a side stream of the
    caller's, with one event each way, 14 us (profiles/r05_rccl_direct.txt).
    The communicator is created from an ncclUniqueId that rank 0 generates and the other ranks receive through the existing
    torch.distributed process group (any backend: 128 bytes, once)."""

    NCCL_FLOAT64 = 8

    class _UniqueId(C.Structure):           # rccl.h: typedef struct { char internal[NCCL_UNIQUE_ID_BYTES]; } ncclUniqueId, 128 bytes
        _fields_ = [("internal", C.c_ubyte * 128)]

    def __init__(self, world: int, rank: int, dist=None, lib_path: Optional[str] = None):
        """Bind the library and create the communicator (the two halves below, for callers that need no agreement between them)."""
        self._bind(world, rank, lib_path)
        self.init_rank(dist)

    @classmethod
    def bind_library(cls, world: int, rank: int, lib_path: Optional[str] = None) -> "RcclDirect":
        """Local half only: dlopen + symbol lookups.  Cannot wait for a peer; raises OSError / AttributeError when it cannot be done."""
        self = cls.__new__(cls)
        self._bind(world, rank, lib_path)
        return self

    def _bind(self, world, rank, lib_path=None):
        import os
        import torch
        path = lib_path or os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        self.torch = torch
        self.comm = None
        self.world, self.rank = world, rank
        self.L = L = C.CDLL(path)
        L.ncclGetUniqueId.argtypes = [C.POINTER(self._UniqueId)]
        L.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, self._UniqueId, C.c_int]
        L.ncclAllGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
        L.ncclCommDestroy.argtypes = [C.c_void_p]
        L.ncclGetErrorString.restype = C.c_char_p

    def init_rank(self, dist=None):
        """Collective half: rank 0's ncclUniqueId to every rank through `dist`, then ncclCommInitRank (a rendezvous of all ranks)."""
        L, world, rank, torch = self.L, self.world, self.rank, self.torch
        uid = self._UniqueId()
        if rank == 0:
            self._chk(L.ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
        if world > 1:
            assert dist is not None, "the unique id travels through an existing torch.distributed process group"
            # (string_at: a c_char array FIELD reads back truncated at its first NUL byte)
            raw = share_bytes(C.string_at(C.byref(uid), 128) if rank == 0 else None, 128, dist, torch)
            C.memmove(C.byref(uid), raw, 128)
        comm = C.c_void_p()
        self._chk(L.ncclCommInitRank(C.byref(comm), world, uid, rank), "ncclCommInitRank")
        self.comm = comm

    def _chk(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed: %s" % (what, self.L.ncclGetErrorString(rc).decode()))

    def all_gather(self, send, recv, stream):
        """recv[world * n, ...] <- every rank's send[n, ...] (float64, contiguous, on the device), enqueued on `stream`."""
        assert send.dtype == self.torch.float64 and send.is_contiguous() and recv.is_contiguous() and recv.numel() == self.world * send.numel()
        self._chk(self.L.ncclAllGather(C.c_void_p(send.data_ptr()), C.c_void_p(recv.data_ptr()), send.numel(), self.NCCL_FLOAT64,
                                       self.comm, C.c_void_p(stream.cuda_stream)), "ncclAllGather")

    def self_test(self, rank: int, device) -> bool:
        """One all-gather of the rank numbers on the current stream: True iff every rank's block arrived in rank order."""
        torch = self.torch
        send = torch.full((8,), float(rank), dtype=torch.float64, device=device)
        recv = torch.full((8 * self.world,), -1.0, dtype=torch.float64, device=device)
        self.all_gather(send, recv, torch.cuda.current_stream(device))
        torch.cuda.current_stream(device).synchronize()
        want = torch.arange(self.world, dtype=torch.float64, device=device).repeat_interleave(8)
        return bool(torch.equal(recv, want))

    def destroy(self):
        """Give the communicator back (after the streams it was used on have drained: the gathers' close() synchronises)."""
        if self.comm:
            self.L.ncclCommDestroy(self.comm)
            self.comm = None
