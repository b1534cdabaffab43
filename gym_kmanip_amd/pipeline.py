"""Several independent env batches kept in flight on one GPU (the deployment shape that uses the chip).

One batch's control step is ONE launch that ends with its slowest wave: at 4096 single-arm envs the mean wave runs 0.72 M cycles
and the launch 1.5 M, so over a launch the SIMDs sit idle half the time.  The reference steps one env per call
(`KManipEnv.step`, env_base.py:241-259) and has no vector env at all; a batched caller that alternates K >= 2 batches around its
policy -- batch A steps while the policy works on batch B's observations -- gets the idle half back WITHOUT giving up the per-step
boundary: every batch is a handle of its own (`kmanip_create` with `env_id_offset` = its first global env id, so the K batches are
exactly the envs of one K x n batch: RNG streams are keyed by the global id) on a stream of its own; the second batch's waves take
the SIMD slots the first batch's early finishers free, and a batch's next step starts when ITS slowest wave ends.
Measured (profiles/r04_multi_handle_timing.txt): 2 x 4096 single-arm envs 10.9 M env steps/s against 6.5 M for one batch; DualArm
2 x 4096: 4.9 M against 3.9 M for one 8192-env handle.
"""
from __future__ import annotations

import time
from typing import List, Optional

from . import env_hip
from .model import compile_model


def _torch():
    import torch
    return torch


class InterleavedBatches:
    """K handles of `envs_per_batch` envs each, one stream per handle.

        batches = InterleavedBatches("KManipSoloArm", 4096, k=2)
        for i in range(batches.k):
            with batches.on(i):
                batches.env[i].k_reset()
        for t in range(T):
            for i in range(batches.k):
                with batches.on(i):                         # torch stream context of batch i
                    act = policy(batches.env[i].obs)        # the caller's kernels, on the same stream
                    batches.env[i].step_flat(act)           # returns at once; nothing synchronises

    Batch i holds the global envs [i * n, (i + 1) * n) of seed `seed`: its trajectories are bit for bit those of the same envs in
    one K * n-env handle (tests/test_gpu_config_sizes.py)."""

    def __init__(self, env_id: str, envs_per_batch: int, k: int = 2, device: int = 0, seed: int = 0, env_id_offset: int = 0,
                 calibrate: bool = True, **compile_kw):
        torch = _torch()
        assert k >= 1
        self.k, self.n = k, envs_per_batch
        self.cm = compile_model(env_id, **compile_kw)
        self.env: List[env_hip.KManipEnvHip] = [
            env_hip.KManipEnvHip(self.cm, num_envs=envs_per_batch, device=device, seed=seed, env_id_offset=env_id_offset + i * envs_per_batch)
            for i in range(k)]
        with torch.cuda.device(device):
            self.stream = [torch.cuda.Stream() for _ in range(k)]
        self.device = device
        if calibrate and k > 1:
            self._separate_queues()
        # the handles were created, their output tensors zero-filled and sim_time bound on the default stream; the batch
        # streams are non-blocking, so nothing the caller enqueues on them may start before that is complete -- on every path
        torch.cuda.synchronize(device)

    def on(self, i: int):
        """torch stream context of batch i: everything enqueued inside runs on that batch's stream."""
        return _torch().cuda.stream(self.stream[i])

    def synchronize(self, i: Optional[int] = None):
        for s in (self.stream if i is None else [self.stream[i]]):
            s.synchronize()

    def close(self):
        for e in self.env:
            e.k_close()

    # Two streams of torch's pool can share a hardware queue; their kernels then run one after the other and two batches take
    # twice one batch's time.  For every stream after the first: try a few pool streams on a short burst of real control steps
    # (zero actions) and keep the candidate that overlaps best.  The probe's trace is removed: the state (qpos, qvel, ctrl, warm
    # start, step and episode counters) is check-pointed before and restored after, and the outputs the probe's steps wrote are
    # recomputed from the restored state (observe(): obs, reward and the contact masks get_diag / the first cost sort read) or
    # zeroed (done, sim_time as a never-stepped handle has them).  NOT restored: the IK diagnostics (ik_nfev / ik_status of the
    # probe's last step stay until the caller's first step overwrites them; the first cost sort of a two-arm handle reads them
    # as a predictor -- an order, never a result).
    def _separate_queues(self, tries: int = 3, reps: int = 4):
        torch = _torch()
        fresh = [int(e.get_episode().max()) < 0 for e in self.env]       # never reset: the probe needs a state to step from
        for e, f in zip(self.env, fresh):
            if f:
                e.k_reset()
        ck = [e.checkpoint() for e in self.env]
        zero = torch.zeros((self.n, self.cm.act_dim), dtype=torch.float32, device=self.env[0].device)
        torch.cuda.synchronize(self.device)

        def burst(streams):
            for _ in range(reps):
                for e, s in zip(self.env, streams):
                    with torch.cuda.stream(s):
                        e.step_flat(zero)
            torch.cuda.synchronize(self.device)

        for i in range(1, self.k):
            best = None
            for _ in range(tries):
                with torch.cuda.device(self.device):
                    cand = torch.cuda.Stream()
                trial = self.stream[:i] + [cand]
                burst(trial)                                     # warm
                t0 = time.perf_counter()
                burst(trial)
                dt = time.perf_counter() - t0
                if best is None or dt < best[0]:
                    best = (dt, cand)
            self.stream[i] = best[1]
        for e, c, f in zip(self.env, ck, fresh):
            e.restore(c)
            if f:
                e.set_episode([-1] * self.n)                     # as created: the caller's first k_reset is episode 0
            e.observe()
            e.done.zero_()
            if f:
                e.obs.zero_(); e.reward.zero_()                  # a handle that was never reset has no observation yet
            # sim_time = steps since the env's last reset x control_timestep (kmanip_bind_sim_time), from the restored counters
            e.sim_time.copy_(torch.from_numpy(c[4].astype("float64") * (self.cm.desc.n_sub_steps * self.cm.desc.timestep)))
        torch.cuda.synchronize(self.device)


class RenderBehind:
    """Camera images rendered BEHIND the steps, on a stream of their own (include/kmanip.h: kmanip_snapshot_render_state).

    For a loop whose policy does not look at the images -- the reference's scripted data generation (examples/2_synthetic_data.py:
    28-41 acts on the state and LOGS the cameras) or an open-loop action chunk.  After step k the step's stream copies qpos (all a
    render reads) into snapshot k & 1; the render stream waits for that copy and renders the snapshot into image set k & 1 while
    step k + 1 already runs: one 2048-env single-arm step is 1024 single-wave workgroups that end between 0.45 and 1.0 of the launch,
    and the render's workgroups take the SIMDs they free.  KManipSoloArmVision @ 2048 envs: 0.96 ms per step in sequence, 0.67 ms
    this way (profiles/r05_render_behind.txt).  The images of step k are those a render right after step k would have produced,
    bit for bit (tests/test_check_env_gpu.py).

        rb = RenderBehind(env)                       # env: KManipEnvHip of a *Vision id
        for t in range(T):
            env.step_flat(policy_from_state(env.obs))
            rb.after_step()                          # snapshot + render of step t, behind
            if t: log(rb.images(t - 1))              # dict name -> uint8 [n, h, w, 3]; current stream ordered after that render

    CONTRACT of images(): it returns the LIVE ring buffers of that step, not copies -- two image sets exist, and the render of step
    k + 2 rewrites set k & 1.  The reads a consumer enqueues on the stream it called images() from, before the second following
    after_step(), are safe: that after_step() makes the render stream wait for every stream images(k) was called from before it
    rewrites the set (for the step's own stream the snapshot copy already orders it).  A dict kept ACROSS two after_step() calls, or
    read from a stream images() never saw, shows torn or newer images: clone what has to live longer.
    """

    def __init__(self, env, cams=None, depth=None):
        """cams: RGB cameras (default: the env id's camera observations; [] for none).  depth: (camera, height, width) of a float32
        depth image rendered with them (BASELINE config 5's 64 x 64 gripper image), under the key "depth"."""
        torch = _torch()
        self.torch, self.env, self.cams, self.depth = torch, env, cams, depth
        self.k = 0
        self.stream = torch.cuda.Stream(device=env.device)
        self.bufs = [self._render(None), self._render(None)]                      # two image sets (allocated by a first render each)
        self.copied = [torch.cuda.Event() for _ in range(2)]
        self.rendered = [torch.cuda.Event() for _ in range(2)]
        self._used = [False, False]
        self._readers = [set(), set()]            # streams images() handed set s to since its last render

    def _render(self, out):
        bufs = dict(self.env.render_cameras(self.cams, out=out)) if (self.cams is None or len(self.cams)) else {}
        if self.depth is not None:
            cam, h, w = self.depth
            bufs["depth"] = self.env.render_depth(cam, h, w, out=None if out is None else out["depth"])
        return bufs

    def after_step(self):
        """Call right after the step whose images are wanted, on the step's stream.  Returns the step's index."""
        torch, env = self.torch, self.env
        s = self.k & 1
        cur = torch.cuda.current_stream(env.device)
        if self._used[s]:
            cur.wait_event(self.rendered[s])          # the render of step k - 2 read snapshot s: it must be done before the copy
        env.snapshot_render_state(s)
        self.copied[s].record(cur)
        self.stream.wait_event(self.copied[s])
        for st in self._readers[s]:                   # consumers of set s on other streams: their reads come before the rewrite
            if st != cur:
                self.stream.wait_stream(st)
        self._readers[s].clear()
        env.set_render_source(s)
        try:
            with torch.cuda.stream(self.stream):
                self._render(self.bufs[s])
        finally:
            env.set_render_source(-1)
        self.rendered[s].record(self.stream)
        self._used[s] = True
        self.k += 1
        return self.k - 1

    def images(self, step: int):
        """Image set of `step` (one of the last two after_step calls): the live ring buffers (see the class docstring's CONTRACT);
        the current stream waits for the step's render, and the render that will rewrite the set waits for the current stream."""
        if not (self.k - 2 <= step < self.k) or step < 0:
            raise ValueError("RenderBehind keeps the images of the last two steps (asked for %d, at %d)" % (step, self.k))
        s = step & 1
        cur = self.torch.cuda.current_stream(self.env.device)
        cur.wait_event(self.rendered[s])
        self._readers[s].add(cur)
        return self.bufs[s]

    def synchronize(self):
        self.stream.synchronize()
