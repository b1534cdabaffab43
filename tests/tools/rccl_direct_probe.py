"""Diagnostic (GPU box): what does the per-step (reward, done) exchange cost when the collective is issued DIRECTLY to RCCL on the
step's own stream (ncclAllGather through ctypes on torch's librccl.so) instead of through torch.distributed's wrapper, which puts it
on a stream of its own and synchronises the two with events?  One rank on cuda:0.
   python tests/tools/rccl_direct_probe.py [steps]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from gym_kmanip_amd import env_hip
from gym_kmanip_amd.dist import RcclDirect

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 512
timing = len(sys.argv) > 2 and sys.argv[2] == "timing"      # the library's own event pair around k_step, as bench.py runs it
n = 4096
e = env_hip.make("KManipSoloArm", num_envs=n, seed=0)
e.k_reset()
import numpy as np
e.set_state(step=(np.arange(n) % 64).astype(np.int32))
for k in range(70):
    e.step_flat(e.sample_action())
acts = [e.sample_action(ahead=k).clone() for k in range(8)]
rec = [torch.zeros((n, 2), dtype=torch.float64, device="cuda") for _ in range(2)]
out = [torch.zeros((n, 2), dtype=torch.float64, device="cuda") for _ in range(2)]
e.bind_reward_done_record(rec[0], rec[1])
if timing:
    e.enable_timing(True)


def run(gather):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        b = k & 1
        e.select_reward_done_record(b)
        e.step_flat(acts[k & 7])
        if gather is not None:
            gather(rec[b], out[b])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


base = run(None); base = run(None)
comm = RcclDirect(world=1, rank=0)
direct = lambda s, r: comm.all_gather(s, r, torch.cuda.current_stream())
d1 = run(direct); d1 = run(direct)
assert torch.equal(out[0], rec[0]) and torch.equal(out[1], rec[1])
# (b) the same direct call on a side stream of our own, tied to the step's stream the way torch.distributed ties its own: an event
# after the step that the side stream waits for, an event after the exchange that the step two later waits for
side = torch.cuda.Stream()
ev_step = [torch.cuda.Event() for _ in range(2)]
ev_done = [torch.cuda.Event() for _ in range(2)]
cnt = [0]
def side_gather(s, r):
    b = cnt[0] & 1; cnt[0] += 1
    cur = torch.cuda.current_stream()
    ev_step[b].record(cur)
    side.wait_event(ev_step[b])
    comm.all_gather(s, r, side)
    ev_done[b].record(side)
def run_side():
    torch.cuda.synchronize()
    cnt[0] = 0
    t0 = time.perf_counter()
    for k in range(steps):
        b = k & 1
        if k >= 2:
            torch.cuda.current_stream().wait_event(ev_done[b])
        e.select_reward_done_record(b)
        e.step_flat(acts[k & 7])
        side_gather(rec[b], out[b])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
d2 = run_side(); d2 = run_side()
assert torch.equal(out[0], rec[0]) and torch.equal(out[1], rec[1])
# (c) events only, no exchange: what the cross-stream bookkeeping alone costs
def ev_only(s, r):
    b = cnt[0] & 1; cnt[0] += 1
    ev_step[b].record(torch.cuda.current_stream())
d3 = run(ev_only); d3 = run(ev_only)
# (d) torch.distributed's wrapper on a one-rank RCCL group, for the same box
import torch.distributed as dist
from gym_kmanip_amd.dist import RewardDoneGather
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
g = RewardDoneGather(n, 1, torch.device("cuda", 0), dist, force_collective=True)
g.bind(e)
def run_torch():
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        g.before_step(); e.step_flat(acts[k & 7]); g.post()
    g.wait(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
d4 = run_torch(); d4 = run_torch()
# a second round of every variant, interleaved, so that drift of the box shows
r2 = (run(None), run(direct), run_side(), run_torch())
print("library timing events %s; second round (none, direct, side, torch): %s" % ("ON" if timing else "off", " ".join("%.4f" % x for x in r2)))
print("  same call on a side stream (event after the step, wait two steps later) %.4f ms/step (+%.1f us); one event record per step, no exchange %.4f (+%.1f us); torch.distributed all_gather_into_tensor(async_op=True) %.4f (+%.1f us)" % (
    d2, (d2 - base) * 1e3, d3, (d3 - base) * 1e3, d4, (d4 - base) * 1e3))
print("KManipSoloArm @ %d envs, %d steps, one rank: no exchange %.4f ms/step; ncclAllGather on the step's stream %.4f ms/step (+%.1f us)" % (
    n, steps, base, d1, (d1 - base) * 1e3))
comm.destroy()
