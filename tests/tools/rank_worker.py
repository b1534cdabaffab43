#!/usr/bin/env python3
"""One rank of a multi-process HIP-engine run (started by tests/test_multi_rank_gpu.py with the torch.distributed
environment of its rank): steps its contiguous env shard on its GPU, all-gathers (reward, done) every step, and saves what
it saw.  `--same-device` puts every rank on cuda:0 and `--backend gloo` swaps RCCL for gloo, so that the whole multi-rank
code path can be rehearsed on a one-GPU box (RCCL itself refuses two ranks on one device)."""
import argparse
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--env", default="KManipSoloArm")
    ap.add_argument("--total", type=int, default=512)
    ap.add_argument("--steps", type=int, default=70)
    ap.add_argument("--seed", type=int, default=9)
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--same-device", action="store_true")
    ap.add_argument("--gather-every", type=int, default=1, help="K > 1: dist.BlockRewardDoneGather, K steps per exchange")
    ap.add_argument("--force-collective", action="store_true",
                    help="issue the real all_gather_into_tensor(async_op=True) even at world_size 1 (dist.RewardDoneGather)")
    ap.add_argument("--direct", choices=["off", "stream", "side"], default="off",
                    help="ncclAllGather through dist.RcclDirect on the step's stream / on the gather's side stream")
    ap.add_argument("--depth", type=int, default=2, help="dist.RewardDoneGather(depth=...): ring of record buffers")
    a = ap.parse_args()
    direct = a.direct if a.direct != "off" else False
    import numpy as np
    import torch
    import torch.distributed as dist
    from gym_kmanip_amd import env_hip
    from gym_kmanip_amd.dist import BlockRewardDoneGather, RewardDoneGather, shard_range
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = 0 if a.same_device else int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(dev)
    kw = {"device_id": torch.device("cuda", dev)} if a.backend == "nccl" else {}
    dist.init_process_group(a.backend, rank=rank, world_size=world, **kw)
    lo, hi = shard_range(a.total, world, rank)
    n = hi - lo
    env = env_hip.make(a.env, num_envs=n, device=dev, seed=a.seed, env_id_offset=lo)
    env.k_reset()
    K = a.gather_every
    if K > 1:
        assert a.steps % K == 0
        g = BlockRewardDoneGather(n, world, torch.device("cuda", dev), dist, block=K, force_collective=a.force_collective, direct=direct)
    else:
        g = RewardDoneGather(n, world, torch.device("cuda", dev), dist, force_collective=a.force_collective, direct=direct, depth=a.depth)
    if a.force_collective:
        assert g.force_collective and not g.host_stage
        assert (g.direct is not None) == bool(direct) and (g.side is not None) == (direct == "side")
    g.bind(env)                                                         # the step writes the packed record itself (as bench.py runs it)
    gen = torch.Generator(); gen.manual_seed(1234)                      # CPU generator: the same stream on every rank
    rew, done = [], []

    def take(b):
        r_all, d_all = g.result(b)
        if K > 1:                                                       # a block: K steps at once
            rew.extend(r_all.cpu().numpy().copy()); done.extend(d_all.cpu().numpy().copy())
        else:
            rew.append(r_all.cpu().numpy().copy()); done.append(d_all.cpu().numpy().copy())

    prev = None
    for k in range(a.steps):
        act_all = torch.rand((a.total, env.cm.act_dim), generator=gen) * 2 - 1
        g.before_step()                                                 # (ordering invariant: gym_kmanip_amd/dist.py)
        env.step_flat(act_all[lo:hi].contiguous().cuda())
        b = g.post(env.reward, env.done)
        if b is None:                                                   # (block mode: no exchange completed by this step)
            continue
        if prev is not None:                                            # pipelined like bench.py: step k's exchange is in
            take(prev)                                                  # flight while step k-1's result is consumed
        prev = b
    take(prev)
    if a.force_collective:             # evidence for the test: every step's exchange was a real asynchronous collective
        assert all(w is None for w in g.pending) and g.k == a.steps
    st = env.get_state()
    np.savez(os.path.join(a.out, "rank%d.npz" % rank), rew=np.array(rew), done=np.array(done), obs=env.obs.cpu().numpy(),
             qpos=st[0], qvel=st[1], ctrl=st[2], lo=lo, hi=hi, backend=dist.get_backend(),
             ipc_legacy=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", ""))
    env.k_close()
    g.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
