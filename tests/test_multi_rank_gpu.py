"""The N > 1 path with the HIP engine: world_size-2 rank processes, each stepping its contiguous env shard through the
C ABI and all-gathering (reward, done) every step, must reproduce a single-process run of all envs BIT FOR BIT (envs
share only the read-only model -- one independent Physics per env in the reference, env_sim.py:206-211 -- and the RNG is
keyed by the global env id).  With >= 2 GPUs: one rank per GPU over RCCL.  On a one-GPU box the same code is rehearsed
with both ranks on cuda:0 and gloo in place of RCCL (RCCL refuses two ranks on one device)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _run_ranks(out, world, extra):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "tools", "rank_worker.py"), "--out", out] + extra, env=env))
    try:
        codes = [p.wait(timeout=300) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert codes == [0] * world, codes


@pytest.mark.parametrize("env_id,total,steps,direct,depth", [("KManipSoloArm", 512, 70, "off", 2), ("KManipTorso", 128, 66, "off", 2),
                                                             ("KManipSoloArm", 512, 70, "off", 16),      # bench.py's N > 1 default
                                                             ("KManipSoloArm", 512, 70, "side", 16)])    # its opt-in direct exchange
def test_two_ranks_match_one_rank_bitwise(tmp_path, env_id, total, steps, direct, depth):
    import torch
    assert torch.cuda.is_available()
    from gym_kmanip_amd import env_hip
    multi = torch.cuda.device_count() >= 2
    if direct != "off" and not multi:
        pytest.skip("the direct ncclAllGather exchange needs the two ranks on two GPUs (RCCL refuses two ranks on one device)")
    extra = ["--env", env_id, "--total", str(total), "--steps", str(steps), "--direct", direct, "--depth", str(depth)]
    extra += ["--backend", "nccl"] if multi else ["--backend", "gloo", "--same-device"]
    _run_ranks(str(tmp_path), 2, extra)
    ranks = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(2)]
    # single-process reference over all envs
    ref = env_hip.make(env_id, num_envs=total, seed=9, env_id_offset=0)
    ref.k_reset()
    gen = torch.Generator(); gen.manual_seed(1234)
    for k in range(steps):
        act_all = torch.rand((total, ref.cm.act_dim), generator=gen) * 2 - 1
        ref.step_flat(act_all.cuda())
        rew, done = ref.reward.cpu().numpy(), ref.done.cpu().numpy()
        for r in ranks:                                     # every rank saw the whole job's (reward, done) at every step
            assert np.array_equal(r["rew"][k], rew), k
            assert np.array_equal(r["done"][k], done), k
    assert ranks[0]["done"][63].all() and not ranks[0]["done"][62].any()      # the auto-reset boundary was crossed
    st = ref.get_state(); obs = ref.obs.cpu().numpy()
    for r in ranks:
        lo, hi = int(r["lo"]), int(r["hi"])
        assert np.array_equal(r["obs"], obs[lo:hi]) and np.array_equal(r["qpos"], st[0][lo:hi])
        assert np.array_equal(r["qvel"], st[1][lo:hi]) and np.array_equal(r["ctrl"], st[2][lo:hi])
    assert int(ranks[0]["hi"]) == int(ranks[1]["lo"]) == total // 2
    ref.k_close()


@pytest.mark.parametrize("every,direct,depth", [(1, "off", 2), (7, "off", 2), (1, "stream", 2), (1, "side", 2), (7, "side", 2),
                                                (1, "side", 16), (1, "off", 5)])
def test_rccl_device_collective_world1(tmp_path, every, direct, depth):
    """RCCL executes: ONE rank, backend "nccl", on cuda:0, under the HSA_ENABLE_IPC_MODE_LEGACY=0 the ranks of an N > 1 job get.
    RewardDoneGather(force_collective=True) keeps the world == 1 short cut out of the way, so every step goes through
    before_step() (Work.wait() of the collective two steps back + record selection), the step launch that writes the packed
    record, and post() = all_gather_into_tensor(async_op=True) on the DEVICE buffers -- the code path of the multi-GPU job, which
    a gloo rehearsal (host staging, synchronous) never touches.  What the rank gathered must be what a plain run computes.
    every = 7: dist.BlockRewardDoneGather -- the engine's record re-bound to row j of a [7, n, 2] block every step, one
    asynchronous all-gather per seven steps, two blocks in flight.
    direct = "stream" / "side": the same exchange as ncclAllGather through dist.RcclDirect (ctypes on torch's librccl.so, a
    communicator of its own made from an ncclUniqueId) on the step's stream / on the gather's side stream with its two events.
    depth > 2: the ring of record buffers that lets a rank run ahead of a crawling peer (the record is re-bound every step)."""
    import torch
    from gym_kmanip_amd import env_hip
    env_id, total, steps = "KManipSoloArm", 512, 70
    _run_ranks(str(tmp_path), 1, ["--env", env_id, "--total", str(total), "--steps", str(steps), "--backend", "nccl", "--force-collective",
                                  "--gather-every", str(every), "--direct", direct, "--depth", str(depth)])
    r = np.load(os.path.join(str(tmp_path), "rank0.npz"))
    assert str(r["backend"]) == "nccl" and str(r["ipc_legacy"]) == "0"
    ref = env_hip.make(env_id, num_envs=total, seed=9, env_id_offset=0)
    ref.k_reset()
    gen = torch.Generator(); gen.manual_seed(1234)
    for k in range(steps):
        ref.step_flat((torch.rand((total, ref.cm.act_dim), generator=gen) * 2 - 1).cuda())
        assert np.array_equal(r["rew"][k], ref.reward.cpu().numpy()), k
        assert np.array_equal(r["done"][k], ref.done.cpu().numpy()), k
    assert r["done"][63].all() and not r["done"][62].any()
    assert np.array_equal(r["obs"], ref.obs.cpu().numpy()) and np.array_equal(r["qpos"], ref.get_state()[0])
    ref.k_close()


@pytest.mark.parametrize("how", ["default", "flag_side", "env_side"])
def test_bench_rccl_world1_line(how):
    """`python bench.py --rccl-world1`: the N = 1 bench as a one-rank RCCL job with the per-step all-gather forced through the
    device collective (init_process_group("nccl"), the rank-count all-reduce, bind + before_step / post around every step).
    Default = torch.distributed's all_gather_into_tensor (the mainstream path, what an N > 1 run gets); the direct exchange is
    opt-in by flag or by KMANIP_GATHER_DIRECT; the line records which one ran and every KMANIP_* variable that was set."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "KMANIP_GATHER_DIRECT")}
    flags = ["--gather-direct", "side"] if how == "flag_side" else []
    if how == "env_side":
        env["KMANIP_GATHER_DIRECT"] = "side"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rccl-world1", "--steps", "16", "--warmup", "4",
                        "--envs-per-gpu", "4096", "--no-variants", "--no-cpu-baseline"] + flags, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["config"]["backend"] == "nccl" and d["config"]["rccl_ranks_seen"] == 1
    assert d["config"]["collective"].startswith("async all_gather") and d["value"] > 0
    assert d["config"]["gather_direct"] == ("off" if how == "default" else "side") and d["config"]["gather_depth"] == 2
    assert ("ncclAllGather issued directly" in d["config"]["collective"]) == (how != "default")
    assert d["config"]["kmanip_env"].get("KMANIP_GATHER_DIRECT") == ("side" if how == "env_side" else None)


@pytest.mark.parametrize("shape", ["solo_256", "config4_torso_8192"])
def test_bench_self_launch_two_ranks(tmp_path, shape):
    """`python bench.py --gpus 2` (no torchrun) end to end: the parent spawns both ranks, rank 0 prints the one JSON line.
    One-GPU boxes rehearse it with KMANIP_BENCH_ONE_GPU=1 / KMANIP_BENCH_BACKEND=gloo (both ranks on cuda:0)."""
    import json
    import torch
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    if torch.cuda.device_count() < 2:
        env.update(KMANIP_BENCH_ONE_GPU="1", KMANIP_BENCH_BACKEND="gloo")
    # config4_torso_8192: BASELINE config 4's per-GPU shape (KManipTorso, 8192 envs per rank; cost-sorted wave slots on)
    cfg = ["--envs-per-gpu", "256"] if shape == "solo_256" else ["--env", "KManipTorso", "--envs-per-gpu", "8192"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "8", "--warmup", "2"] + cfg
                       + ["--no-variants", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["rccl_ranks_seen"] == 2 and d["scaling"] == "weak"
    assert d["config"]["envs_per_gpu"] == (256 if shape == "solo_256" else 8192) and d["value"] > 0 and d["steps"] == 8
    assert "collective" in d["config"] and d["config"]["collective"].startswith("async all_gather")
    # default flags: the exchange goes through torch.distributed's wrapper (never the ctypes-made second communicator)
    assert d["config"]["gather_direct"] == "off" and d["config"]["gather_depth"] == (16 if d["config"]["backend"] == "nccl" else 2)
    assert set(d["config"]["kmanip_env"]) >= ({"KMANIP_BENCH_ONE_GPU", "KMANIP_BENCH_BACKEND"} if torch.cuda.device_count() < 2 else set())
