#!/usr/bin/env python3
"""bench.py -- env steps/sec of the hot path (KManipSoloArm @ 4096 envs per GPU) on N MI355X.

A "step" is one control step of every env on the rank = one pass of the hot path
(decode + IK -> 10 physics sub-steps -> reward/obs/done, auto-reset every 64 steps) over one batch of
synthetic actions (i.i.d. U(-1,1) float32, pre-generated, resident in HBM).  One process per GPU; envs
shard by global env index with no data-path collective except the per-step reward/done all-gather the
north star names (RCCL, async, off the critical path).  Rank 0 prints ONE JSON line.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--envs-per-gpu 4096] [--env KManipSoloArm]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def algorithmic_bytes_per_env_step(cm):
    """DESIGN.md section 'Roofline': float64 state read once + written once, action read, outputs written."""
    state = (cm.nq + cm.nv + cm.nu + cm.nv) * 8          # qpos, qvel, ctrl, qacc_warmstart
    return 2 * state + cm.act_dim * 4 + cm.obs_dim * 8 + 8 + 1


def measured_traffic(kernel_prefix="void k_step"):
    """HBM bytes per launch of the dominant kernel from the committed PMC summary (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE in separate passes, KB units; profiles/<round>_pmc_hbm.json).  bench.py cannot run rocprofv3 on
    itself, so the number is the last committed measurement of this kernel, or None."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_hbm.json")))
    if not files:
        return None, None
    try:
        d = json.load(open(files[-1]))
        for k, v in d.items():
            if k.startswith(kernel_prefix + "<10, 16, 1") and "true>" not in k:   # SoloArm / Newton, single-step kernel
                # FETCH_SIZE under-reports wide 16 B/lane streams by 2x on gfx950; these are 8 B/lane column reads
                # (uncalibrated width): reported as counted, see DESIGN.md
                return (v.get("FETCH_SIZE_KB_avg_per_launch", 0) + v.get("WRITE_SIZE_KB_avg_per_launch", 0)) * 1024.0, os.path.basename(files[-1])
    except Exception:  # noqa: BLE001
        pass
    return None, None


def measure_chunked(env, cm, n, K, gen):
    """Secondary measurement: the same workload through kmanip_step_chunk (K pre-supplied actions per env and launch,
    as an action-chunking policy or a scripted stream provides).  Not the headline: the metric is per-step stepping."""
    import torch
    acts = (torch.rand((K, n, cm.act_dim), generator=gen, device="cuda") * 2 - 1).contiguous()
    obs = torch.empty((K, n, cm.obs_dim), dtype=torch.float64, device="cuda")
    rew = torch.empty((K, n), dtype=torch.float64, device="cuda"); done = torch.empty((K, n), dtype=torch.uint8, device="cuda")
    for _ in range(4):
        env.step_chunk(acts, obs, rew, done)
    torch.cuda.synchronize()
    launches = 16
    t0 = time.perf_counter()
    for _ in range(launches):
        env.step_chunk(acts, obs, rew, done)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"api": "kmanip_step_chunk", "steps_per_launch": K, "value": n * K * launches / dt, "unit": "env steps/s",
            "ms_per_env_step_batch": dt / (K * launches) * 1e3,
            "note": "no launch boundary per step: waves do not wait for the batch's slowest env at every step"}


def measured_valu(kernel_prefix="void k_step<10, 16, 1"):
    """VALU-side view of the dominant kernel from the committed SQ counters (profiles/<round>_sq_counters.json, two
    rocprofv3 --pmc passes, tools/collect_sq.sh): the share of wave cycles spent issuing VALU instructions and the
    share of lanes active in them.  Their product is the fraction of the chip's FP64 vector issue slots doing work
    (one wave per SIMD here), the bound SURVEY 8d asks for beside the HBM one; MFMA utilisation is 0 (no MFMA
    instruction in the kernel: there is no dense contraction on this path)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_sq_counters.json")))
    if not files:
        return None
    try:
        d = json.load(open(files[-1]))
        for k, v in d.items():
            if k.startswith(kernel_prefix) and "true>" not in k:
                issue = v["SQ_ACTIVE_INST_VALU"] / v["SQ_WAVE_CYCLES"]
                lanes = v["SQ_THREAD_CYCLES_VALU"] / (v["SQ_INSTS_VALU"] * 64.0)
                return {"valu_issue_frac": issue, "lane_util": lanes, "fp64_vector_slot_frac": issue * lanes,
                        "wait_frac": v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], "mfma_util": 0.0,
                        "source": os.path.basename(files[-1])}
    except Exception:  # noqa: BLE001
        pass
    return None


def usable_cores():
    """Host cores this process may really use: affinity mask capped by the cgroup CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0]); p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, int(q / p + 0.5)))
            break
        except Exception:  # noqa: BLE001
            continue
    return n


def measure_variant(env_id, solver, n, local_rank, rank, steps=16, warmup=12):
    """Short single-rank measurement of the other constraint solver on the same workload (reported, not `value`)."""
    from gym_kmanip_amd import env_hip
    from gym_kmanip_amd.model import compile_model
    cm = compile_model(env_id, auto_reset=True, solver=solver)
    env = env_hip.KManipEnvHip(cm, num_envs=n, device=local_rank, seed=0, env_id_offset=rank * n)
    gen = torch.Generator(device="cuda"); gen.manual_seed(99)
    acts = [(torch.rand((n, cm.act_dim), generator=gen, device="cuda") * 2 - 1).contiguous() for _ in range(8)]
    env.k_reset()
    for k in range(warmup):
        env.step_flat(acts[k % 8])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(steps):
        env.step_flat(acts[k % 8])
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    env.k_close()
    return {"solver": solver, "value": n * steps / dt, "unit": "env steps/s", "steps": steps, "ms_per_step": dt / steps * 1e3}


def cpu_baseline(cm, n_envs, budget_s=12.0):
    """The oracle (a C port of the reference path; the reference itself cannot run here) timed on this
    box's host cores with OpenMP over envs, on a bounded sample of the same workload."""
    from oracle.oracle import Oracle
    cores = usable_cores()
    n = min(n_envs, 1024)
    o = Oracle(cm, n, seed=0)
    o.reset()
    rng = np.random.default_rng(0)
    acts = rng.uniform(-1, 1, (8, n, cm.act_dim)).astype(np.float32)
    for k in range(10):                   # untimed: let the cubes land so contacts are in the sample
        o.step(acts[k % 8], nthreads=cores)
    t0 = time.perf_counter(); steps = 0
    while True:
        o.step(acts[steps % 8], nthreads=cores); steps += 1
        dt = time.perf_counter() - t0
        if dt > budget_s or steps >= 54:
            break
    return {"value": n * steps / dt, "unit": "env steps/s", "cores": cores, "kind": "port",
            "sample": "%d envs x %d control steps (episode steps 10..%d, contacts active), OpenMP over envs, %d threads"
                      % (n, steps, 10 + steps, cores)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=64, help="untimed steps (default: one full 64-step episode, so timed steps see the steady-state mix)")
    ap.add_argument("--envs-per-gpu", type=int, default=4096)
    ap.add_argument("--envs-total", type=int, default=0,
                    help="strong-scaling variant: split this many envs over the GPUs (contiguous shards) instead of --envs-per-gpu each")
    ap.add_argument("--env", default="KManipSoloArm")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--depth", type=int, default=0, help="also render a DxD gripper-cam depth image per env each step (BASELINE config 5)")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--solver-iterations", type=int, default=100, help="solver iteration cap (MuJoCo default 100); ablation only")
    ap.add_argument("--solver", default="newton", choices=["pgs", "newton"],
                    help="newton = MuJoCo default, what the reference runs; pgs = the north star's named solver (100 sweeps)")
    ap.add_argument("--no-pgs-variant", action="store_true", help="skip the extra short PGS measurement")
    ap.add_argument("--chunk", type=int, default=16, help="also time kmanip_step_chunk with this many control steps per launch (0: skip)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus

    from gym_kmanip_amd import env_hip
    from gym_kmanip_amd.model import compile_model
    cm = compile_model(args.env, auto_reset=True, solver_iterations=args.solver_iterations, solver=args.solver)
    if args.envs_total:
        assert args.envs_total % world == 0, "--envs-total must be a multiple of the GPU count (equal shards for the gather)"
        from gym_kmanip_amd.dist import shard_range
        lo, hi = shard_range(args.envs_total, world, rank)
        n, off = hi - lo, lo
    else:
        n, off = args.envs_per_gpu, rank * args.envs_per_gpu
    env = env_hip.KManipEnvHip(cm, num_envs=n, device=local_rank, seed=0, env_id_offset=off)
    gen = torch.Generator(device="cuda"); gen.manual_seed(1234 + rank)
    nbank = 16
    acts = [(torch.rand((n, cm.act_dim), generator=gen, device="cuda") * 2 - 1).contiguous() for _ in range(nbank)]
    env.k_reset()

    gather = None
    if dist is not None and not args.no_gather:
        from gym_kmanip_amd.dist import RewardDoneGather
        gather = RewardDoneGather(n, world, torch.device("cuda", local_rank), dist)

    depth_buf = torch.empty((n, args.depth, args.depth), dtype=torch.float32, device="cuda") if args.depth else None

    def one_step(k):
        env.step_flat(acts[k % nbank])
        if depth_buf is not None:
            env.render_depth("grip_r", args.depth, args.depth, out=depth_buf)
        if gather is not None:
            gather.post(env.reward, env.done)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for k in range(args.warmup):
        one_step(k)
    barrier()
    env.enable_timing(True)
    t0 = time.perf_counter()
    for k in range(args.steps):
        one_step(args.warmup + k)
    if gather is not None:
        gather.wait()
    barrier()
    dt = time.perf_counter() - t0
    ik_ms, dyn_ms, nt = env.timing_summary()
    env.enable_timing(False)
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        total_envs = args.envs_total if args.envs_total else world * n
        total_env_steps = total_envs * args.steps
        bytes_per_launch = algorithmic_bytes_per_env_step(cm) * n
        dyn_avg_s = dyn_ms / max(nt, 1) * 1e-3
        achieved = bytes_per_launch / dyn_avg_s / 1e9
        traffic, traffic_src = measured_traffic() if (args.solver == "newton" and args.env == "KManipSoloArm" and n == 4096) else (None, None)
        out = {
            "metric": "env steps/sec (whole node), KManipSoloArm @4096 envs, 1/2/4/8 MI355X",
            "value": total_env_steps / dt, "unit": "env steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if args.envs_total else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s, %d envs per GPU (%d total), no cameras, random U(-1,1) actions, 64-step episodes with auto-reset"
                                   % (args.env, n, total_envs),
                       "envs_per_gpu": n, "depth_image": ("%dx%d float32 grip_r" % (args.depth, args.depth)) if args.depth else None,
                       "solver": args.solver, "solver_iterations": args.solver_iterations,
                       "sharding": "contiguous env-index blocks, 1 process per GPU",
                       "collective": "async all_gather of (reward, done) per step" if gather is not None else "none"},
            "roofline": {"bound": "hbm", "kernel": "k_step (before_step decode+IK fused with the 10 physics sub-steps)", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "bytes_per_env_step": algorithmic_bytes_per_env_step(cm),
                         "kernel_ms_avg": {"k_step": dyn_ms / max(nt, 1), "launch_gap": ik_ms / max(nt, 1)},
                         "valu": measured_valu() if (args.solver == "newton" and args.env == "KManipSoloArm" and n == 4096) else None,
                         "note": "latency/FP64-VALU bound by construction (SURVEY 8d): HBM traffic per env-step is ~1.2 KB"},
        }
        if not args.no_pgs_variant and args.solver == "newton" and world == 1:
            out["pgs_variant"] = measure_variant(args.env, "pgs", n, local_rank, rank)
        if args.chunk > 1 and world == 1 and not args.depth:
            out["chunked_variant"] = measure_chunked(env, cm, n, args.chunk, gen)
        if not args.no_cpu_baseline and world == 1:      # the CPU leg is timed on rank 0 of the 1-GPU run only
            out["cpu_baseline"] = cpu_baseline(cm, n)
            out["cpu_baseline"]["solver"] = args.solver
        print(json.dumps(out), flush=True)
    env.k_close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
