#!/usr/bin/env python3
"""bench.py -- env steps/sec of the hot path (default: KManipSoloArm @ 4096 envs per GPU) on N MI355X.

A "step" is one control step of every env on the rank = one pass of the hot path
(decode + IK -> 10 physics sub-steps -> reward/obs/done, auto-reset every 64 steps) over one batch of
synthetic actions: action_space.sample() of every env (i.i.d. U[-1,1) float32) from the counter-based stream SURVEY 8d specifies
-- Philox4x32-10 keyed (seed; global env id, episode, step), drawn ON THE DEVICE by kmanip_sample_action for every step of the
run before the timed region starts (resident in HBM); the CPU baseline leg draws the same stream from the oracle.  One process per GPU; envs
shard by global env index with no data-path collective except the per-step reward/done all-gather the
north star names (RCCL, async, off the critical path).  Rank 0 prints ONE JSON line.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--envs-per-gpu 4096] [--env KManipSoloArm]

`--gpus N` with N > 1 and no RANK in the environment: this process starts N fresh rank processes (one per GPU,
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set) BEFORE touching the GPU, waits for them and exits with their worst
code; under `python -m torch.distributed.run` the ranks are already there and it just runs as one of them.

Steady state: the timed region always runs on DESYNCHRONISED envs -- before the W warm-up steps an untimed
preparation staggers the per-env episode phase uniformly over 0..63 and pre-rolls one full episode, so that every
launch sees the same mix of falling / landing / resting cubes and reset envs whatever K and W are (episodes are
exactly 64 steps long, so envs that all start together would otherwise stay phase-locked for ever and a short window
would time whichever phase it happens to fall on).  `phase_locked` in the JSON line is the same workload without
the stagger over whole episodes.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

EPISODE = 64                      # MAX_EPISODE_STEPS, gym_kmanip/__init__.py:28
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md
FP64_VECTOR_PEAK_TFLOPS = 78.6    # 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1024, help="timed control steps (default ~1 s at 4096 envs)")
    ap.add_argument("--warmup", type=int, default=16, help="untimed steps after the (also untimed) desynchronising preparation")
    ap.add_argument("--envs-per-gpu", type=int, default=4096)
    ap.add_argument("--envs-total", type=int, default=0,
                    help="strong-scaling variant: split this many envs over the GPUs (contiguous shards) instead of --envs-per-gpu each")
    ap.add_argument("--env", default="KManipSoloArm")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--depth", type=int, default=0, help="also render a DxD gripper-cam depth image per env each step (BASELINE config 5)")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--gather-every", type=int, default=1,
                    help="K > 1: the (reward, done) records of K steps per all-gather (dist.BlockRewardDoneGather) instead of one per step")
    ap.add_argument("--gather-direct", choices=["auto", "off", "stream", "side"], default="auto",
                    help="issue the (reward, done) all-gather as ncclAllGather through dist.RcclDirect -- on the step's own stream, or on a "
                         "side stream tied to it by two events -- instead of through torch.distributed's wrapper (off).  auto = off: "
                         "torch.distributed's all_gather_into_tensor is the mainstream path and the default until the direct one has "
                         "met two devices (it has only ever run at world size 1).  Opt in with --gather-direct side|stream or "
                         "KMANIP_GATHER_DIRECT=side|stream (the wrapper costs the step's stream 36 us per exchange on one MI355X, "
                         "the side stream 19 us: profiles/r05_rccl_direct.txt)")
    ap.add_argument("--rendezvous-timeout", type=float, default=float(os.environ.get("KMANIP_RENDEZVOUS_TIMEOUT", "90")),
                    help="wall-clock bound in seconds on every start-up step that waits for the other ranks (init_process_group, the "
                         "rank-count all-reduce, the direct communicator and its self-test): past it the rank prints one line and exits 6")
    ap.add_argument("--time-every", type=int, default=0,
                    help="HIP events around every k-th launch of the timed region (an event pair costs the stream ~5 us).  0 (default): "
                         "every launch for windows under 256 steps, else steps // 128; 1: every launch (tools/ab.sh: libraries older than 0.30 time every launch whatever k is)")
    ap.add_argument("--run-timeout", type=float, default=-1.0,
                    help="N > 1 only: wall-clock bound in seconds on the warm-up + timed steps (their exchanges, barriers and the max over "
                         "ranks wait for peers too); past it the rank prints one line and exits 6.  -1 (default): 120 s + 10 ms a step; 0: none")
    ap.add_argument("--gather-depth", type=int, default=0,
                    help="ring of (reward, done) record buffers (dist.RewardDoneGather(depth=...)): a rank may run that many steps ahead "
                         "of the slowest one before it waits.  0: 16 on RCCL across ranks (an IK crawl is up to seven steps long; "
                         "tools/scaling_model.py), 2 otherwise")
    ap.add_argument("--gather-serial", action="store_true",
                    help="N > 1 (or --rccl-world1): the step's stream waits for its own (reward, done) all-gather before the next step "
                         "instead of overlapping it with the next step (A/B switch; dist.RewardDoneGather(overlap=False))")
    ap.add_argument("--rccl-world1", action="store_true",
                    help="N = 1 only: run as a one-rank RCCL job (backend nccl, world_size 1) with the per-step (reward, done) "
                         "all-gather forced through all_gather_into_tensor(async_op=True) -- executes the device-collective path "
                         "of the N > 1 job on a one-GPU box and prices it against the plain N = 1 line")
    ap.add_argument("--rccl-one-channel", action="store_true",
                    help="N > 1 only: NCCL_MIN/MAX_NCHANNELS=1, NCCL_NTHREADS=64 for the per-step all-gather (A/B switch; default: RCCL's own settings)")
    ap.add_argument("--no-stagger", action="store_true", help="keep all envs phase-locked (episode phase = step index for every env)")
    ap.add_argument("--solver-iterations", type=int, default=100, help="solver iteration cap (MuJoCo default 100); ablation only")
    ap.add_argument("--solver", default="newton", choices=["pgs", "newton"],
                    help="newton = MuJoCo default, what the reference runs; pgs = the north star's named solver (100 sweeps)")
    ap.add_argument("--no-variants", action="store_true", help="skip the secondary measurements (long window, phase-locked, PGS, chunked, seam, two handles, the other BASELINE configs)")
    ap.add_argument("--chunk", type=int, default=16, help="also time kmanip_step_chunk with this many control steps per launch (0: skip)")
    return ap.parse_args(argv)


# --------------------------------------------------------------------------------------------------------------------
# multi-GPU self-launch: the parent never initialises HIP and never execs
import contextlib


@contextlib.contextmanager
def stdout_to_stderr():
    """File descriptor 1 -> 2 for the duration (native libraries' prints included): rank 0's stdout carries ONE JSON line."""
    sys.stdout.flush()
    saved = os.dup(1)
    try:
        os.dup2(2, 1)
        yield
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


class Deadline:
    """Wall-clock bound on a start-up step that waits for other ranks.  A rendezvous that HANGS (a peer that never arrives, an
    IPC mode RCCL cannot use) raises nothing, so a try/except never sees it and the driver's own timeout would eat the whole
    run: a watchdog thread prints a one-line diagnosis and ends THIS process with code 6 (os._exit: no re-exec, no retry in
    the process; the parent -- spawn_ranks or torch.distributed.run -- then stops the peers).  The waits it guards release the
    GIL (ctypes calls, torch's c10d store / stream synchronisation), so the thread gets to run."""
    EXIT_CODE = 6

    def __init__(self, what, seconds, rank=0, exit_code=None, _exit=os._exit):
        self.what, self.seconds, self.rank = what, float(seconds), rank
        self.code = self.EXIT_CODE if exit_code is None else exit_code
        self._exit = _exit
        self._timer = None

    def _expired(self):
        try:
            sys.stderr.write("bench.py: rank %d: %s did not finish within %.0f s (a peer rank missing or hung, MASTER_ADDR/PORT, or "
                             "HSA_ENABLE_IPC_MODE_LEGACY=%s) -- exiting %d\n"
                             % (self.rank, self.what, self.seconds, os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "<unset>"), self.code))
            sys.stderr.flush()
        finally:
            self._exit(self.code)

    def __enter__(self):
        import threading
        if self.seconds > 0:
            self._timer = threading.Timer(self.seconds, self._expired)
            self._timer.daemon = True
            self._timer.start()
        return self

    def __exit__(self, *exc):
        if self._timer is not None:
            self._timer.cancel()
        return False


TIME_EVERY = 0          # --time-every: 0 = automatic (below); k = events around every k-th launch (1: every launch -- what an A/B against a
                        # library older than 0.30, which knows only "on", must use for both sides)


def timing_every(steps):
    """The library's HIP events (kmanip_enable_timing) around every k-th launch of the timed region: an event pair costs the step's
    stream about 5 us -- 1 % of a 4096-env step, paid inside `value` -- so long windows sample (>= 128 launches timed, spread evenly
    over the window; the episode phases are staggered, every step holds the same mix of envs) and short ones time every launch."""
    if TIME_EVERY > 0:
        return TIME_EVERY
    return max(1, int(steps) // 128)


def run_timeout_seconds(args):
    """The bound on the warm-up + timed region of an N > 1 run: --run-timeout if given (0 = none), else 120 s + 10 ms a step --
    two hundred times what a healthy step takes, so that only a hang meets it."""
    if getattr(args, "run_timeout", None) is not None and args.run_timeout >= 0:
        return float(args.run_timeout)
    return 120.0 + 0.01 * (args.warmup + args.steps)


def kmanip_env_vars():
    """Every KMANIP_* variable set in this process's environment: several change the timed launch's shape (KMANIP_EPB,
    KMANIP_SPREAD*, KMANIP_COST_SORT, ...) though never its results -- the bench line records them (`config.kmanip_env`)."""
    return {k: v for k, v in sorted(os.environ.items()) if k.startswith("KMANIP_") and k != "KMANIP_BENCH_SPAWNED"}


def free_port():
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(args, argv):
    """Start args.gpus fresh rank processes of this script (one per GPU) and wait for them.  Rank 0 inherits stdout, so
    its single JSON line is this command's output; the other ranks' stdout goes to stderr."""
    world = args.gpus
    port = int(os.environ.get("MASTER_PORT") or free_port())
    procs = []
    for r in range(world):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), KMANIP_BENCH_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL across processes needs it on this pool
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    deadline = time.time() + float(os.environ.get("KMANIP_BENCH_TIMEOUT", "1500"))
    live = list(procs)
    while live:
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code
        if (rc != 0 or time.time() > deadline) and live:
            # a rank died (or the run timed out): its peers would wait for it in the rendezvous -- stop exactly the
            # processes started above, after a short grace for their own error messages
            time.sleep(2.0)
            for q in live:
                if q.poll() is None:
                    q.kill()
            for q in live:
                q.wait()
            rc = rc or -9
            break
        time.sleep(0.05)
    if rc != 0:
        sys.stderr.write("bench.py: a rank process failed (exit code %s)\n" % rc)
    return 1 if rc else 0


# --------------------------------------------------------------------------------------------------------------------
def camera_bytes_per_env_step(cm):
    """uint8 RGB camera observations of the *Vision env ids at the reference resolutions (__init__.py:157-161)."""
    from gym_kmanip_amd.model import CAMERAS
    return sum(CAMERAS[c].h * CAMERAS[c].w * 3 for c in cm.cameras)


def algorithmic_bytes_per_env_step(cm, depth=0, rgb=False):
    """float64 state read once + written once, action read, outputs written (+ the float32 depth image
    of BASELINE config 5 when rendered in the step, + the uint8 camera observations of a *Vision env id)."""
    state = (cm.nq + cm.nv + cm.nu + cm.nv) * 8          # qpos, qvel, ctrl, qacc_warmstart
    return 2 * state + cm.act_dim * 4 + cm.obs_dim * 8 + 8 + 1 + depth * depth * 4 + (camera_bytes_per_env_step(cm) if rgb else 0)


def _committed(name_glob, version=None):
    """The committed counter summary measured on library `version` (the last such file by name); without one, the last file
    by name (its `_meta.version` then tells the caller that it is stale)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", name_glob)))
    for f in reversed(files):
        try:
            if json.load(open(f)).get("_meta", {}).get("version") == version:
                return f
        except (OSError, ValueError):
            pass
    return files[-1] if files else None


def committed_counters(version, kernel_prefix):
    """Counter-derived figures of the dominant kernel (HBM traffic, VALU issue / lane utilisation, hardware-counted FP64
    FLOPs) from the LAST committed rocprofv3 --pmc summaries.  bench.py cannot run rocprofv3 on itself, so these are the
    last committed measurement of this kernel -- and only if that measurement was taken on the same library version
    (`_meta.version` in the file == kmanip_version()): a stale file is refused (None + `stale`)."""
    out = {"traffic": None, "traffic_source": None, "valu": None, "flops": None}
    f = _committed("*_pmc_hbm.json", version)
    if f:
        d = json.load(open(f))
        if d.get("_meta", {}).get("version") != version:
            out["traffic_source"] = "stale: %s was measured on %r" % (os.path.basename(f), d.get("_meta", {}).get("version"))
        else:
            for k, v in d.items():
                if k.startswith(kernel_prefix) and "true>" not in k:
                    # FETCH_SIZE under-reports wide 16 B/lane streams by 2x on gfx950; these are 8 B/lane column reads
                    # (uncalibrated width): reported as counted, see DESIGN.md
                    out["traffic"] = (v.get("FETCH_SIZE_KB_avg_per_launch", 0) + v.get("WRITE_SIZE_KB_avg_per_launch", 0)) * 1024.0
                    out["traffic_source"] = os.path.basename(f)
    f = _committed("*_sq_counters.json", version)
    if f:
        d = json.load(open(f))
        if d.get("_meta", {}).get("version") != version:
            out["valu"] = {"stale": "%s was measured on %r" % (os.path.basename(f), d.get("_meta", {}).get("version"))}
        else:
            for k, v in d.items():
                if k.startswith(kernel_prefix) and "true>" not in k:
                    issue = v["SQ_ACTIVE_INST_VALU"] / v["SQ_WAVE_CYCLES"]
                    lanes = v["SQ_THREAD_CYCLES_VALU"] / (v["SQ_INSTS_VALU"] * 64.0)
                    out["valu"] = {"valu_issue_frac": issue, "lane_util": lanes, "fp64_vector_slot_frac": issue * lanes,
                                   "wait_frac": v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], "mfma_util": 0.0,
                                   "source": os.path.basename(f)}
                    if "SQ_INSTS_VALU_FMA_F64" in v:
                        # per-launch FP64 FLOPs as the hardware counted them: wave-level instruction counts x 64 lanes
                        # (an upper bound: masked lanes are counted), FMA = 2 flops
                        fl = 64.0 * (2 * v["SQ_INSTS_VALU_FMA_F64"] + v.get("SQ_INSTS_VALU_ADD_F64", 0) + v.get("SQ_INSTS_VALU_MUL_F64", 0)
                                     + v.get("SQ_INSTS_VALU_TRANS_F64", 0))
                        out["flops"] = {"fp64_flops_per_launch_issued": fl, "fp64_flops_per_launch_active_lanes": fl * lanes,
                                        "source": os.path.basename(f)}
    return out


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


def probe_mujoco():
    """Run-time probe (SURVEY 8d plan (1)): is `mujoco` importable on THIS box, now?"""
    import importlib.util
    present = importlib.util.find_spec("mujoco") is not None
    return {"present": present,
            "note": "present (probed now)" if present else
                    "absent (probed now: importlib.util.find_spec('mujoco') is None on this box) -- no MuJoCo-timed baseline; "
                    "tools/mjcf_export.py writes the surrogate as mesh-free MJCF for a box that has it"}


def mujoco_baseline(cm, n_envs, budget_s=10.0):
    """Only when `mujoco` is importable (never on this pool): time MuJoCo's own mj_step on the build's mesh-free MJCF
    (tools/mjcf_export.py) with build-owned harness code -- 10 sub-steps per control step, ctrl = a random target inside
    ctrlrange every control step, one model, envs run back to back on one core.  It times the ENGINE under the path (rows
    a-2 / a-9), not the reference's Python around it."""
    import tempfile
    import numpy as np
    import mujoco
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
    import mjcf_export
    xml = mjcf_export.export(cm.asset)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, cm.asset["name"] + ".xml")
        open(path, "w").write(xml)
        m = mujoco.MjModel.from_xml_path(path)
    d = mujoco.MjData(m)
    rng = np.random.default_rng(0)
    lo, hi = m.actuator_ctrlrange[:, 0], m.actuator_ctrlrange[:, 1]
    t0 = time.perf_counter(); steps = 0
    while time.perf_counter() - t0 < budget_s:
        d.ctrl[:] = rng.uniform(lo, hi)
        for _ in range(cm.desc.n_sub_steps):
            mujoco.mj_step(m, d)
        steps += 1
        if steps % 64 == 0:
            mujoco.mj_resetData(m, d)
    dt = time.perf_counter() - t0
    return {"value": steps / dt, "unit": "env steps/s", "cores": 1, "kind": "mujoco-engine",
            "sample": "%d control steps of one env (10 x mj_step each), mujoco %s, the exported mesh-free surrogate MJCF" % (steps, mujoco.__version__)}


def cpu_baseline(cm, n_envs, budget_s=15.0):
    """The oracle (a C port of the reference path; the reference itself cannot run here: mujoco / dm_control / gymnasium
    are absent from this image and from the GPU box -- probed at run time below) timed on this box's host cores with OpenMP over
    envs, on a bounded sample of the SAME workload: all n_envs envs of the bench line, the device's Philox action stream, as many
    control steps as fit the time budget (the first ten are untimed so that the cubes have landed and contacts are in the sample)."""
    import numpy as np
    from oracle.oracle import Oracle
    cores = usable_cores()
    n = n_envs
    o = Oracle(cm, n, seed=0)
    o.reset()
    for k in range(10):                   # untimed: let the cubes land so contacts are in the sample
        o.step(o.sample_action(), nthreads=cores)
    t0 = time.perf_counter(); steps = 0
    while True:
        o.step(o.sample_action(), nthreads=cores); steps += 1       # the device's action stream (same Philox keys), drawn per step
        dt = time.perf_counter() - t0
        if dt > budget_s or steps >= 4 * EPISODE:
            break
    mj = probe_mujoco()
    out = {"value": n * steps / dt, "unit": "env steps/s", "cores": cores, "kind": "port",
           "sample": "%d envs x %d control steps (episode steps 10..%d, contacts active, auto-reset at 64; the device's Philox action stream), OpenMP over envs, %d threads"
                     % (n, steps, 10 + steps, cores),
           "mujoco": mj["note"]}
    if mj["present"]:
        try:
            out["mujoco_engine"] = mujoco_baseline(cm, n)
        except Exception as e:  # noqa: BLE001
            out["mujoco_engine"] = {"error": "%s: %s" % (type(e).__name__, e)}
    return out


def resolve_gather_direct(flag, env_value=None):
    """(mode, who asked): --gather-direct auto is torch.distributed's wrapper ("off") -- the path every gloo / world-1 / two-rank
    test runs -- unless KMANIP_GATHER_DIRECT names a direct mode; an explicit flag wins over the environment."""
    if flag != "auto":
        return flag, "flag"
    if env_value in ("side", "stream"):
        return env_value, "env"
    return "off", None


def check_ranks_seen(ranks_seen, world, rank):
    """Start-up self-check of the rank path: the all-reduce of ones must count every rank, or the collective backend is not
    spanning the job (a mis-set MASTER_* / IPC mode would otherwise show up only as a wrong whole-job value).  Returns the
    process exit code (0 = fine)."""
    if ranks_seen != world:
        sys.stderr.write("bench.py: rank %d: the collective backend sees %d rank(s), WORLD_SIZE is %d -- refusing to report a whole-job value\n"
                         % (rank, ranks_seen, world))
        return 4
    return 0


class Workload:
    """One rank's envs, prepared to the desynchronised steady state, stepping on the counter-based action stream
    (kmanip_sample_action: Philox keyed (seed; global env id, episode, step) -- fresh actions every step of every episode, as
    the loop this imitates samples them: examples/2_log_with_h5py.py:22-26)."""
    BANK_MAX = 4096                       # steps laid out ahead in HBM at most (4096 x 4096 envs x 7 x 4 B = 470 MB)

    def __init__(self, torch, env_id, n, device_index, rank, off, solver, solver_iterations, stagger=True, ik_max_nfev=0):
        from gym_kmanip_amd import env_hip
        from gym_kmanip_amd.model import compile_model
        import numpy as np
        self.torch = torch
        self.cm = compile_model(env_id, auto_reset=True, solver_iterations=solver_iterations, solver=solver, ik_max_nfev=ik_max_nfev)
        self.n = n
        self.env = env_hip.KManipEnvHip(self.cm, num_envs=n, device=device_index, seed=0, env_id_offset=off)
        self.act = torch.empty((n, self.cm.act_dim), dtype=torch.float32, device="cuda")
        self.bank = None
        self.k = 0
        self.env.k_reset()
        if stagger:
            # episode phase of env e = (global env id) % 64: the auto-reset then fires for n/64 envs at every step
            phase = ((off + np.arange(n)) % EPISODE).astype(np.int32)
            self.env.set_state(step=phase)
            self.run(EPISODE)             # one full episode: every env has been through its own reset since the stagger
        torch.cuda.synchronize()

    def lay_out(self, steps):
        """Draw the actions of the next `steps` control steps into HBM now (the timed region then only steps)."""
        torch = self.torch
        steps = min(steps, self.BANK_MAX)
        self.bank = torch.empty((steps, self.n, self.cm.act_dim), dtype=torch.float32, device="cuda")
        for k in range(steps):
            self.env.sample_action(self.bank[k], ahead=k)
        self.k = 0
        torch.cuda.synchronize()

    def step(self):
        if self.bank is not None and self.k < self.bank.shape[0]:
            a = self.bank[self.k]
        else:                             # beyond the bank: drawn in the loop (one tiny extra launch per step)
            a = self.env.sample_action(self.act)
        self.env.step_flat(a); self.k += 1

    def run(self, steps):
        for _ in range(steps):
            self.step()

    def timed(self, steps, warmup):
        torch = self.torch
        self.lay_out(warmup + steps)
        self.run(warmup)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        self.run(steps)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    def close(self):
        self.env.k_close()


def measure_variant(torch, args, n, local_rank, rank, solver, stagger, steps, warmup):
    w = Workload(torch, args.env, n, local_rank, rank, rank * n, solver, args.solver_iterations, stagger=stagger)
    dt = w.timed(steps, warmup)
    w.close()
    return {"solver": solver, "staggered": stagger, "value": n * steps / dt, "unit": "env steps/s", "steps": steps,
            "ms_per_step": dt / steps * 1e3, "timed_window_s": dt}


def measure_chunked(torch, w, K):
    """Secondary: the same workload through kmanip_step_chunk (K pre-supplied actions per env and launch, as an
    action-chunking policy or a scripted stream provides).  Not the headline: the metric is per-step stepping."""
    n, cm, env = w.n, w.cm, w.env
    launches, warm = 16, 4
    acts = torch.empty((warm + launches, K, n, cm.act_dim), dtype=torch.float32, device="cuda")
    for i in range(warm + launches):          # the stream's next (warm + launches) * K steps, laid out before the timed region
        for k in range(K):
            env.sample_action(acts[i, k], ahead=i * K + k)
    obs = torch.empty((K, n, cm.obs_dim), dtype=torch.float64, device="cuda")
    rew = torch.empty((K, n), dtype=torch.float64, device="cuda"); done = torch.empty((K, n), dtype=torch.uint8, device="cuda")
    for i in range(warm):
        env.step_chunk(acts[i], obs, rew, done)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(launches):
        env.step_chunk(acts[warm + i], obs, rew, done)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"api": "kmanip_step_chunk", "steps_per_launch": K, "value": n * K * launches / dt, "unit": "env steps/s",
            "ms_per_env_step_batch": dt / (K * launches) * 1e3,
            "note": "no launch boundary per step: waves do not wait for the batch's slowest env at every step"}


def measure_two_handles(torch, args, n, local_rank, steps=256, env_id=None):
    """Secondary: TWO independent batches of n envs (two handles, two streams -- e.g. a double-buffered actor: one batch steps
    while the policy works on the other's observations), each stepped one control step per call like `value`'s.  A single batch's
    launch is one residency round and ends with its slowest wave -- half the SIMD time of the launch is idle --;
    a second batch's waves fill the slots the first one's early finishers free.  Whole-GPU env steps/s over both batches."""
    ws = [Workload(torch, env_id or args.env, n, local_rank, 0, i * n, args.solver, args.solver_iterations, stagger=True) for i in range(2)]
    for w in ws:
        w.lay_out(16 + 3 * 8 + steps)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]

    def rounds(k):
        for _ in range(k):
            for w, s in zip(ws, streams):
                with torch.cuda.stream(s):
                    w.step()
    # two pool streams can land on the same hardware queue -- their kernels then run one after the other and the measurement
    # reads twice a single batch --: try three candidates for the second stream on eight rounds each, keep the fastest pairing
    best = None
    for _ in range(3):
        cand = torch.cuda.Stream()
        streams[1] = cand
        rounds(2)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rounds(6)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        if best is None or dt < best[0]:
            best = (dt, cand)
    streams[1] = best[1]
    rounds(16)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rounds(steps)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    for w in ws:
        w.close()
    return {"what": "two independent handles of %d envs each, one stream each, every handle one kmanip_step per call" % n,
            "value": 2 * n * steps / dt, "unit": "env steps/s", "steps": steps, "ms_per_round_of_both_steps": dt / steps * 1e3,
            "note": "not the headline (its workload is ONE batch of %d envs): what the GPU sustains when a second batch is in flight" % n}


def measure_seam(torch, w, steps=256):
    """The drop-in path: KManipEnvHip.k_step with a DICT of device tensors keyed like the reference action space
    (env_base.py:241-259 -> env_sim.py:196-200), returning the 5-tuple -- vs step_flat on the same envs."""
    n, cm, env = w.n, w.cm, w.env
    w.lay_out(2 * steps + 16)
    flat = [w.bank[k] for k in range(w.bank.shape[0])]
    bank = [{k: a[:, sl] for k, sl in cm.act_slices.items()} for a in flat]
    for k in range(8):
        env.k_step(bank[k % len(bank)])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(steps):
        terminated, reward, discount, obs, sim_time = env.k_step(bank[8 + k])
    torch.cuda.synchronize(); dt_seam = time.perf_counter() - t0
    for k in range(8):
        env.step_flat(flat[8 + steps + k])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(steps):
        env.step_flat(flat[16 + steps + k])
    torch.cuda.synchronize(); dt_flat = time.perf_counter() - t0
    return {"api": "KManipEnvHip.k_step(dict of device tensors) -> (terminated, reward, discount, obs dict, sim_time)",
            "value": n * steps / dt_seam, "unit": "env steps/s", "steps": steps, "step_flat_value": n * steps / dt_flat,
            "ratio_to_step_flat": dt_flat / dt_seam}


def measure_config(torch, env_id, n, local_rank, steps, warmup, depth=0, solver="newton"):
    """One short driver-clocked line for another BASELINE config on this GPU: the same desynchronised preparation, `steps` timed
    control steps (wall clock around them, HIP events on the launch stream around the kernels), its own roofline figures."""
    w = Workload(torch, env_id, n, local_rank, 0, 0, solver, 100, stagger=True)
    env, cm = w.env, w.cm
    if depth:
        env.bind_step_depth("grip_r", depth, depth)
    rgb_bufs = {c: env.render_rgb(c) for c in cm.cameras}

    def one():
        w.step()
        if rgb_bufs:
            env.render_cameras(out=rgb_bufs)         # all cameras of the observation in one launch (timed by the library: the step's render leg)

    w.lay_out(warmup + steps)
    for _ in range(warmup):
        one()
    torch.cuda.synchronize()
    env.enable_timing(timing_every(steps))
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ik_ms, dyn_ms, rnd_ms, nt = env.timing_summary()
    env.enable_timing(False)
    bpe = algorithmic_bytes_per_env_step(cm, depth, rgb=bool(rgb_bufs))
    kern_s = (dyn_ms + rnd_ms) / max(nt, 1) * 1e-3
    achieved = bpe * n / kern_s / 1e9
    behind = None
    if rgb_bufs or depth:
        # the same steps with the cameras rendered BEHIND them (pipeline.RenderBehind: qpos snapshot + a second stream) -- the
        # shape of a data-generation loop whose policy acts on the state and logs the images.  Not the line above: there every
        # step's images exist before the next step starts.
        from gym_kmanip_amd.pipeline import RenderBehind
        if depth:
            env.bind_step_depth(None)                 # (the in-step render of the line above is unbound: the image is rendered behind instead)
        rb = RenderBehind(env, cams=None if rgb_bufs else [], depth=("grip_r", depth, depth) if depth else None)
        w.lay_out(warmup + steps)
        for _ in range(warmup):
            w.step(); rb.after_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            w.step(); rb.after_step()
        torch.cuda.synchronize()
        dtb = time.perf_counter() - t0
        behind = {"what": "the same steps, every step's %s rendered from a qpos snapshot on a second stream while the next step runs (pipeline.RenderBehind)" % ("cameras" if rgb_bufs else "depth image"),
                  "value": n * steps / dtb, "unit": "env steps/s", "steps": steps, "ms_per_step": dtb / steps * 1e3}
    w.close()
    return {**({"render_behind": behind} if behind else {}), "workload": "%s, %d envs, 1 GPU%s, desynchronised" % (env_id, n, (", %dx%d float32 grip_r depth in the step" % (depth, depth)) if depth else
                                                                 ((", uint8 RGB cameras %s after the step" % "+".join(cm.cameras)) if rgb_bufs else "")),
            "value": n * steps / dt, "unit": "env steps/s", "steps": steps, "warmup": warmup, "ms_per_step": dt / steps * 1e3,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "bytes_per_env_step": bpe, "algorithmic_bytes_per_launch": bpe * n,
                         "kernel_ms_avg": {"k_step": dyn_ms / max(nt, 1), "k_render": rnd_ms / max(nt, 1), "launches_timed": nt, "timed_every": timing_every(steps)}}}


OTHER_CONFIGS = [   # BASELINE.json configs 3, 4 (one GPU's shard), 5 and the reference's own *Vision observation at config 5's width
    # (windows of >= 128 steps: one IK-crawl launch -- 3 ms, one in ~80 at 2048 envs -- moves a 48-step window by 6 %)
    ("config3_dualarm_8192", dict(env_id="KManipDualArm", n=8192, steps=128, warmup=8)),
    ("config4_torso_8192_shard", dict(env_id="KManipTorso", n=8192, steps=128, warmup=8)),
    ("config5_soloarm_2048_depth64", dict(env_id="KManipSoloArm", n=2048, steps=384, warmup=8, depth=64)),
    ("vision_soloarm_2048", dict(env_id="KManipSoloArmVision", n=2048, steps=192, warmup=8)),
]


def run_rank(args):
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        # dmabuf IPC (RCCL across processes needs it on this pool); read when the HIP runtime starts, so before torch is imported --
        # the driver exports it already, spawn_ranks sets it for its children: this covers a bare torch.distributed.run
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write("bench.py: WORLD_SIZE=%d but --gpus %d (launch with --nproc-per-node %d, or let bench.py spawn the ranks)\n"
                         % (world, args.gpus, args.gpus))
        return 2
    if not torch.cuda.is_available():
        sys.stderr.write("bench.py needs a HIP device (no CPU fallback)\n")
        return 3
    if local_rank >= torch.cuda.device_count() and os.environ.get("KMANIP_BENCH_ONE_GPU") != "1":
        sys.stderr.write("bench.py: rank %d has no GPU (%d visible)\n" % (local_rank, torch.cuda.device_count()))
        return 3
    # rehearsal knobs for a one-GPU box (never set by the driver): all ranks on cuda:0 and gloo in place of RCCL, which
    # refuses two ranks on one device -- everything else of the multi-rank path is the same code
    backend = os.environ.get("KMANIP_BENCH_BACKEND", "nccl")
    if os.environ.get("KMANIP_BENCH_ONE_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or args.rccl_world1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:      # --rccl-world1: a one-rank RCCL job on this GPU, so that the device-collective path executes
            os.environ.setdefault("MASTER_PORT", str(free_port()))
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # RCCL runs with its own defaults.  --rccl-one-channel (opt-in, never measured on two devices: DESIGN.md section 7) asks
        # for one channel of one wave for the 64 KB all-gather: k_step fills every SIMD with exactly one wave, so a collective
        # kernel that takes whole compute units pushes some of k_step's workgroups behind it; one wave can sit beside it.
        if args.rccl_one_channel:
            os.environ["NCCL_MIN_NCHANNELS"] = "1"
            os.environ["NCCL_MAX_NCHANNELS"] = "1"
            os.environ["NCCL_NTHREADS"] = "64"
        import datetime
        kw = {"device_id": torch.device("cuda", local_rank)} if backend == "nccl" else {}
        # every start-up wait for the peers is bounded (Deadline): a hang becomes a one-line diagnosis and exit code 6
        with stdout_to_stderr(), Deadline("init_process_group(%r)" % backend, args.rendezvous_timeout, rank):
            # RCCL prints a five-line version banner to STDOUT when its communicator is created
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=max(args.rendezvous_timeout, 10.0)), **kw)

    if args.envs_total:
        assert args.envs_total % world == 0, "--envs-total must be a multiple of the GPU count (equal shards for the gather)"
        from gym_kmanip_amd.dist import shard_range
        lo, hi = shard_range(args.envs_total, world, rank)
        n, off = hi - lo, lo
    else:
        n, off = args.envs_per_gpu, rank * args.envs_per_gpu
    w = Workload(torch, args.env, n, local_rank, rank, off, args.solver, args.solver_iterations, stagger=not args.no_stagger)
    env, cm = w.env, w.cm

    gather = None
    ranks_seen = 1
    gather_depth = None
    if dist is not None:
        with stdout_to_stderr(), Deadline("the rank-count all-reduce (first collective: RCCL creates its communicator here)", args.rendezvous_timeout, rank):
            t = torch.ones(1, device="cuda"); dist.all_reduce(t); ranks_seen = int(t.item())      # RCCL really spans all ranks
        rc = check_ranks_seen(ranks_seen, world, rank)
        if rc:
            return rc
        if not args.no_gather:
            from gym_kmanip_amd.dist import BlockRewardDoneGather, RewardDoneGather
            direct, opted_in = resolve_gather_direct(args.gather_direct, os.environ.get("KMANIP_GATHER_DIRECT"))
            gather_depth = args.gather_depth if args.gather_depth > 0 else (16 if (backend == "nccl" and world > 1 and not args.gather_serial) else 2)

            def make_gather(direct):
                d = direct if direct != "off" else False
                if args.gather_every > 1:
                    return BlockRewardDoneGather(n, world, torch.device("cuda", local_rank), dist, block=args.gather_every,
                                                 force_collective=args.rccl_world1, direct=d)
                return RewardDoneGather(n, world, torch.device("cuda", local_rank), dist, force_collective=args.rccl_world1,
                                        overlap=not args.gather_serial, direct=d, depth=gather_depth)
            try:
                # dist._make_direct moves the ranks in lock-step (agreement before and after the communicator is made), so a failure
                # raises on every rank together; a HANG in ncclCommInitRank / the self-test ends the rank through the deadline
                with stdout_to_stderr(), Deadline("the direct RCCL communicator (ncclCommInitRank + self-test)" if direct != "off"
                                                  else "the (reward, done) gather's set-up", args.rendezvous_timeout, rank):
                    gather = make_gather(direct)
            except Exception as ex:  # noqa: BLE001
                sys.stderr.write("bench.py: rank %d: %s (%s: %s)\n" % (rank, "direct RCCL exchange unavailable" if direct != "off" else
                                                                      "the (reward, done) gather could not be set up", type(ex).__name__, ex))
                if direct == "off" or opted_in == "flag":
                    return 5                 # asked for on the command line: do not measure something else instead
                direct = "off"               # asked for by environment only: every rank falls back to torch.distributed's wrapper
                gather = make_gather(direct)
            args.gather_direct = direct

    # BASELINE config 5: the gripper-cam depth render is bound to the step (kmanip_bind_step_depth): every kmanip_step call
    # ends by rendering the state it produced, on the same stream
    depth_buf = env.bind_step_depth("grip_r", args.depth, args.depth) if args.depth else None

    # *Vision env ids: the camera branch of get_observation (env_sim.py:140-145) -- every camera of the observation space is
    # rendered after the step, on the step's stream, at its reference resolution; the render launches are timed with events
    rgb_bufs = {c: env.render_rgb(c) for c in cm.cameras}

    def one_step():
        if gather is not None:
            gather.before_step()          # the step's stream waits for the exchange of step k-2, which reads the record buffer step k fills
        w.step()
        if rgb_bufs:
            env.render_cameras(out=rgb_bufs)         # all cameras of the observation in one launch (timed by the library: the step's render leg)
        if gather is not None:
            gather.post(env.reward, env.done)                 # (bound below: the step wrote the record, nothing is packed here)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    w.lay_out(args.warmup + args.steps)
    if gather is not None:
        gather.bind(env)          # from here on every step writes its packed (reward, done) record itself: one step, one post
    # N > 1: the steps themselves wait for peers (the per-step exchange, the two barriers, the max over ranks).  A collective that
    # never completes would sit there until the driver's own timeout; the same watchdog bounds the whole region generously
    # (run_timeout_seconds: two minutes plus 10 ms a step unless --run-timeout says otherwise) -- it never fires on a healthy run
    import contextlib
    run_guard = (Deadline("the warm-up and timed steps (a collective that never completed?)", run_timeout_seconds(args), rank)
                 if dist is not None else contextlib.nullcontext())
    with run_guard:
        for _ in range(args.warmup):
            one_step()
        barrier()
        env.enable_timing(timing_every(args.steps))
        t0 = time.perf_counter()
        for _ in range(args.steps):
            one_step()
        if gather is not None:
            gather.wait()
        barrier()
        dt = time.perf_counter() - t0
        ik_ms, dyn_ms, rnd_ms, nt = env.timing_summary()
        env.enable_timing(False)
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())

    if rank == 0:
        version = env.L.kmanip_version().decode()
        total_envs = args.envs_total if args.envs_total else world * n
        total_env_steps = total_envs * args.steps
        bpe = algorithmic_bytes_per_env_step(cm, args.depth, rgb=bool(rgb_bufs))
        # the launch(es) of one step: k_step, plus k_render when the depth image is rendered in the step (config 5) -- the
        # algorithmic bytes of BOTH over the time of BOTH
        bytes_per_launch = bpe * n
        dyn_avg_s = dyn_ms / max(nt, 1) * 1e-3
        step_kernels_s = (dyn_ms + rnd_ms) / max(nt, 1) * 1e-3
        achieved = bytes_per_launch / step_kernels_s / 1e9
        nl = cm.nlink
        kprefix = "void k_step<%d, %d, %d" % (10 if nl <= 10 else 20, 16 if nl <= 10 else 32, 1 if args.solver == "newton" else 0)
        cc = committed_counters(version, kprefix) if (args.env == "KManipSoloArm" and n == 4096 and not args.no_stagger) else \
            {"traffic": None, "traffic_source": "committed counters are for KManipSoloArm @ 4096 envs only", "valu": None, "flops": None}
        headline = args.env == "KManipSoloArm" and n == 4096
        metric = "env steps/sec (whole node), KManipSoloArm @4096 envs, 1/2/4/8 MI355X" if headline else \
            "env steps/sec (whole node), %s @%d envs per GPU, %d MI355X" % (args.env, n, world)
        valu = cc["valu"]
        if cc["flops"] and valu is not None and "stale" not in valu:
            fl = cc["flops"]["fp64_flops_per_launch_active_lanes"]
            valu = dict(valu, fp64_flops_per_env_step=fl / n, achieved_tflops=fl / dyn_avg_s / 1e12,
                        flop_frac=fl / dyn_avg_s / 1e12 / FP64_VECTOR_PEAK_TFLOPS, peak_tflops=FP64_VECTOR_PEAK_TFLOPS,
                        flops_source=cc["flops"]["source"])
        out = {
            "metric": metric,
            "value": total_env_steps / dt, "unit": "env steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if args.envs_total else "weak",
            "vs_baseline": None,          # BASELINE.md holds no published number for this metric (the CPU leg below is a port, not a baseline to divide by)
            "dtype": "f64", "data": "synthetic",
            "timed_window_s": dt,
            "config": {"workload": "%s, %d envs per GPU (%d total), %s, action_space.sample() per step from the Philox stream keyed (seed; env id, episode, step), 64-step episodes with auto-reset, %s"
                                   % (args.env, n, total_envs, ("%dx%d float32 grip_r depth render in the step" % (args.depth, args.depth)) if args.depth else
                                      (("uint8 RGB cameras %s rendered after the step" % "+".join(cm.cameras)) if rgb_bufs else "no cameras"),
                                      "phase-locked envs" if args.no_stagger else "envs desynchronised (episode phase = global env id % 64, one pre-rolled episode)"),
                       "envs_per_gpu": n, "depth_image": ("%dx%d float32 grip_r" % (args.depth, args.depth)) if args.depth else None,
                       "solver": args.solver, "solver_iterations": args.solver_iterations,
                       "sharding": "contiguous env-index blocks, 1 process per GPU",
                       "collective": (("async all_gather of (reward, done) per step" if args.gather_every <= 1 else "async all_gather of (reward, done), %d steps per exchange" % args.gather_every) + ("" if not args.gather_serial else ", step waits for its own exchange") + ("" if (args.gather_every > 1 or gather is None) else ", ring of %d records" % gather.depth) + ("" if args.gather_direct == "off" else ", ncclAllGather issued directly (%s)" % ("the step's stream" if args.gather_direct == "stream" else "side stream"))) if gather is not None else "none",
                       "gather_direct": args.gather_direct if gather is not None else None, "gather_depth": getattr(gather, "depth", None) if gather is not None else None,
                       "gather_every": args.gather_every if gather is not None else None,
                       "kmanip_env": kmanip_env_vars(),       # diagnostic variables that shape the launch (never the results)
                       "rccl_ranks_seen": ranks_seen, "backend": backend if dist is not None else None, "library": version},
            "roofline": {"bound": "hbm", "bound_note": "the contract's two choices are hbm | mfma; this kernel is bound by FP64 VALU issue and dependent latency (see valu), its HBM fraction is small by construction",
                         "kernel": "k_step (before_step decode+IK fused with the 10 physics sub-steps)" + (" + k_render (in-step depth image)" if args.depth else (" + k_render_rgb (camera observations)" if rgb_bufs else "")), "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": cc["traffic"], "traffic_source": cc["traffic_source"],
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "bytes_per_env_step": bpe,
                         "kernel_ms_avg": {"k_step": dyn_ms / max(nt, 1), "k_render": rnd_ms / max(nt, 1), "launch_gap": max(dt / args.steps * 1e3 - (dyn_ms + rnd_ms + ik_ms) / max(nt, 1), 0.0),   # ms per step outside the timed kernels: launch overhead (+ the event records of the sampled steps)
                                           "launches_timed": nt, "timed_every": timing_every(args.steps)},
                         "valu": valu,
                         "note": "latency/FP64-VALU bound by construction (SURVEY 8d): HBM traffic per env-step is ~1.2 KB"},
        }
        if dt < 0.25:
            out["warning"] = "timed window %.3f s < 0.25 s: too short for a stable rate (use --steps >= %d)" % (dt, int(0.3 / (dt / args.steps)) + 1)
        out["secondary_errors"] = 0
        gpu_dead = []      # a HIP / runtime error is sticky: every later GPU measurement would fail the same way

        def secondary(key, fn, into=None, gpu=True):
            """A secondary measurement must never cost the run its one JSON line: a failure is recorded under its key and counted
            in `secondary_errors`; after a HIP / runtime error (as opposed to a Python-level one) no further GPU secondary runs."""
            tgt = out if into is None else into
            if gpu and gpu_dead:
                tgt[key] = {"error": "skipped: an earlier secondary measurement (%s) left the GPU in an error state" % gpu_dead[0]}
                out["secondary_errors"] += 1
                return
            try:
                tgt[key] = fn()
            except Exception as e:  # noqa: BLE001
                tgt[key] = {"error": "%s: %s" % (type(e).__name__, e)}
                out["secondary_errors"] += 1
                txt = "%s %s" % (type(e).__name__, e)
                if gpu and (isinstance(e, (RuntimeError, MemoryError)) or "HIP" in txt or "hip" in txt):
                    try:
                        torch.cuda.synchronize()
                    except Exception:  # noqa: BLE001
                        gpu_dead.append(key)
                    else:
                        if "out of memory" in txt.lower() or "hipError" in txt:
                            gpu_dead.append(key)

        if world == 1 and not args.no_variants:
            torch.cuda.synchronize()
            if not args.depth and not rgb_bufs:
                if args.steps < 512 and not args.no_stagger:
                    # the driver's window (20 steps) holds none of the rare IK crawls (one launch in ~40 runs 2-7x long, DESIGN.md
                    # 3.4): the same handle, still desynchronised, over 512 further steps of the stream
                    def long_window():
                        dtl = w.timed(512, 8)
                        return {"value": n * 512 / dtl, "unit": "env steps/s", "steps": 512, "ms_per_step": dtl / 512 * 1e3,
                                "note": "same handle and stream as `value`, 512 further desynchronised steps: includes the rare IK-crawl launches a 20-step window misses"}
                    secondary("long_window", long_window)
                secondary("seam_variant", lambda: measure_seam(torch, w))
                if args.chunk > 1:
                    secondary("chunked_variant", lambda: measure_chunked(torch, w, args.chunk))
            w.close()
            if headline and args.solver == "newton" and not args.no_stagger:
                secondary("two_handles_variant", lambda: measure_two_handles(torch, args, n, local_rank))
                for name, kw in OTHER_CONFIGS:      # driver-clocked lines for the other BASELINE configs (never part of `value`)
                    secondary(name, lambda kw=kw: measure_config(torch, local_rank=local_rank, **kw))
                    if kw["n"] == 8192 and "error" not in out[name]:   # the same 8192 envs as two independent 4096-env batches in flight
                        secondary("as_two_handles", lambda kw=kw: measure_two_handles(torch, args, 4096, local_rank, steps=64, env_id=kw["env_id"]), into=out[name])
            if not args.no_stagger:
                def phase_locked():
                    r = measure_variant(torch, args, n, local_rank, rank, args.solver, False, 4 * EPISODE, EPISODE)
                    r["note"] = "all envs reset together (what plain auto-reset stepping gives: episodes never end early); whole episodes timed"
                    return r
                secondary("phase_locked", phase_locked)
            if args.solver == "newton":
                secondary("pgs_variant", lambda: measure_variant(torch, args, n, local_rank, rank, "pgs", not args.no_stagger, 32, 8))
        if not args.no_cpu_baseline and world == 1:      # the CPU leg is timed on rank 0 of the 1-GPU run only
            secondary("cpu_baseline", lambda: dict(cpu_baseline(cm, n), solver=args.solver), gpu=False)
        print(json.dumps(out), flush=True)
        if out["secondary_errors"] and os.environ.get("KMANIP_BENCH_STRICT") == "1":
            sys.exit(5)      # opt-in: the JSON line is out; tell a caller that wants it (CI, collect scripts) that part of it failed
    try:
        w.close()
    except Exception:  # noqa: BLE001
        pass
    if dist is not None:
        # the measurement is out: a teardown that hangs must not cost the run its result (exit 0 from the watchdog)
        with Deadline("the teardown barrier / destroy_process_group", args.rendezvous_timeout, rank, exit_code=0):
            dist.barrier()
            if gather is not None:
                gather.close()
            dist.destroy_process_group()
    return 0


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    global TIME_EVERY
    TIME_EVERY = max(0, int(args.time_every))
    if args.gpus > 1 and "RANK" not in os.environ:
        return spawn_ranks(args, argv)          # nothing above this line touches the GPU
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
