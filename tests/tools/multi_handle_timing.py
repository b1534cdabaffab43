import sys, os, time
sys.path.insert(0, '.')
import numpy as np, torch
from gym_kmanip_amd import env_hip
"""Diagnostic (GPU box): several independent handles stepped one kmanip_step per call on their own streams -- what the GPU sustains
when more than one batch is in flight.  python tests/tools/multi_handle_timing.py [env id]"""
ENV = sys.argv[1] if len(sys.argv) > 1 else "KManipSoloArm"
def mk(n, off):
    e = env_hip.make(ENV, num_envs=n, seed=0, env_id_offset=off)
    e.k_reset(); e.set_state(step=((off + np.arange(n)) % 64).astype(np.int32))
    return e
def run(handles, steps, warm=70):
    streams = [torch.cuda.Stream() for _ in handles]
    banks = []
    for e in handles:
        b = torch.empty((warm + steps, e.num_envs, e.cm.act_dim), dtype=torch.float32, device="cuda")
        for k in range(warm + steps): e.sample_action(b[k], ahead=k)
        banks.append(b)
    imgs = [e.render_cameras() for e in handles]               # (*Vision ids: the camera observations after every step, same stream)
    torch.cuda.synchronize()
    def one(e, s, b, im, k):
        with torch.cuda.stream(s):
            e.step_flat(b[k])
            if im: e.render_cameras(out=im)
    for k in range(warm):
        for e, s, b, im in zip(handles, streams, banks, imgs): one(e, s, b, im, k)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(warm, warm + steps):
        for e, s, b, im in zip(handles, streams, banks, imgs): one(e, s, b, im, k)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    return sum(e.num_envs for e in handles) * steps / dt, dt / steps * 1e3
CFGS = ([4096], [4096, 4096], [2048, 2048], [4096, 4096, 4096, 4096], [1024] * 4) if ENV == "KManipSoloArm" else (
    ([2048], [2048, 2048], [1024, 1024], [2048] * 4) if "Vision" in ENV else ([8192], [4096, 4096], [2048] * 4, [8192, 8192]))
for cfg in CFGS:
    hs = [mk(n, sum(cfg[:i])) for i, n in enumerate(cfg)]
    v, ms = run(hs, 256)
    print(ENV, "handles %s on their own streams: %.3f M env steps/s, %.4f ms per round of steps" % (cfg, v / 1e6, ms))
    for e in hs: e.k_close()
