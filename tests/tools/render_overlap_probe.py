"""Diagnostic (GPU box): does the RGB render of step k overlap the physics of step k + 1 when they are issued on two streams?
TIMING ONLY -- the two streams are not ordered against each other here (the render reads whatever state it finds).
   [KMANIP_EPB=4] python tests/tools/render_overlap_probe.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from gym_kmanip_amd import env_hip
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = 2048
e = env_hip.make("KManipSoloArmVision", num_envs=n, seed=0)
e.k_reset(); e.set_state(step=(np.arange(n) % 64).astype(np.int32))
acts = [e.sample_action(ahead=k).clone() for k in range(8)]
for k in range(70):
    e.step_flat(acts[k & 7])
bufs = e.render_cameras()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

def run(mode):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        if mode == "serial":
            e.step_flat(acts[k & 7]); e.render_cameras(out=bufs)
        elif mode == "step":
            e.step_flat(acts[k & 7])
        elif mode == "render":
            e.render_cameras(out=bufs)
        else:
            with torch.cuda.stream(sa):
                e.step_flat(acts[k & 7])
            with torch.cuda.stream(sb):
                e.render_cameras(out=bufs)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3

for mode in ("serial", "step", "render", "two streams", "serial", "two streams"):
    run(mode); ms = run(mode)
    print("%-12s %.4f ms per step  (%.2f M env steps/s)" % (mode, ms, n / ms / 1e3))
