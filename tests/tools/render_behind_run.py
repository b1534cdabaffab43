"""Workload for a rocprofv3 --kernel-trace run: KManipSoloArmVision @ 2048 envs, cameras rendered behind the steps
(pipeline.RenderBehind), then the same steps with the cameras in sequence.   python3 tests/tools/render_behind_run.py [steps]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from gym_kmanip_amd import env_hip
from gym_kmanip_amd.pipeline import RenderBehind
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = 2048
e = env_hip.make("KManipSoloArmVision", num_envs=n, seed=0)
e.k_reset(); e.set_state(step=(np.arange(n) % 64).astype(np.int32))
acts = [e.sample_action(ahead=k).clone() for k in range(8)]
for k in range(70):
    e.step_flat(acts[k & 7])
rb = RenderBehind(e)
torch.cuda.synchronize()
for k in range(steps):
    e.step_flat(acts[k & 7]); rb.after_step()
torch.cuda.synchronize()
bufs = e.render_cameras()
torch.cuda.synchronize()
for k in range(steps):
    e.step_flat(acts[k & 7]); e.render_cameras(out=bufs)
torch.cuda.synchronize()
