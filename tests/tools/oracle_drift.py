"""Diagnostic (GPU box): per-control-step max |qpos_gpu - qpos_oracle| for a library given by KMANIP_LIB."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from gym_kmanip_amd.model import compile_model
from gym_kmanip_amd import env_hip
from oracle.oracle import Oracle
env_id = sys.argv[1] if len(sys.argv) > 1 else "KManipSoloArmQPos"
cm = compile_model(env_id)
n = 64
dev = env_hip.KManipEnvHip(cm, n, 0, seed=5); orc = Oracle(cm, n, seed=5)
dev.k_reset(); orc.reset()
rng = np.random.default_rng(1)
out = []
for k in range(12):
    act = rng.uniform(-1, 1, (n, cm.act_dim)).astype(np.float32)
    dev.step_flat(torch.from_numpy(act).cuda()); orc.step(act, nthreads=8)
    out.append(np.abs(np.asarray(dev.get_state()[0]) - np.asarray(orc.get_state()[0])).max())
print(os.environ.get("KMANIP_LIB", "default"), " ".join("%.1e" % x for x in out))
