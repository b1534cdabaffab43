import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from gym_kmanip_amd import env_hip
n = 4096
a = env_hip.make("KManipSoloArm", num_envs=n, seed=3)
c = env_hip.make("KManipSoloArm", num_envs=512, seed=3, env_id_offset=1024)
gen = torch.Generator(device="cuda"); gen.manual_seed(0)
a.k_reset(); c.k_reset()
for k in range(12):
    act = torch.rand((n, 7), generator=gen, device="cuda") * 2 - 1
    a.step_flat(act); c.step_flat(act[1024:1536].contiguous())
    sa, sc = a.get_state(), c.get_state()
    d = [np.abs(x[1024:1536] - z).max() for x, z in zip(sa[:4], sc[:4])]
    bad = np.where(np.abs(sa[0][1024:1536] - sc[0]).max(axis=1) > 0)[0]
    print(k, ["%.2e" % v for v in d], len(bad), bad[:6], a.get_diag()[0][1024:1536][bad[:6]] if len(bad) else "")
