"""Diagnostic (GPU box): long random-action run at BASELINE sizes -- finiteness, divergence flags, determinism."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from gym_kmanip_amd import env_hip
from gym_kmanip_amd.model import KM_DONE_DIVERGED
for env_id, n, steps in [("KManipSoloArm", 4096, 640), ("KManipDualArm", 4096, 200), ("KManipTorso", 4096, 200), ("KManipSoloArmQPos", 2048, 200)]:
    a = env_hip.make(env_id, num_envs=n, seed=7); b = env_hip.make(env_id, num_envs=n, seed=7)
    a.k_reset(); b.k_reset()
    gen = torch.Generator(device="cuda"); gen.manual_seed(1)
    ndiv = 0; rmin, rmax = 1e9, -1e9
    for k in range(steps):
        act = torch.rand((n, a.cm.act_dim), generator=gen, device="cuda") * 2 - 1
        a.step_flat(act); b.step_flat(act)
        ndiv += int(((a.done & KM_DONE_DIVERGED) != 0).sum())
        assert torch.isfinite(a.obs).all() and torch.isfinite(a.reward).all(), (env_id, k)
        rmin = min(rmin, float(a.reward.min())); rmax = max(rmax, float(a.reward.max()))
    same = all(np.array_equal(x, y) for x, y in zip(a.get_state(), b.get_state()))
    mask, nfev, st = a.get_diag()
    print(env_id, "steps", steps, "diverged env-steps", ndiv, "reward range %.3f..%.3f" % (rmin, rmax), "deterministic", same,
          "nfev max", int(nfev.max()), "contact bits seen %#x" % int(np.bitwise_or.reduce(mask)))
    a.k_close(); b.k_close()
