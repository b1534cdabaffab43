"""Diagnostic (GPU box): k_render_depth against the oracle's ray caster over many states -- how many pixels differ by more than the
1e-6 m bar (rays grazing a silhouette land on the other side of it), per image and overall, and the largest difference elsewhere.
   python tests/tools/depth_parity_soak.py [envs] [steps]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from gym_kmanip_amd import env_hip
from oracle.oracle import Oracle
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
for env_id, cams in (("KManipSoloArm", ["grip_r"]), ("KManipTorso", ["grip_r", "grip_l"])):
    e = env_hip.make(env_id, num_envs=n, seed=11)
    e.k_reset()
    orc = Oracle(e.cm, 1, seed=11)
    worst_frac, nbad, npx, worst_ok = 0.0, 0, 0, 0.0
    for k in range(steps):
        e.step_flat(e.sample_action())
        if k % 5 != 4:
            continue
        qpos = e.get_state()[0]
        for ci, cam in enumerate(cams):
            img = e.render_depth(cam, 64, 64).cpu().numpy()
            for i in range(n):
                ref = orc.render_depth(qpos[i], ci, 64, 64)
                d = np.abs(img[i] - ref)
                bad = d > 1e-6
                worst_frac = max(worst_frac, bad.mean()); nbad += int(bad.sum()); npx += bad.size
                worst_ok = max(worst_ok, float(d[~bad].max()))
    print("%s: %d images of 64 x 64: pixels off by more than 1e-6 m: %d of %d (%.2e); worst image %.2e of its pixels (test bar 5e-4); largest difference among the others %.2e m" % (
        env_id, npx // 4096, nbad, npx, nbad / npx, worst_frac, worst_ok))
    e.k_close()
