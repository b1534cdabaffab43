#!/usr/bin/env python3
"""Diagnostic (GPU box): per-step GPU-vs-oracle differences of the step-parity scenario, to tell an operation-order drift
from a semantic mismatch.  Usage: python tests/tools/parity_drift.py KManipDualArm pgs 16 66"""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import torch
from gym_kmanip_amd import env_hip
from gym_kmanip_amd.model import compile_model
from oracle.oracle import Oracle

env, solver, n, steps = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
cm = compile_model(env, auto_reset=True, solver=solver)
dev = env_hip.KManipEnvHip(cm, num_envs=n, seed=5, env_id_offset=7); orc = Oracle(cm, n, seed=5, env_id_offset=7)
dev.k_reset(); orc.reset()
rng = np.random.default_rng(42)
for k in range(steps):
    act = rng.uniform(-1, 1, (n, cm.act_dim)).astype(np.float32)
    dev.step_flat(torch.from_numpy(act).cuda()); orc.step(act)
    sg, so = dev.get_state(), orc.get_state()
    bad = np.argwhere(sg[2] != so[2])
    print("step %2d dq %.2e dv %.2e dwarm %.2e ctrl mismatches %d %s" % (
        k, np.abs(sg[0] - so[0]).max(), np.abs(sg[1] - so[1]).max(), np.abs(sg[3] - so[3]).max(), len(bad),
        [(int(e), int(j), float(sg[2][e, j] - so[2][e, j])) for e, j in bad[:3]]))
