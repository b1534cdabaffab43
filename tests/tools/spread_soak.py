"""Diagnostic (GPU box): a default single-arm handle (SPREAD) against one with the identity map over thousands of steps, bit for bit.
   python tests/tools/spread_soak.py"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from gym_kmanip_amd import env_hip
n, steps = 4096, 4000
a = env_hip.make("KManipSoloArm", num_envs=n, seed=7)
os.environ["KMANIP_SPREAD"] = "0"
b = env_hip.make("KManipSoloArm", num_envs=n, seed=7)
del os.environ["KMANIP_SPREAD"]
a.k_reset(); b.k_reset()
ph = (np.arange(n) % 64).astype(np.int32); a.set_state(step=ph); b.set_state(step=ph)
slot = np.zeros(a.L.kmanip_dbg_wave_slots(a.h), dtype=np.int32); moved = 0
assert len(slot) == n
for k in range(steps):
    act = a.sample_action().clone()
    a.step_flat(act); b.step_flat(act)
    if k % 500 == 499 or k == steps - 1:
        assert torch.equal(a.obs, b.obs) and torch.equal(a.reward, b.reward) and torch.equal(a.done, b.done), k
        assert a.L.kmanip_dbg_wave_clocks(a.h, None, slot.ctypes.data_as(C.POINTER(C.c_int32)), None) == 0
        assert np.array_equal(np.sort(slot.reshape(-1, 64), axis=1), np.arange(n).reshape(-1, 64)), k
        moved += int((slot != np.arange(n)).sum())
assert all(np.array_equal(x, y) for x, y in zip(a.get_state(), b.get_state()))
print("spread against the identity map: %d envs x %d steps bit for bit (obs, reward, done at every 500th step; full state at the end); %d slot moves seen at the checkpoints" % (n, steps, moved))
