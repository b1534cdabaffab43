"""Diagnostic (GPU box): k_render_rgb / k_render_depth launch times by image size -> fixed per-workgroup cost vs per-pixel cost."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from gym_kmanip_amd import env_hip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
e = env_hip.make("KManipSoloArmVision", num_envs=n, seed=0)
e.k_reset()
a = e.sample_action()
for k in range(20):
    e.step_flat(e.sample_action(a))
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for cam, h, w in (("head", 480, 640), ("head", 240, 320), ("head", 48, 64), ("head", 8, 8), ("grip_r", 40, 60), ("top", 480, 640)):
    buf = torch.empty((n, h, w, 3), dtype=torch.uint8, device="cuda")
    ms = t(lambda: e.render_rgb(cam, h, w, out=buf))
    print("rgb   %-6s %3dx%3d  %.4f ms  %.2f TB/s" % (cam, h, w, ms, n * h * w * 3 / ms / 1e9))
for h, w in ((64, 64), (8, 8)):
    buf = torch.empty((n, h, w), dtype=torch.float32, device="cuda")
    ms = t(lambda: e.render_depth("grip_r", h, w, out=buf))
    print("depth grip_r %3dx%3d  %.4f ms  %.2f TB/s" % (h, w, ms, n * h * w * 4 / ms / 1e9))
bufs = e.render_cameras()
ms = t(lambda: e.render_cameras(out=bufs))
print("rgb   all cameras of the observation (%s) in one launch  %.4f ms  %.2f TB/s" % ("+".join(bufs), ms, sum(b.numel() for b in bufs.values()) / ms / 1e9))
# the ceiling for a write-only stream on this box: a plain fill of the same number of bytes (torch's fill kernel, dword stores)
big = torch.empty(sum(b.numel() for b in bufs.values()) // 4, dtype=torch.int32, device="cuda")
ms = t(lambda: big.fill_(7))
print("fill  %d MB with one value (torch fill_)  %.4f ms  %.2f TB/s" % (big.numel() * 4 // 2**20, ms, big.numel() * 4 / ms / 1e9))
src = torch.empty_like(big)
ms = t(lambda: big.copy_(src))
print("copy  the same bytes device to device (read + write)  %.4f ms  %.2f TB/s written, %.2f TB/s moved" % (ms, big.numel() * 4 / ms / 1e9, 2 * big.numel() * 4 / ms / 1e9))
