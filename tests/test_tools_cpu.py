"""The CPU-side analysis tools behind profiles/r05_scaling_model.txt and r05_render_behind_trace.txt, on synthetic inputs."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_scaling_model_limits():
    """No jitter: every rendezvous is free; with outliers, a deeper ring / a longer block recovers what a per-step rendezvous loses."""
    import scaling_model as sm
    flat = np.full(256, 0.6)
    assert abs(sm.model(flat, 8, 1) - 1.0) < 1e-12 and abs(sm.model_ring(flat, 8, 2, steps=2000) - 1.0) < 1e-9
    rng = np.random.default_rng(0)
    t = np.where(rng.random(2048) < 0.02, 3.0, 0.57)
    e1, e64 = sm.model(t, 8, 1, blocks=4000), sm.model(t, 8, 64, blocks=2000)
    r1, r2, r16 = (sm.model_ring(t, 8, d, steps=20000) for d in (1, 2, 16))
    assert e1 < 0.7 < 0.85 < e64 <= 1.0 + 1e-9
    assert r1 < r2 < r16 and r16 > 0.93 and abs(r1 - e1) < 0.03            # depth 1 IS the per-step rendezvous
    assert sm.model(t, 2, 1, blocks=4000) > e1                              # fewer ranks wait less


def test_trace_overlap_counts_what_runs_concurrently(tmp_path):
    rows = ['"Kind","Kernel_Name","Start_Timestamp","End_Timestamp"']
    # first half: each render inside a k_step; second half: renders after their k_step
    t = 0
    for k in range(4):
        rows.append('"KERNEL_DISPATCH","void k_step<10, 16, 1, 2, false>(...)",%d,%d' % (t, t + 1000))
        rows.append('"KERNEL_DISPATCH","k_render_rgb(...)",%d,%d' % (t + 200, t + 700))
        t += 1000
    for k in range(4):
        rows.append('"KERNEL_DISPATCH","void k_step<10, 16, 1, 2, false>(...)",%d,%d' % (t, t + 1000))
        rows.append('"KERNEL_DISPATCH","k_render_rgb(...)",%d,%d' % (t + 1000, t + 1400))
        t += 1400
    p = tmp_path / "trace.csv"
    p.write_text("\n".join(rows) + "\n")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "trace_overlap.py"), str(p)], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.splitlines()
    assert "100.0 %" in lines[0] and ", 0.0 %" in lines[1] and "k_step dispatches: 8" in lines[2], out.stdout
