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


def test_shipped_step_kernels_scratch_and_lds():
    """The shipped library's k_step variants (parsed from the built .so: no GPU): every per-step variant runs without scratch and a
    four-env single-arm workgroup's LDS lets four workgroups share a CU (one wave per SIMD: DESIGN.md 2, 3.2); the chunk variants are
    pinned at what round 6 left (two-arm Newton: 128 B -- the one shipped kernel that spills)."""
    import io
    import contextlib
    import re
    import kernel_resources as kr
    so = os.path.join(ROOT, "gym_kmanip_amd", "libkmanip_hip.so")
    if not os.path.exists(so):
        import pytest
        pytest.skip("library not built")
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        kr.main(so, "k_stepILi")
    rows = {}
    for ln in buf.getvalue().splitlines():
        m = re.match(r"_Z6k_stepILi(\d+)ELi(\d+)ELi(\d)ELi(\d)ELb(\d)E\S*\s+vgpr (\d+) agpr (\d+) sgpr \d+ scratch (\d+) lds (\d+)", ln)
        assert m, ln
        nl, g, solver, epb, chunk, vgpr, agpr, scratch, lds = map(int, m.groups())
        rows[(nl, solver, epb, chunk)] = (vgpr, scratch, lds)
    assert len(rows) == 14
    for (nl, solver, epb, chunk), (vgpr, scratch, lds) in rows.items():
        assert vgpr <= 512
        if not chunk and solver == 1:
            assert scratch == 0, (nl, solver, epb, scratch)
        if not chunk and solver == 0:      # (the PGS two-env single-arm variant reserves 36 B its ISA never touches: no scratch_ instruction in it)
            assert scratch <= 64, (nl, solver, epb, scratch)
    assert rows[(20, 0, 2, 1)][1] == 0 and rows[(10, 1, 4, 1)][1] == 0 and rows[(10, 0, 4, 1)][1] == 0
    assert rows[(20, 1, 2, 1)][1] <= 128
    assert 4 * rows[(10, 1, 4, 0)][2] <= 160 * 1024            # four four-env workgroups per CU


def test_asm_phase_mix_charges_an_instruction_to_the_phase_that_called_it(tmp_path):
    """tools/asm_phase_mix.py (profiles/r06_valu_mix.txt): an instruction of an inlined helper belongs to the PHASE at the bottom of its
    inlined-at chain -- a miniature source + listing with the real tool."""
    src = tmp_path / "kmanip_dyn.hip"
    src.write_text("\n".join([
        "__device__ __forceinline__ void helper(double& x) {",                        # 1
        "  x = x * 2;",                                                               # 2
        "}",                                                                          # 3
        "__device__ __forceinline__ void step1_products(int s) {",                    # 4
        "  fk_parallel<NL, G>(w, lm, sub);",                                          # 5
        "  build_constraints_newton<NL, G>(w, lm, m, sub, cr, invm);",                # 6
        "}",                                                                          # 7
        "__global__ __launch_bounds__(64) void k_step(int n) {",                      # 8
        "  coop_before_step<7>(dm, arm, c, arow, io, &pf);",                          # 9
        "  step1_products<NL, G, SOLVER>(w, lm, m, subv, cr, invm, pf);",             # 10
        "  integrate<NL, G>(w, m, sub, a);",                                          # 11
        "}"]) + "\n")
    (tmp_path / "kmanip_ik_coop.hpp").write_text("\n".join([
        "__device__ int coop_trf(int x) {",          # 1
        "  setup();",                                # 2
        "  for (;;) {",                              # 3
        "    outer();",                              # 4
        "    while (actual <= 0 && nfev < max_nfev) {",   # 5
        "      trial();",                            # 6
        "      P.pf->ph(37);",                       # 7
        "    }",
        "  }",
        "}"]) + "\n")
    lst = tmp_path / "k.s"
    lst.write_text("\n".join([
        "_Z6k_stepTEST:",
        "\t.loc\t1 2 3 ; kmanip_dyn.hip:2:3 @[ kmanip_dyn.hip:5:3 @[ kmanip_dyn.hip:10:3 ] ]",      # helper <- fk <- step1 <- k_step
        "\tv_mul_f64 v[0:1], v[0:1], 2.0",
        "\tv_mov_b32_dpp v2, v0 row_mirror row_mask:0xf bank_mask:0xf",
        "\t.loc\t1 2 3 ; kmanip_dyn.hip:2:3 @[ kmanip_dyn.hip:6:3 @[ kmanip_dyn.hip:10:3 ] ]",      # helper <- build_constraints
        "\tv_cndmask_b32_e32 v3, v1, v2, vcc",
        "\tds_read_b64 v[4:5], v6",
        "\ts_waitcnt lgkmcnt(0)",
        "\t.loc\t1 2 3 ; kmanip_dyn.hip:2:3 @[ kmanip_dyn.hip:11:3 ]",                              # helper <- integrate
        "\tv_add_f64 v[0:1], v[0:1], v[4:5]",
        "\t.loc\t2 6 7 ; ./kmanip_ik_coop.hpp:6:7 @[ kmanip_dyn.hip:9:3 ]",                          # the IK's trial point
        "\tv_fma_f64 v[0:1], v[0:1], v[2:3], v[4:5]",
        "\t.loc\t2 2 3 ; ./kmanip_ik_coop.hpp:2:3 @[ kmanip_dyn.hip:9:3 ]",                          # the IK's set-up
        "\tv_accvgpr_read_b32 v7, a3",
        "other_kernel:",
        "\tv_mul_f64 v[0:1], v[0:1], 2.0"]) + "\n")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "asm_phase_mix.py"), str(lst), "k_stepTEST", "--src=%s" % src,
                          "--weights=fk=10,IK: per trial=9"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    rows = {}
    for ln in out.stdout.splitlines():
        if "|" in ln and not ln.startswith(("phase", "TOTAL", "SUM")):
            name = ln.split("|")[0].rsplit(None, 1)[0].strip()
            rows.setdefault(name, ln)
    assert set(rows) >= {"fk", "build_constraints", "integrate", "IK: per trial point", "IK: trf set-up"}, out.stdout
    cols = lambda name: [int(x) for x in rows[name].split("|")[1].split()]
    # classes: f64 mov dppmov sel agpr lane cmp valu lds vmem salu nop wait br
    assert cols("fk")[:3] == [1, 0, 1] and cols("build_constraints")[3] == 1 and cols("build_constraints")[8] == 1 and cols("build_constraints")[12] == 1
    assert cols("integrate")[0] == 1 and cols("IK: per trial point")[0] == 1 and cols("IK: trf set-up")[4] == 1
    assert "TOTAL" in out.stdout and "other_kernel" not in out.stdout
    assert "VALU instructions per wave and control step, estimated: 29" in out.stdout        # fk (2 VALU) x 10 + trial (1) x 9
