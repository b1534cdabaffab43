"""CPU check (no GPU needed: hipcc cross-compiles) of the hazard class DESIGN.md 3.4 records: the kernels fold DPP row broadcasts
into FMAs with inline asm, LLVM's hazard recogniser does not look inside inline asm, and a DPP read of a VGPR less than two wait
states after a VALU write of it returns stale data on some lanes.  tools/check_dpp_hazard.py scans the FINAL ISA of the shipped
kernel variants for that pattern."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gym_kmanip_amd", "csrc")
sys.path.insert(0, os.path.join(ROOT, "tools"))
HIPCC = "/opt/rocm/bin/hipcc"


def test_checker_finds_a_planted_hazard(tmp_path):
    import check_dpp_hazard as chk
    s = tmp_path / "t.s"
    s.write_text("k:\n"
                 "\tv_fmac_f64_dpp v[2:3], -v[4:5], v[6:7] row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                 "\tv_mov_b64_dpp v[8:9], v[2:3] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"      # 0 wait states: hazard
                 "\tv_add_f64 v[10:11], v[0:1], v[2:3]\n\ts_nop 0\n"
                 "\tv_mov_b32_dpp v12, v10 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n"               # 1 wait state: hazard
                 "\tv_mul_f64 v[20:21], v[0:1], v[2:3]\n\ts_nop 1\n"
                 "\tv_fmac_f64_dpp v[2:3], v[20:21], v[6:7] row_newbcast:3 row_mask:0xf bank_mask:0xf\n")       # 2 wait states: fine
    assert chk.check(str(s)) == 2
    s2 = tmp_path / "u.s"
    s2.write_text("k2:\n"
                  "\tv_permlane16_swap_b32 v4, v5\n"
                  "\tv_mov_b32_dpp v9, v5 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n"                 # the swap wrote v5 too: hazard
                  "\tv_add_f64 v[10:11], v[0:1], v[2:3]\n"
                  ".LBB0_1:\n"                                                                                 # fall-through label: window carried
                  "\tv_mov_b64_dpp v[12:13], v[10:11] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"  # hazard across the label
                  "\ts_nop 1\n\tv_mul_f64 v[30:31], v[0:1], v[2:3]\n"
                  "\tv_add_f64 v[10:11], v[0:1], v[2:3]\n"
                  "\ts_cbranch_vccnz .LBB0_1\n"                                                               # back edge: head reads v[10:11] 1 wait state after
                  "\ts_branch .LBB0_3\n"
                  ".LBB0_2:\n"                                                                                 # behind an unconditional branch: window reset
                  "\tv_mov_b64_dpp v[12:13], v[10:11] row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                  ".LBB0_3:\n\ts_endpgm\n")
    assert chk.check(str(s2)) == 3


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_shipped_kernels_have_no_dpp_read_after_write_hazard(tmp_path):
    import check_dpp_hazard as chk
    flags = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=fast-honor-pragmas", "-S", "--cuda-device-only"]
    # every kmanip_dyn variant the Makefile links (DYN_VARIANTS) + the stand-alone IK object
    jobs = [("kmanip_dyn.hip", ["-DKM_VAR_NL=%d" % nl, "-DKM_VAR_G=%d" % g, "-DKM_VAR_SOLVER=%d" % sv], "dyn_%d_%d_%d.s" % (nl, g, sv))
            for nl, g in ((10, 16), (20, 32)) for sv in (1, 0)] + [("kmanip_ik_coop.hip", [], "ik_coop.s")]
    mk = open(os.path.join(CSRC, "Makefile")).read()
    assert "DYN_VARIANTS := 10_16_1 10_16_0 20_32_1 20_32_0" in mk          # (keep this list in step with the Makefile)

    def build(job):
        src, defs, out = job
        path = str(tmp_path / out)
        subprocess.check_call([HIPCC] + flags + defs + [os.path.join(CSRC, src), "-o", path], stderr=subprocess.DEVNULL)
        return path
    with ThreadPoolExecutor(5) as ex:
        listings = list(ex.map(build, jobs))
    for p in listings:
        assert chk.check(p) == 0, p
