"""CPU-side checks of the C-ABI boundary: the library loads, exports every symbol include/kmanip.h
declares, agrees on the descriptor layout, and FAILS LOUDLY without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from gym_kmanip_amd import lib as klib
from gym_kmanip_amd.model import ENV_SPECS, KModelDesc, compile_model


@pytest.fixture(scope="module")
def L():
    if not os.path.exists(klib.LIB_PATH):
        klib.build()
    return klib.load()


def test_header_symbols_exported(L):
    hdr = open(os.path.join(ROOT, "include", "kmanip.h")).read()
    declared = set(re.findall(r"\b(kmanip_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(klib.EXPORTS), declared ^ set(klib.EXPORTS)
    for name in declared:
        assert hasattr(L, name), name


def _declared(header, macros=()):
    """Function names a header declares, after the C preprocessor (so #ifdef'd diagnostics count only when their macro is set)."""
    import subprocess
    out = subprocess.check_output(["gcc", "-E", "-P", "-x", "c"] + ["-D" + m for m in macros]
                                  + [os.path.join(ROOT, "include", header)], text=True)
    return set(re.findall(r"\b(kmanip_[a-z_0-9]+)\s*\(", out))


def test_exported_symbols_are_exactly_the_declared_ones(L):
    """-fvisibility=hidden + csrc/exports.map: the dynamic symbol table holds the functions of include/kmanip.h and
    include/kmanip_debug.h and NOTHING else (no C++ launch helpers, no kernel stubs, no libstdc++ instantiations)."""
    import shutil
    import subprocess
    if shutil.which("nm") is None or shutil.which("gcc") is None:
        pytest.skip("no binutils / gcc")
    rows = [l.split() for l in subprocess.check_output(["nm", "-D", "--defined-only", klib.LIB_PATH], text=True).splitlines()]
    exported = {r[-1] for r in rows}
    assert all(r[-2] == "T" for r in rows), [r for r in rows if r[-2] != "T"]
    assert _declared("kmanip.h") == set(klib.EXPORTS)
    assert _declared("kmanip_debug.h") == set(klib.EXPORTS) | set(klib.DEBUG_EXPORTS)
    assert exported == set(klib.EXPORTS) | set(klib.DEBUG_EXPORTS), exported ^ (set(klib.EXPORTS) | set(klib.DEBUG_EXPORTS))
    prof = os.path.join(os.path.dirname(klib.LIB_PATH), "libkmanip_hip_prof.so")
    if os.path.exists(prof):      # the diagnostic build: the same rule with KM_PROFILE's extra declarations
        exp = {l.split()[-1] for l in subprocess.check_output(["nm", "-D", "--defined-only", prof], text=True).splitlines()}
        assert exp == _declared("kmanip_debug.h", ["KM_PROFILE"]), exp ^ _declared("kmanip_debug.h", ["KM_PROFILE"])


def test_desc_layout_matches(L):
    assert L.kmanip_model_desc_size() == C.sizeof(KModelDesc)
    assert b"gfx950" in L.kmanip_version()


@pytest.mark.parametrize("env_id", sorted(ENV_SPECS))
def test_all_env_ids_compile(env_id):
    cm = compile_model(env_id)
    spec = cm.spec
    # action / observation layout follows the reference Dict-space order (env_base.py:115-190)
    assert cm.obs_dim == 2 * cm.nlink + 7
    width = sum((sl.stop - sl.start) for sl in cm.act_slices.values())
    assert width == cm.act_dim and set(cm.act_slices) == set(spec.act_list)
    exp = {"KManipSoloArm": 7, "KManipSoloArmVision": 7, "KManipSoloArmQPos": 8, "KManipDualArm": 14,
           "KManipDualArmVision": 14, "KManipDualArmQPos": 16, "KManipTorso": 14, "KManipTorsoVision": 14}[env_id]
    assert cm.act_dim == exp
    assert cm.desc.n_sub_steps == 10 and cm.desc.max_episode_steps == 64


def test_no_gpu_fails_loudly(L):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    cm = compile_model("KManipSoloArm")
    h = C.c_void_p()
    rc = L.kmanip_create(C.byref(cm.desc), 4, 0, C.c_uint64(0), C.c_int64(0), C.byref(h))
    assert rc != 0 and not h.value
    assert b"no CPU path" in L.kmanip_last_error(None) or b"HIP" in L.kmanip_last_error(None)
    from gym_kmanip_amd import env_hip
    with pytest.raises(klib.KManipError):
        env_hip.make("KManipSoloArm", num_envs=2)


def test_bad_model_rejected(L):
    cm = compile_model("KManipSoloArm")
    d = KModelDesc.from_buffer_copy(cm.desc)
    d.nlink = 13
    h = C.c_void_p()
    assert L.kmanip_create(C.byref(d), 4, 0, C.c_uint64(0), C.c_int64(0), C.byref(h)) != 0
    assert b"nlink" in L.kmanip_last_error(None)


def test_header_is_plain_c():
    """The drop-in boundary is a C ABI: include/kmanip.h must compile as C99 (and as C++) on its own."""
    import shutil
    import subprocess
    hdr = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "include", "kmanip.h")
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    for h in (hdr, hdr.replace("kmanip.h", "kmanip_debug.h")):
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", h])
        subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-x", "c++", h])
