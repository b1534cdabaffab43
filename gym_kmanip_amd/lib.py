"""ctypes binding of libkmanip_hip.so (C ABI: include/kmanip.h).

The north star names cffi for this layer; cffi is not installed in the build image
(`import cffi` -> ModuleNotFoundError), so the stdlib `ctypes` loader is used.
There is NO CPU fallback: a missing library or a missing GPU raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

from .model import KModelDesc

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KMANIP_LIB") or os.path.join(_HERE, "libkmanip_hip.so")   # KMANIP_LIB: diagnostic builds only
_lib = None

# every symbol include/kmanip.h declares (tests check the library exports all of them)
EXPORTS = [
    "kmanip_model_desc_size", "kmanip_create", "kmanip_reset", "kmanip_step", "kmanip_step_chunk", "kmanip_get_state",
    "kmanip_set_state", "kmanip_get_episode", "kmanip_set_episode", "kmanip_get_counters", "kmanip_bind_sim_time", "kmanip_bind_reward_done_record", "kmanip_select_reward_done_record", "kmanip_observe", "kmanip_set_seed", "kmanip_get_diag", "kmanip_timing_summary", "kmanip_enable_timing", "kmanip_ik", "kmanip_ik_eval",
    "kmanip_render_depth", "kmanip_render_rgb", "kmanip_render_rgb_multi", "kmanip_snapshot_render_state", "kmanip_set_render_source", "kmanip_bind_step_depth", "kmanip_scripted_action", "kmanip_sample_action", "kmanip_num_envs", "kmanip_last_error", "kmanip_version", "kmanip_destroy",
]


# include/kmanip_debug.h: diagnostics, not part of the boundary (product build; the -DKM_PROFILE build adds kmanip_dbg_prof*)
DEBUG_EXPORTS = ["kmanip_dbg_wave_clocks", "kmanip_dbg_wave_slots"]


class KManipError(RuntimeError):
    pass


def build(force: bool = False) -> str:
    """Compile the HIP extension in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    src_dir = os.path.join(_HERE, "csrc")
    cmd = ["make", "-C", src_dir, "-s", "-j4"] + (["-B"] if force else [])
    subprocess.check_call(cmd)
    return LIB_PATH


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise KManipError(
            "libkmanip_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'`; "
            "there is no CPU fallback for the simulation path." % LIB_PATH)
    # torch bundles its own HIP runtime (torch/lib/libamdhip64.so); this library links against the same SONAME.  Whichever is
    # loaded FIRST serves the whole process, and a process that loaded /opt/rocm's copy first and torch's afterwards ends up with
    # two runtimes, the second of which sees no device ("no ROCm-capable device is detected" from kmanip_create after
    # `build(); smoke()` in one process).  The host side of this package is torch-based anyway: load torch's runtime first.
    try:
        import torch  # noqa: F401
    except Exception:  # noqa: BLE001  (a torch-less C caller binds the ABI directly)
        pass
    lib = C.CDLL(LIB_PATH)
    vp, i32p, u8p, f64p, f32p = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_uint8), C.POINTER(C.c_double), C.POINTER(C.c_float)
    lib.kmanip_model_desc_size.restype = C.c_int
    lib.kmanip_create.argtypes = [C.POINTER(KModelDesc), C.c_int, C.c_int, C.c_uint64, C.c_int64, C.POINTER(vp)]
    lib.kmanip_reset.argtypes = [vp, vp, vp, vp]
    lib.kmanip_step.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.kmanip_step_chunk.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp]
    lib.kmanip_get_state.argtypes = [vp, f64p, f64p, f64p, f64p, i32p]
    lib.kmanip_set_state.argtypes = [vp, f64p, f64p, f64p, f64p, i32p]
    lib.kmanip_get_episode.argtypes = [vp, i32p]
    lib.kmanip_set_episode.argtypes = [vp, i32p]
    lib.kmanip_get_counters.argtypes = [vp, vp, vp, vp]
    lib.kmanip_bind_sim_time.argtypes = [vp, vp]
    lib.kmanip_bind_reward_done_record.argtypes = [vp, vp, vp]
    lib.kmanip_select_reward_done_record.argtypes = [vp, C.c_int]
    lib.kmanip_observe.argtypes = [vp, vp, vp, vp]
    lib.kmanip_set_seed.argtypes = [vp, C.c_uint64, C.c_int]
    lib.kmanip_get_diag.argtypes = [vp, C.POINTER(C.c_uint32), i32p, i32p]
    lib.kmanip_timing_summary.argtypes = [vp, f64p, f64p, f64p, i32p]
    lib.kmanip_enable_timing.argtypes = [vp, C.c_int]
    lib.kmanip_ik.argtypes = [vp, C.c_int, C.c_int, f64p, f64p, f64p, f64p, i32p, i32p]
    lib.kmanip_ik_eval.argtypes = [vp, C.c_int, C.c_int, f64p, f64p, f64p, f64p, f64p]
    lib.kmanip_render_depth.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp]
    lib.kmanip_render_rgb.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp]
    lib.kmanip_render_rgb_multi.argtypes = [vp, C.c_int, i32p, i32p, i32p, C.POINTER(vp), vp]
    lib.kmanip_snapshot_render_state.argtypes = [vp, C.c_int, vp]
    lib.kmanip_set_render_source.argtypes = [vp, C.c_int]
    lib.kmanip_bind_step_depth.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp]
    lib.kmanip_scripted_action.argtypes = [vp, vp, vp]
    lib.kmanip_sample_action.argtypes = [vp, vp, C.c_int, vp]
    lib.kmanip_num_envs.argtypes = [vp]
    lib.kmanip_last_error.argtypes = [vp]
    lib.kmanip_last_error.restype = C.c_char_p
    lib.kmanip_version.restype = C.c_char_p
    lib.kmanip_destroy.argtypes = [vp]
    lib.kmanip_destroy.restype = None
    lib.kmanip_dbg_wave_clocks.argtypes = [vp, C.POINTER(C.c_ulonglong), i32p, i32p]
    lib.kmanip_dbg_wave_slots.argtypes = [vp]
    if lib.kmanip_model_desc_size() != C.sizeof(KModelDesc):
        raise KManipError("KModelDesc layout mismatch: lib %d vs python %d"
                          % (lib.kmanip_model_desc_size(), C.sizeof(KModelDesc)))
    _lib = lib
    return lib
