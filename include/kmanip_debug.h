/*
 * kmanip_debug.h -- diagnostic entry points of libkmanip_hip.so.  NOT part of the drop-in boundary (include/kmanip.h):
 * nothing here replaces a reference interface; the measurement tools under tools/ and tests/tools/ and one GPU test
 * (tests/test_gpu_config_sizes.py, the cost-sorted dispatch order) bind them.  They are declared here so that every
 * exported symbol of the library is declared in a header (tests/test_abi.py).
 */
#ifndef KMANIP_DEBUG_H
#define KMANIP_DEBUG_H

#include "kmanip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Product build.  With KMANIP_WAVE_CLOCKS=1 in the environment at kmanip_create: per wave slot, the s_memtime ticks its wave
 * spent in the last k_step (clk, HOST unsigned long long[S]); always: the env each dispatch slot held in the LAST step launch
 * (slot_env, HOST int32[S]: the order k_sort_envs chose, the SPREAD map of that launch's shape, the identity otherwise) and every
 * env's work counter of its last step (work, HOST int32[num_envs]; zero on the single-arm kernel, which ships without the
 * counters).  S = kmanip_dbg_wave_slots(h) -- size clk and slot_env from THAT call, not from num_envs: a handle with the
 * heavy-first dispatch has more lane groups than envs.  Any pointer may be NULL; clk needs the environment variable (an error
 * otherwise).  Synchronous. */
KMANIP_API int kmanip_dbg_wave_clocks(KHandle h, unsigned long long* clk, int32_t* slot_env, int32_t* work);
/* Entries of the clk / slot_env arrays above: num_envs, or -- on a handle with the heavy-first dispatch (kmanip.h, launch shape),
 * whose grid has more lane groups than envs -- 4 per workgroup of the largest grid; slot_env is -1 for a lane group that held no env. */
KMANIP_API int kmanip_dbg_wave_slots(KHandle h);

#ifdef KM_PROFILE
/* -DKM_PROFILE build only (libkmanip_hip_prof.so, `make prof`; never shipped, never timed as the product): the in-kernel
 * phase stamps' accumulators, KM_NPH slots per variant object (tools/phase_profile.py). */
KMANIP_API int kmanip_dbg_prof(unsigned long long* out, int reset);          /* single-arm Newton object */
KMANIP_API int kmanip_dbg_prof_blocks(unsigned long long* out, int nblocks);  /* its first workgroups, per env */
KMANIP_API int kmanip_dbg_prof20(unsigned long long* out, int reset);        /* two-arm Newton object */
#endif

#ifdef __cplusplus
}
#endif
#endif /* KMANIP_DEBUG_H */
