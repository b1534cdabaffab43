// kmanip_device.hpp -- device-side helpers shared by the IK and dynamics kernels (gfx950, wave64).
// Arithmetic type is double: the reference computes in float64 (MuJoCo mjtNum, OBS_DTYPE
// gym_kmanip/__init__.py:50) and MI355X runs FP64 FMA at half the FP32 vector rate.
#pragma once
#include <stdlib.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "../../include/kmanip_debug.h"   // includes kmanip.h
#include "kmanip_math.hpp"

#define KM_MAX_CHAIN 8
#define KM_BLOCK_MAX 11
#define MJ_MINVAL 1e-15
#define MJ_MINIMP 0.0001
#define MJ_MAXIMP 0.9999

// Host-precomputed lookup data that is derived from KModelDesc (kept out of the ABI struct).
struct KModelAux {
  uint32_t anc_mask[KM_MAX_LINKS];           // bit j set <=> dof j is link i or one of its ancestors
  uint32_t desc_mask[KM_MAX_LINKS];          // bit j set <=> link j is link i or one of its descendants
  int32_t jump[4][KM_MAX_LINKS];             // jump[k][i] = the 2^k-th ancestor of link i (-1: none) -- FK pointer jumping
  int32_t fk_rounds;                         // ceil(log2(tree depth)) <= 4
  int32_t split;                             // links [0, split) and [split, nlink) share no kinematic tree (two-arm models: the joint-space
                                             // inertia is two diagonal blocks of <= KM_BLOCK_MAX dofs); 0 = no such split
  int32_t chain_len[KM_MAX_ARMS];            // IK kinematic chain root -> site link
  int32_t chain_link[KM_MAX_ARMS][KM_MAX_CHAIN];
  int32_t chain_xidx[KM_MAX_ARMS][KM_MAX_CHAIN];  // index into the IK unknowns, -1 = fixed at current qpos
  double chain_R[KM_MAX_ARMS][KM_MAX_CHAIN][9];   // constant rotation of each chain link in its parent (from link_quat)
  double site_R[KM_MAX_ARMS][9];                  // constant rotation of the EE site in its link
  // camera renders (kmanip_render.hip): per-model constants every workgroup would otherwise recompute from the desc
  double link_R[KM_MAX_LINKS][9];                 // constant rotation of each link in its parent (from link_quat, normalised)
  double cam_tanhalf[KM_MAX_CAMS];                // tan(fovy / 2)
  int32_t nvis;                                   // visible spheres (the finger tips), at most KM_RENDER_MAXVIS
  int32_t vis_sphere[4];
};
#define KM_RENDER_MAXVIS 4

// `staged` (round 6): the per-workgroup LDS image of the model constants (kmanip_dyn.hip: LModel<NL>), built ONCE at kmanip_create by
// k_prepare_model.  Every workgroup of every step used to derive it itself -- ~40 per-lane global loads and, on lane 0, eighteen
// dependent scalar-load round trips (solref / solimp staging, the collider pairs' diagApprox constants): 6-8 k cycles at the start
// of every wave of every launch.  Now a launch's workgroups copy 4-8 KB with one batch of 16-byte loads.
#define KM_LMODEL_MAX 8192
struct KDeviceModel {
  KModelDesc d;
  KModelAux x;
  alignas(16) unsigned char staged[KM_LMODEL_MAX];
};

typedef double real;

// XCD-aware workgroup -> env-block mapping.  Workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share one), and
// the state is stored field-major ([field][env]): a workgroup's few envs touch 32 B of every 64-128 B line, its neighbours in
// env order the rest.  Giving each XCD a CONTIGUOUS range of env blocks keeps those neighbours behind the same L2, so a line is
// fetched from HBM once instead of once per XCD (measured: FETCH_SIZE per k_step launch 5.6 MB -> 3.2 MB, profiles/r02f_pmc_hbm.json).  A bijection
// of [0, nblocks) for any nblocks; the results do not depend on it (envs are independent).
// position of the k-th (0-based) set bit of x (popcount(x) > k)
__host__ __device__ inline int select64(unsigned long long x, int k) {
  int pos = 0;
#pragma unroll
  for (int w = 32; w >= 1; w >>= 1) {
    const unsigned long long lowm = (1ull << w) - 1ull;
#if defined(__HIP_DEVICE_COMPILE__)
    const int c = __popcll(x & lowm);
#else
    const int c = __builtin_popcountll(x & lowm);
#endif
    if (k >= c) { k -= c; pos += w; x >>= w; }
  }
  return pos;
}
// SPREAD: the env (index inside its 64-env block) that lane group `grp` of the block's wave `j` takes, from the block's masks:
// M = heavy (a collider on or near the cube); S1, S2 = the two bits of a cost score of the others (bit 0: the cube does not rest on
// four corners -- a longer cube problem; bit 1: a sphere on the table -- a non-trivial arm problem whose iterations outlast a plain
// one's).  Waves j < nh1 = min(popcount(M), waves per block) take the j-th heavy env into group 0; everything else is dealt in
// ascending score order -- the heavy waves' other groups first -- so that a heavy env's wave-mates are the cheapest envs of the
// block and envs of a kind sit together in the block's later waves (like with like: their loops run in lockstep).
// EPB envs per wave, 64 / EPB waves per block.  A permutation of 0..63 whatever the masks are.
__host__ __device__ inline int km_popc64(unsigned long long x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __popcll(x);
#else
  return __builtin_popcountll(x);
#endif
}
__host__ __device__ inline int spread_pick(unsigned long long M, unsigned long long S1, unsigned long long S2, int j, int grp, int EPB) {
  const int WPB = 64 / EPB;
  const int nhb = km_popc64(M);
  const int nh1 = nhb < WPB ? nhb : WPB;
  unsigned long long H = M;
  if (nhb > WPB) { const int pos = select64(M, WPB - 1); H = M & ((2ull << pos) - 1ull); }
  if (j < nh1 && grp == 0) return select64(H, j);
  int k = j < nh1 ? (EPB - 1) * j + grp - 1 : (EPB - 1) * nh1 + EPB * (j - nh1) + grp;
  const unsigned long long cls[4] = {~H & ~S2 & ~S1, ~H & ~S2 & S1, ~H & S2 & ~S1, ~H & S2 & S1};
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const int n = km_popc64(cls[c]);
    if (k < n) return select64(cls[c], k);
    k -= n;
  }
  return select64(cls[3], k);
}

__device__ __forceinline__ int xcd_block(int b, int nblocks) {
  const int x = b & 7, i = b >> 3, base = nblocks >> 3, rem = nblocks & 7;
  return x * base + (x < rem ? x : rem) + i;
}


// LDS LATENCY (round 6).  With one wave per SIMD nothing hides an LDS round trip (~64-130 cycles), and the compiler both SINKS
// loads into the conditional blocks that use them and waits for each before it issues the next: a phase that reads its inputs
// where it needs them pays one round trip per input (the constraint assembly: ~30 in a row).  km_pin names values that must be
// in registers at this point -- ONE empty asm with all of them as operands -- so that their (independent) loads are issued
// back to back in front of it and waited for once.  No instruction is emitted; the arithmetic is untouched.
__device__ __forceinline__ void km_pin(real& a) { asm volatile("" : "+v"(a)); }
__device__ __forceinline__ void km_pin(real& a, real& b) { asm volatile("" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void km_pin(real& a, real& b, real& c) { asm volatile("" : "+v"(a), "+v"(b), "+v"(c)); }
__device__ __forceinline__ void km_pin(real& a, real& b, real& c, real& d) { asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
__device__ __forceinline__ void km_pin(real& a, real& b, real& c, real& d, real& e) { asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e)); }
__device__ __forceinline__ void km_pin(real& a, real& b, real& c, real& d, real& e, real& f) {
  asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));
}
__device__ __forceinline__ void km_pin(real (&v)[3]) { asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2])); }
__device__ __forceinline__ void km_pin(real (&v)[9]) {
  asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]));
}
__device__ __forceinline__ void km_pin(real (&a)[3], real (&b)[3]) {
  asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]));
}
__device__ __forceinline__ void km_pin(real (&a)[3], real (&b)[9]) {
  asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]), "+v"(b[8]));
}
__device__ __forceinline__ void km_pin_i(int& a) { asm volatile("" : "+v"(a)); }
__device__ __forceinline__ void km_pin_i(int& a, int& b) { asm volatile("" : "+v"(a), "+v"(b)); }

__device__ __forceinline__ real dot3(const real* a, const real* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
__device__ __forceinline__ void cross3(real* r, const real* a, const real* b) {
  real x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
// mju_normalize3
__device__ __forceinline__ real normalize3(real* v) {
  real n = sqrt(dot3(v, v));
  if (n < MJ_MINVAL) { v[0] = 1; v[1] = 0; v[2] = 0; }
  else { real inv = 1.0 / n; v[0] *= inv; v[1] *= inv; v[2] *= inv; }
  return n;
}
__device__ __forceinline__ void normalize4(real* q) {
  real n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < MJ_MINVAL) { q[0] = 1; q[1] = 0; q[2] = 0; q[3] = 0; }
  else { real inv = 1.0 / n; q[0] *= inv; q[1] *= inv; q[2] *= inv; q[3] *= inv; }
}
__device__ __forceinline__ void qmul(real* r, const real* a, const real* b) {
  real w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  real x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  real y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  real z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  r[0] = w; r[1] = x; r[2] = y; r[3] = z;
}
__device__ __forceinline__ void quat2mat(real* m, const real* q) {
  real w = q[0], x = q[1], y = q[2], z = q[3];
  m[0] = w * w + x * x - y * y - z * z; m[1] = 2 * (x * y - w * z); m[2] = 2 * (x * z + w * y);
  m[3] = 2 * (x * y + w * z); m[4] = w * w - x * x + y * y - z * z; m[5] = 2 * (y * z - w * x);
  m[6] = 2 * (x * z - w * y); m[7] = 2 * (y * z + w * x); m[8] = w * w - x * x - y * y + z * z;
}
__device__ __forceinline__ void mat_vec3(real* r, const real* m, const real* v) {
  real x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2];
  real y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2];
  real z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
__device__ __forceinline__ void matT_vec3(real* r, const real* m, const real* v) {
  real x = m[0] * v[0] + m[3] * v[1] + m[6] * v[2];
  real y = m[1] * v[0] + m[4] * v[1] + m[7] * v[2];
  real z = m[2] * v[0] + m[5] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
__device__ __forceinline__ void axis_angle2quat(real* q, const real* axis, real angle) {
  if (angle == 0) { q[0] = 1; q[1] = 0; q[2] = 0; q[3] = 0; return; }
  real s, c;
  km_sincos(angle * 0.5, &s, &c);
  q[0] = c; q[1] = axis[0] * s; q[2] = axis[1] * s; q[3] = axis[2] * s;
}
// 1/sqrt(s) to double precision: hardware estimate + two Newton steps (no IEEE sqrt / divide sequences)
__device__ __forceinline__ real rsqrt_nr(real s) {
  real y = __builtin_amdgcn_rsq(s);
  y = y * (1.5 - 0.5 * s * y * y);
  y = y * (1.5 - 0.5 * s * y * y);
  return y;
}

// 1/x to double precision: hardware estimate + two Newton steps (6 instructions instead of the ~14 of an IEEE divide; last-bit
// differences only).
__device__ __forceinline__ real frcp(real x) {
  real r = __builtin_amdgcn_rcp(x);
  r = r + r * (1.0 - x * r);
  r = r + r * (1.0 - x * r);
  return r;
}
// mju_normalize3 / normalize4 without sqrt + divide sequences
__device__ __forceinline__ real normalize3_fast(real* v) {
  const real s = dot3(v, v);
  if (s < MJ_MINVAL * MJ_MINVAL) { v[0] = 1; v[1] = 0; v[2] = 0; return km_sqrt(s); }
  const real inv = rsqrt_nr(s);
  v[0] *= inv; v[1] *= inv; v[2] *= inv;
  return s * inv;
}
__device__ __forceinline__ void normalize4_fast(real* q) {
  const real s = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  if (s < MJ_MINVAL * MJ_MINVAL) { q[0] = 1; q[1] = 0; q[2] = 0; q[3] = 0; return; }
  const real inv = rsqrt_nr(s);
  q[0] *= inv; q[1] *= inv; q[2] *= inv; q[3] *= inv;
}
// mju_mat2Quat: ONE square root and one reciprocal whichever of the four cases applies (the case picks their argument)
__device__ __forceinline__ void mat2quat(real* q, const real* m) {
  const real tr = m[0] + m[4] + m[8];
  const int cs = tr > 0 ? 0 : ((m[0] > m[4] && m[0] > m[8]) ? 1 : (m[4] > m[8] ? 2 : 3));
  const real arg = cs == 0 ? 1 + tr : (cs == 1 ? 1 + m[0] - m[4] - m[8] : (cs == 2 ? 1 - m[0] + m[4] - m[8] : 1 - m[0] - m[4] + m[8]));
  const real r = 0.5 * km_sqrt(arg), s = 0.25 * frcp(r);
  const real d75 = s * (m[7] - m[5]), d26 = s * (m[2] - m[6]), d31 = s * (m[3] - m[1]);
  const real s13 = s * (m[1] + m[3]), s26 = s * (m[2] + m[6]), s57 = s * (m[5] + m[7]);
  q[0] = cs == 0 ? r : (cs == 1 ? d75 : (cs == 2 ? d26 : d31));
  q[1] = cs == 0 ? d75 : (cs == 1 ? r : (cs == 2 ? s13 : s26));
  q[2] = cs == 0 ? d26 : (cs == 1 ? s13 : (cs == 2 ? r : s57));
  q[3] = cs == 0 ? d31 : (cs == 1 ? s26 : (cs == 2 ? s57 : r));
  normalize4_fast(q);
}
// mju_subQuat(res, qa, qb)
__device__ __forceinline__ void sub_quat(real* res, const real* qa, const real* qb) {
  real qn[4] = {qb[0], -qb[1], -qb[2], -qb[3]}, qd[4];
  qmul(qd, qn, qa);
  real ax[3] = {qd[1], qd[2], qd[3]};
  real sn = normalize3_fast(ax);
  real speed = 2 * km_atan2(sn, qd[0]);
  if (speed > M_PI) speed -= 2 * M_PI;
  res[0] = ax[0] * speed; res[1] = ax[1] * speed; res[2] = ax[2] * speed;
}

// mju_subQuat that also hands out sin and |cos| of HALF the (wrapped) rotation angle: tan(|res| / 2) = sn / ac exactly, so
// mjd_subQuat's half / tan(half) needs no tan()
__device__ __forceinline__ void sub_quat_sc(real* res, const real* qa, const real* qb, real& sn_out, real& ac_out) {
  real qn[4] = {qb[0], -qb[1], -qb[2], -qb[3]}, qd[4];
  qmul(qd, qn, qa);
  real ax[3] = {qd[1], qd[2], qd[3]};
  real sn = normalize3_fast(ax);
  real speed = 2 * km_atan2(sn, qd[0]);
  if (speed > M_PI) speed -= 2 * M_PI;
  res[0] = ax[0] * speed; res[1] = ax[1] * speed; res[2] = ax[2] * speed;
  sn_out = sn; ac_out = fabs(qd[0]);
}

// cross-lane double move with a DPP control word (a DPP row is 16 lanes)
template <int CTRL> __device__ __forceinline__ real dpp_f64(real v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  // (old = 0 with bound_ctrl: every control used here reads a lane of the row, so the old value never shows -- and the
  // compiler need not copy the source into the destination first: 2 instead of 4 instructions per 64-bit move)
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
// broadcast of lane K's value (K compile-time) to every lane of its G-lane group, registers only:
// row_newbcast inside a 16-lane DPP row; for two-row groups the gfx950 v_permlane16_swap hands the even row's
// value to the odd row (first result) or the odd row's to the even row (second result).
template <int G, int K> __device__ __forceinline__ real gbcast(real v) {
  static_assert(G == 8 || G == 16 || G == 32, "group = half a DPP row, one row or two rows");
  if constexpr (G == 32) {
    const real b = __builtin_amdgcn_update_dpp(0.0, v, 0x150 + (K & 15), 0xF, 0xF, true);
    const unsigned lo = (unsigned)__double2loint(b), hi = (unsigned)__double2hiint(b);
    const auto rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    constexpr int w = (K >> 4) & 1;
    return __hiloint2double((int)rh[w], (int)rl[w]);
  } else {
    // one v_mov_b64_dpp: gfx90a+ DPP on 64-bit operands exists for exactly this control (row_newbcast)
    return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + K, 0xF, 0xF, true);      // G == 8 callers pass K already offset into the row
  }
}
// acc += bcast_K(x) * t in ONE instruction: v_fmac_f64 with the DPP row_newbcast modifier on its first source (the
// 64-bit DPP encoding gfx90a+ provides for exactly this control).  The compiler does not fold its own v_mov_b64_dpp
// into the FMA, and it does not track hazards inside inline asm, so the two wait states a DPP read needs after a VALU
// write of its source register (e.g. an AGPR reload just before) are spent explicitly.  One 16-lane row only.
template <int K> __device__ __forceinline__ void fmac_bcast16(real& acc, real x, real t) {
  asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(t), "n"(K));
}
// acc -= bcast_K(x) * t
template <int K> __device__ __forceinline__ void fnmac_bcast16(real& acc, real x, real t) {
  asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(t), "n"(K));
}
// The compiler's hazard recogniser does not look inside inline asm: a register WRITTEN by one of the asm runs here and then read
// through DPP by compiler-generated code (gbcast, dpp_f64, bcast8) within the next two instructions would be read too early
// (the 2 wait states a DPP read needs after a VALU write).  dpp_settle(x) spends them and pins x's definition before it.
__device__ __forceinline__ void dpp_settle(real& x) { asm volatile("s_nop 1" : "+v"(x)); }
// The same for one- or two-row groups.  A broadcast source is prepared once (BSrc): for G = 16 the value itself; for
// G = 32 two copies made by one v_permlane16_swap per 32-bit half -- `e` carries the even row's values in both rows
// of the pair, `o` the odd row's -- so that lane K of the 32-lane group is row_newbcast:(K & 15) of the right copy and
// the broadcast still folds into the FMA.
template <int G> struct BSrc;
template <> struct BSrc<16> { real v; };
template <> struct BSrc<32> { real e, o; };
template <int G> __device__ __forceinline__ BSrc<G> bsrc(real x) {
  if constexpr (G == 32) {
    const unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
    const auto rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    BSrc<32> r;
    r.e = __hiloint2double((int)rh[0], (int)rl[0]);
    r.o = __hiloint2double((int)rh[1], (int)rl[1]);
    return r;
  } else {
    BSrc<16> r; r.v = x; return r;
  }
}
template <int G, int K> __device__ __forceinline__ void fmac_b(real& acc, const BSrc<G>& s, real t) {
  if constexpr (G == 32) { if constexpr (K < 16) fmac_bcast16<K>(acc, s.e, t); else fmac_bcast16<K - 16>(acc, s.o, t); }
  else fmac_bcast16<K>(acc, s.v, t);
}
template <int G, int K> __device__ __forceinline__ void fnmac_b(real& acc, const BSrc<G>& s, real t) {
  if constexpr (G == 32) { if constexpr (K < 16) fnmac_bcast16<K>(acc, s.e, t); else fnmac_bcast16<K - 16>(acc, s.o, t); }
  else fnmac_bcast16<K>(acc, s.v, t);
}
// ---- runs of broadcast-FMAs behind ONE pair of DPP wait states.  Inside a run no instruction writes a register that a later
// one reads through DPP (accumulators are plain VALU operands), so only the first DPP read can trail a VALU write of its
// source and need the two wait states; fmac_b / fnmac_b spend them before EVERY instruction.  Same arithmetic, fewer issue slots.
template <int G, int K> __device__ __forceinline__ real bsel(const BSrc<G>& s) {
  if constexpr (G == 32) return K < 16 ? s.e : s.o; else return s.v;
}
#define KM_DPPF(N, A, X, T, K) "v_fmac_f64_dpp %" #A ", " N "%" #X ", %" #T " row_newbcast:%" #K " row_mask:0xf bank_mask:0xf\n\t"
// WAIT = false: the caller knows that no DPP source of the run was written by the last two VALU instructions before it (a later
// run over the SAME sources, ordered behind the first through its accumulators) and saves the two wait states
// (the runs are `asm volatile`: they keep their program order among themselves, which is what makes "a later run" mean "issued
// later" -- a plain asm may be scheduled ahead of an earlier, independent one)
#define KM_RUN(WAIT, BODY, ...) do { if constexpr (WAIT) asm volatile("s_nop 1\n\t" BODY __VA_ARGS__); else asm volatile(BODY __VA_ARGS__); } while (0)
// acc_i (-)= bcast_{K_i}(x_i) * t_i, i = 0..3 (four different accumulators)
template <bool NEG, int K0, int K1, int K2, int K3, bool WAIT = true>
__device__ __forceinline__ void dppfma4(real& a0, real x0, real t0, real& a1, real x1, real t1, real& a2, real x2, real t2, real& a3, real x3, real t3) {
  if constexpr (NEG)
    KM_RUN(WAIT, KM_DPPF("-", 0, 4, 8, 12) KM_DPPF("-", 1, 5, 9, 13) KM_DPPF("-", 2, 6, 10, 14) KM_DPPF("-", 3, 7, 11, 15),
           : "+&v"(a0), "+&v"(a1), "+&v"(a2), "+&v"(a3) : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(t0), "v"(t1), "v"(t2), "v"(t3), "n"(K0), "n"(K1), "n"(K2), "n"(K3));
  else
    KM_RUN(WAIT, KM_DPPF("", 0, 4, 8, 12) KM_DPPF("", 1, 5, 9, 13) KM_DPPF("", 2, 6, 10, 14) KM_DPPF("", 3, 7, 11, 15),
           : "+&v"(a0), "+&v"(a1), "+&v"(a2), "+&v"(a3) : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(t0), "v"(t1), "v"(t2), "v"(t3), "n"(K0), "n"(K1), "n"(K2), "n"(K3));
}
template <bool NEG, int K0, int K1, int K2, bool WAIT = true>
__device__ __forceinline__ void dppfma3(real& a0, real x0, real t0, real& a1, real x1, real t1, real& a2, real x2, real t2) {
  if constexpr (NEG)
    KM_RUN(WAIT, KM_DPPF("-", 0, 3, 6, 9) KM_DPPF("-", 1, 4, 7, 10) KM_DPPF("-", 2, 5, 8, 11),
           : "+&v"(a0), "+&v"(a1), "+&v"(a2) : "v"(x0), "v"(x1), "v"(x2), "v"(t0), "v"(t1), "v"(t2), "n"(K0), "n"(K1), "n"(K2));
  else
    KM_RUN(WAIT, KM_DPPF("", 0, 3, 6, 9) KM_DPPF("", 1, 4, 7, 10) KM_DPPF("", 2, 5, 8, 11),
           : "+&v"(a0), "+&v"(a1), "+&v"(a2) : "v"(x0), "v"(x1), "v"(x2), "v"(t0), "v"(t1), "v"(t2), "n"(K0), "n"(K1), "n"(K2));
}
template <bool NEG, int K0, int K1, bool WAIT = true>
__device__ __forceinline__ void dppfma2(real& a0, real x0, real t0, real& a1, real x1, real t1) {
  if constexpr (NEG)
    KM_RUN(WAIT, KM_DPPF("-", 0, 2, 4, 6) KM_DPPF("-", 1, 3, 5, 7), : "+&v"(a0), "+&v"(a1) : "v"(x0), "v"(x1), "v"(t0), "v"(t1), "n"(K0), "n"(K1));
  else
    KM_RUN(WAIT, KM_DPPF("", 0, 2, 4, 6) KM_DPPF("", 1, 3, 5, 7), : "+&v"(a0), "+&v"(a1) : "v"(x0), "v"(x1), "v"(t0), "v"(t1), "n"(K0), "n"(K1));
}
// single instruction, optional wait states
template <bool NEG, int K, bool WAIT = true>
__device__ __forceinline__ void dppfma1(real& a0, real x0, real t0) {
  if constexpr (NEG) KM_RUN(WAIT, KM_DPPF("-", 0, 1, 2, 3), : "+v"(a0) : "v"(x0), "v"(t0), "n"(K));
  else KM_RUN(WAIT, KM_DPPF("", 0, 1, 2, 3), : "+v"(a0) : "v"(x0), "v"(t0), "n"(K));
}
// ap += bcast_K(xp) * tp ; an -= bcast_K(xn) * tn (one run, two accumulators, one lane K)
template <int K, bool WAIT = true>
__device__ __forceinline__ void dppfma_pn(real& ap, real xp, real tp, real& an, real xn, real tn) {
  KM_RUN(WAIT, KM_DPPF("", 0, 2, 4, 6) KM_DPPF("-", 1, 3, 5, 6), : "+&v"(ap), "+&v"(an) : "v"(xp), "v"(xn), "v"(tp), "v"(tn), "n"(K));
}
// acc += sum_i bcast_K(x_i) * t_i, i = 0..NS-1 (one accumulator, NS = 3 or 4 sources, one lane K), in this order
template <int K, bool WAIT = true>
__device__ __forceinline__ void dppfma_acc4(real& acc, real x0, real t0, real x1, real t1, real x2, real t2, real x3, real t3) {
  KM_RUN(WAIT, KM_DPPF("", 0, 1, 5, 9) KM_DPPF("", 0, 2, 6, 9) KM_DPPF("", 0, 3, 7, 9) KM_DPPF("", 0, 4, 8, 9),
         : "+&v"(acc) : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(t0), "v"(t1), "v"(t2), "v"(t3), "n"(K));
}
template <int K, bool WAIT = true>
__device__ __forceinline__ void dppfma_acc3(real& acc, real x0, real t0, real x1, real t1, real x2, real t2) {
  KM_RUN(WAIT, KM_DPPF("", 0, 1, 4, 7) KM_DPPF("", 0, 2, 5, 7) KM_DPPF("", 0, 3, 6, 7),
         : "+&v"(acc) : "v"(x0), "v"(x1), "v"(x2), "v"(t0), "v"(t1), "v"(t2), "n"(K));
}
// acc (-)= bcast_K(x0) * t0 + bcast_K(x1) * t1 (one accumulator, two sources, one lane K), in this order
template <int K, bool WAIT = true>
__device__ __forceinline__ void dppfma_acc2n(real& acc, real x0, real t0, real x1, real t1) {
  KM_RUN(WAIT, KM_DPPF("-", 0, 1, 3, 5) KM_DPPF("-", 0, 2, 4, 5), : "+&v"(acc) : "v"(x0), "v"(x1), "v"(t0), "v"(t1), "n"(K));
}
template <int K, bool WAIT = true>
__device__ __forceinline__ void dppfma_acc2(real& acc, real x0, real t0, real x1, real t1) {
  KM_RUN(WAIT, KM_DPPF("", 0, 1, 3, 5) KM_DPPF("", 0, 2, 4, 5), : "+&v"(acc) : "v"(x0), "v"(x1), "v"(t0), "v"(t1), "n"(K));
}
// acc -= bcast_P(x) * t0 + bcast_Q(x) * t1: two lanes of ONE distributed vector (x must not be acc: a run never reads through DPP
// what it writes)
template <int P, int Q, bool WAIT = true>
__device__ __forceinline__ void dppfma_row2n(real& acc, real x, real t0, real t1) {
  KM_RUN(WAIT, KM_DPPF("-", 0, 1, 2, 4) KM_DPPF("-", 0, 1, 3, 5), : "+&v"(acc) : "v"(x), "v"(t0), "v"(t1), "n"(P), "n"(Q));
}
// acc += sum_i bcast_{K_i}(x) * t_i, i = 0..3: four lanes of ONE distributed vector against four coefficients (a row-times-
// vector product), in this order
template <int K0, int K1, int K2, int K3, bool WAIT = true>
__device__ __forceinline__ void dppfma_row4(real& acc, real x0, real x1, real x2, real x3, real t0, real t1, real t2, real t3) {
  KM_RUN(WAIT, KM_DPPF("", 0, 1, 5, 9) KM_DPPF("", 0, 2, 6, 10) KM_DPPF("", 0, 3, 7, 11) KM_DPPF("", 0, 4, 8, 12),
         : "+&v"(acc) : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(t0), "v"(t1), "v"(t2), "v"(t3), "n"(K0), "n"(K1), "n"(K2), "n"(K3));
}
// acc += sum_{j in [J0, J1)} bcast_j(x) * row(j), row given as a callable (registers or LDS), runs of four then singles; every
// run reads the same source x and accumulates into the same register, so only the first one waits for x
template <int G, int J0, int J1, bool WAIT = true, class ROW>
__device__ __forceinline__ void fmac_rowvec(real& acc, const BSrc<G>& x, ROW&& row) {
  if constexpr (J1 - J0 >= 4) {
    dppfma_row4<J0 & 15, (J0 + 1) & 15, (J0 + 2) & 15, (J0 + 3) & 15, WAIT>(acc, bsel<G, J0>(x), bsel<G, J0 + 1>(x), bsel<G, J0 + 2>(x), bsel<G, J0 + 3>(x),
                                                                        row(J0), row(J0 + 1), row(J0 + 2), row(J0 + 3));
    fmac_rowvec<G, J0 + 4, J1, false>(acc, x, row);
  } else if constexpr (J1 - J0 >= 1) {
    dppfma1<false, J0 & 15, WAIT>(acc, bsel<G, J0>(x), row(J0));
    fmac_rowvec<G, J0 + 1, J1, false>(acc, x, row);
  }
}
// a[j] -= bcast_j(src) * t for j in [J0, J1): the row update of a right-looking factorisation, in runs of four / two / one
// (one source for all of them: only the first run waits)
template <int G, int J0, int J1, int N, bool WAIT = true>
__device__ __forceinline__ void fnmac_cols(real (&a)[N], const BSrc<G>& src, real t) {
  if constexpr (J1 - J0 >= 4) {
    dppfma4<true, J0 & 15, (J0 + 1) & 15, (J0 + 2) & 15, (J0 + 3) & 15, WAIT>(a[J0], bsel<G, J0>(src), t, a[J0 + 1], bsel<G, J0 + 1>(src), t,
                                                                            a[J0 + 2], bsel<G, J0 + 2>(src), t, a[J0 + 3], bsel<G, J0 + 3>(src), t);
    fnmac_cols<G, J0 + 4, J1, N, false>(a, src, t);
  } else if constexpr (J1 - J0 >= 2) {
    dppfma2<true, J0 & 15, (J0 + 1) & 15, WAIT>(a[J0], bsel<G, J0>(src), t, a[J0 + 1], bsel<G, J0 + 1>(src), t);
    fnmac_cols<G, J0 + 2, J1, N, false>(a, src, t);
  } else if constexpr (J1 - J0 == 1) {
    dppfma1<true, J0 & 15, WAIT>(a[J0], bsel<G, J0>(src), t);
  }
}

// compile-time counted loop: f(std::integral_constant<int, K>) for K in [K0, N)
template <int K0, int N, class F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (K0 < N) { f(std::integral_constant<int, K0>{}); static_for<K0 + 1, N>(f); }
}
#define KM_GSYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

// ---- optional phase profiler (diagnostic build only: make prof -> -DKM_PROFILE).  Stamps go to a buffer of
// their own and never feed an output; the shipped library compiles every call away.
#define KM_NPH 48
#ifdef KM_PROFILE
static __device__ unsigned long long g_prof[KM_NPH];   // one accumulator per variant object; kmanip_dbg_prof reads the Solo/Newton one
#define KM_PROF_BLOCKS 4096
static __device__ unsigned long long g_prof_blk[KM_PROF_BLOCKS][4][KM_NPH];   // last launch, per workgroup and lane group (who is slow?)
struct Prof {
  unsigned long long t0, it0, acc[KM_NPH];
  // a second, independent stopwatch for spans that overlap the phase stamps (one Newton iteration, lone or beside wave-mates)
  __device__ __forceinline__ void it_begin() { it0 = __builtin_amdgcn_s_memtime(); }
  __device__ __forceinline__ void it_end(int i) { acc[i] += __builtin_amdgcn_s_memtime() - it0; }
  __device__ __forceinline__ void start() { for (int i = 0; i < KM_NPH; i++) acc[i] = 0; t0 = __builtin_amdgcn_s_memtime(); }
  __device__ __forceinline__ void ph(int i) {
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t = __builtin_amdgcn_s_memtime();
    acc[i] += t - t0; t0 = t;
    __builtin_amdgcn_sched_barrier(0);
  }
  __device__ __forceinline__ void cnt(int i, unsigned n) { acc[i] += n; }      // event counters share the stamp slots (40..)
  __device__ __forceinline__ void flush() {
    if (threadIdx.x == 0) for (int i = 0; i < KM_NPH; i++) atomicAdd(&g_prof[i], acc[i]);
    if ((threadIdx.x & 15) == 0 && blockIdx.x < KM_PROF_BLOCKS) for (int i = 0; i < KM_NPH; i++) g_prof_blk[blockIdx.x][threadIdx.x >> 4][i] = acc[i];
  }
};
#else
struct Prof {
  __device__ __forceinline__ void start() {}
  __device__ __forceinline__ void ph(int) {}
  __device__ __forceinline__ void it_begin() {}
  __device__ __forceinline__ void it_end(int) {}
  __device__ __forceinline__ void cnt(int, unsigned) {}
  __device__ __forceinline__ void flush() {}
};
#endif

// Philox4x32-10 (Salmon et al. 2011), counter-based RNG for the cube spawn
__device__ __host__ inline void philox4x32_10(const uint32_t* ctr, const uint32_t* key, uint32_t* out) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; r++) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __host__ inline double u53(uint32_t hi, uint32_t lo) {
  return ((double)(hi >> 5) * 67108864.0 + (double)(lo >> 6)) / 9007199254740992.0;
}

// action_space.sample() of a Box(-1, 1, float32) component from 32 random bits: ((r >> 8) - 2^23) * 2^-23, exact in float32
// (a 24-bit signed integer times a power of two), so the device and the CPU oracle produce identical bits
__device__ __host__ inline float km_action_from_u32(uint32_t r) { return (float)((int32_t)(r >> 8) - 8388608) * (1.0f / 8388608.0f); }
// counter word 3 of the action stream: the cube spawn uses 0 and 1 (reset_env), actions 0x10000 + 4 * step + block
#define KM_ACT_CTR3(step, blk) (0x10000u + 4u * (uint32_t)(step) + (uint32_t)(blk))

// Device state, struct-of-arrays over envs: element (k, env) of an [n_k, num_envs] array is at
// k * num_envs + env, so a wave reading component k for consecutive envs is fully coalesced.
#define KM_DISP_HDR 4
struct KDeviceState {
  double* qpos;       // [nq][N]
  double* qvel;       // [nv][N]
  double* ctrl;       // [nu][N]
  double* warm;       // [nv][N]  qacc_warmstart
  double* qpos_ik;    // [nl][N]  robot qpos after the IK's last evaluation (teleport, ik_mujoco.py:34,67)
  int32_t* step_idx;  // [N]
  int32_t* episode;   // [N]
  uint32_t* contact_mask;  // [N]
  int32_t* ik_nfev;   // [2][N]
  int32_t* ik_status; // [2][N]
  double* sim_time;   // [N] caller-owned (kmanip_bind_sim_time), may be NULL: data.time of every env = step_idx * control_dt
  double* rd_rec;     // [N][2] caller-owned (kmanip_bind_reward_done_record), may be NULL: this step's packed (reward, done) record
  int32_t* work;      // [N] Newton work units of the env's last control step (two-arm kernels; 0 otherwise): k_sort_envs' predictor
  unsigned long long* wave_clk;   // [N] or NULL (KMANIP_WAVE_CLOCKS=1, diagnostics): s_memtime ticks the wave that held slot s spent in k_step
  const int32_t* slot_env;   // [N] or NULL: env handled by wave slot s (k_sort_envs: predicted-cost order, heaviest first); NULL = identity
  // Heavy-first dispatch with variable wave occupancy (round 5; single-arm Newton kernel, launches of one residency round).
  // A dispatch table is int32[KM_DISP_HDR + N]: [0] = envs registered as HEAVY, [1] = as light, then the env list -- heavy envs
  // from the front in arrival order, everybody else from the back.  k_step reads disp_in (NULL: classic slot mapping): workgroups
  // [0, ceil(nh / disp_heavy_epb)) take disp_heavy_epb heavy envs each (1: a heavy env has a wave to itself), the following ones
  // EPB light envs each; at its end every env registers itself in disp_out for the NEXT launch (heavy = a collider on or within
  // KM_NEAR_MARGIN of the cube: the coupled Newton loop is on or about to start), and workgroup 0 clears the counters of disp_zero
  // (the table after next).  Three tables rotate on the host; any partition of the env ids is a valid table, and an env's bits
  // depend neither on its slot nor on its wave-mates.  (The round-5 EXPERIMENT, KMANIP_HEAVY_DISPATCH=1; the product is SPREAD, below.)
  // SPREAD (the default of the single-arm launches of two or four envs per wave; kmanip_api.hip): one byte per env -- bit 0 "heavy at
  // the end of its last step", bits 1-2 a cost score of the others (spread_pick) -- written by every step into spread_out and read by the
  // next launch from spread_in.  A wave looks at the 64 flags of ITS block of 64 consecutive envs (three ballots) and deals the block's
  // envs to the block's waves so that no wave holds two heavy ones and a heavy env's wave-mates are the block's plainest envs: a
  // permutation inside the block, whatever the flags are -- the block's cache lines are the ones the identity map touches.
  double near_margin;   // "near the cube" for the heavy flag / the sort's proximity bit (KM_NEAR_MARGIN; KMANIP_NEAR_MARGIN, A/B)
  const uint8_t* spread_in;
  uint8_t* spread_out;
  int spread_table;     // which score the flags carry (KMANIP_SPREAD_TABLE, A/B): 3 = both bits (default), 2 = both as one class, 1 = the table bit, 0 = none
  const int32_t* disp_in;
  int32_t* disp_out;
  int32_t* disp_zero;
  int disp_cap;         // at most this many envs are dispatched as heavy (the grid is sized for it); the rest of them as light
  int disp_heavy_epb;   // heavy envs per wave (1 | 2 | 4)
  double control_dt;  // n_sub_steps * timestep
  int num_envs;
  int64_t env_id_offset;
  uint64_t seed;
};

// fills dm->staged (device memory) for the model's link-count class; kmanip_create, once
void kmanip_launch_prepare_model(KDeviceModel* dm, const KModelDesc& hd, hipStream_t stream);
void kmanip_launch_ik_coop(const KDeviceModel* dm, const KModelDesc& hd, const KDeviceState& st, const float* act,
                           hipStream_t stream);
void kmanip_launch_ik_coop_standalone(const KDeviceModel* dm, const KModelDesc& hd, int arm, int n, double* qpos_env_major,
                                      const double* goal_pos, const double* goal_quat, double* q_out, int32_t* nfev,
                                      int32_t* status, hipStream_t stream);
void kmanip_launch_ik_eval_coop(const KDeviceModel* dm, const KModelDesc& hd, int arm, int n, const double* qpos_env_major,
                                const double* goal_pos, const double* goal_quat, double* res, double* jac, hipStream_t stream);
// act != NULL: the decode + IK of before_step run inside k_step (product path); NULL: they already ran
// nchunk > 1 (act != NULL only): that many control steps per launch, act / obs / reward / done laid out [nchunk][num_envs][..]
void kmanip_launch_step(const KDeviceModel* dm, const KModelDesc& hd, const KDeviceState& st, const float* act, double* obs,
                        double* reward, uint8_t* done, int nchunk, hipStream_t stream);
void kmanip_launch_reset(const KDeviceModel* dm, const KModelDesc& hd, const KDeviceState& st, const uint8_t* mask,
                         int use_done_bits, double* obs, hipStream_t stream);
void kmanip_launch_observe(const KDeviceModel* dm, const KModelDesc& hd, const KDeviceState& st, double* obs, double* reward,
                           hipStream_t stream);
void kmanip_launch_render_depth(const KDeviceModel* dm, const KDeviceState& st, int cam, int height, int width, float* depth,
                                hipStream_t stream);
// up to KM_MAX_CAMS camera images of every env in ONE launch (grid = envs x jobs): the *Vision observation
struct KRenderJobs { int n; int cam[KM_MAX_CAMS], height[KM_MAX_CAMS], width[KM_MAX_CAMS]; uint8_t* rgb[KM_MAX_CAMS]; };
void kmanip_launch_render_rgb(const KDeviceModel* dm, const KDeviceState& st, const KRenderJobs& jobs, hipStream_t stream);
// envs per workgroup (= per wave) of a step / reset launch: as many waves as the chip has SIMD slots for, but no more lanes idle than
// needed; the chunked kernel exists for the full shape only.  Shared by the launchers (kmanip_dyn.hip) and by the host code that
// has to know the LAST launch's shape (kmanip_api.hip: the slot -> env maps of kmanip_dbg_wave_clocks).
#define KM_TARGET_WAVES 1024   // 256 CUs x 4 SIMDs: below this many workgroups, fewer envs per wave fills more SIMDs
static inline int km_pick_epb(int num_envs, int max_epb) {
  const char* e = getenv("KMANIP_EPB");              // diagnostic override (tests exercise every launch shape)
  const int forced = e ? atoi(e) : 0;
  if (forced > 0) return forced < max_epb ? forced : max_epb;
  int epb = max_epb;
  while (epb > 1 && (num_envs + epb - 1) / epb < KM_TARGET_WAVES) epb >>= 1;
  return epb;
}
static inline int km_step_epb(int num_envs, int max_epb, int nchunk) { return nchunk > 1 ? max_epb : km_pick_epb(num_envs, max_epb); }
// slot_env[s] = the env wave slot s handles, envs ordered by the cost their LAST step predicts, heaviest first (LPT dispatch)
struct KCostWeights { int ik, work, coupled, armtab, cubetab, binw; };
void kmanip_launch_sort_envs(const KDeviceState& st, int32_t* slot_env, const KCostWeights& w, hipStream_t stream);
void kmanip_launch_scripted_action(const KDeviceModel* dm, const KDeviceState& st, float* act, hipStream_t stream);
void kmanip_launch_sample_action(const KDeviceModel* dm, const KDeviceState& st, float* act, int ahead, hipStream_t stream);
