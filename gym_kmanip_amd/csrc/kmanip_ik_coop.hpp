// kmanip_ik_coop.hpp -- cooperative batched action decode + bounded trust-region-reflective IK (gfx950).
//
// Same reference semantics as kmanip_ik.hip (KManipTask.before_step env_sim.py:38-108; ik_mujoco.py:20-155;
// scipy least_squares(method='trf', tr_solver='exact', bounds=jnt_range)), different mapping:
//
//   8 lanes per IK problem (env, arm), 8 problems per wave.  Lane c owns unknown c: x_c, its bounds, its
//   Coleman-Li scaling v_c / d_c, gradient g_c, step components, and COLUMN c of the 6 x n task Jacobian.
//   * forward kinematics: the n sincos run in parallel (one per lane), an inclusive prefix product over the lanes (three DPP
//     row-shift rounds) turns each lane's local transform into its link's world transform, the site link's reaches the others
//     by row broadcast.
//   * J^T f, costs, norms, step-to-bound ratios: per-lane scalars + 3-step DPP reductions over 8 lanes
//     (row_half_mirror, quad_perm, quad_perm) -- every lane gets the bitwise-identical result, so all control
//     flow (TRF branches, More iterations, termination tests) is uniform inside a problem.
//   * the 7x7 normal matrix J_h^T J_h + C lives one row per lane in registers; its Cholesky factor and both triangular
//     solves are cooperative (column broadcasts folded into the FMAs).  No LDS anywhere in the IK.
// Divisions are reciprocal (hardware estimate + two Newton steps: correctly rounded in 1 M samples) times numerator -- at most one
// ulp from an IEEE divide, 6 instructions instead of ~14.  Nothing of the TRF state is a 7-vector in one lane any more, which is what pushed the one-lane-per-problem
// kernel into 0.5-1.5 KB of scratch per lane.
#pragma once
#include "kmanip_device.hpp"
// FMA contraction is ON for this code like for the physics (-ffp-contract=fast-honor-pragmas); the one place where it
// must not act is inside the lane reductions (gsum8 below, gsum in kmanip_dyn.hip), whose results steer group-uniform
// control flow and therefore have to be bitwise identical on every lane of a problem.

#define GI 8            // lanes per problem
#define GS 16           // lane stride between problems: ONE problem per 16-lane DPP row (its lanes 0..7), so that a value of
                        // problem lane K is `row_newbcast:K` -- a single DPP control, foldable into v_fmac_f64_dpp (the 64-bit
                        // DPP encodings honour no bank_mask, so two problems per row cannot be told apart; measured)
#define PPW (64 / GS)   // problems per wave / workgroup

// ---- 8-lane group collectives (DPP half-row patterns); results identical in all 8 lanes
__device__ __forceinline__ real gsum8(real v) {
#pragma clang fp contract(off)     // keep the sum bitwise identical on all 8 lanes: no FMA with the caller's product (see gsum)
  v += dpp_f64<0x141>(v);   // row_half_mirror: i <-> 7 - i
  v += dpp_f64<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_f64<0x4E>(v);    // quad_perm [2,3,0,1]
  return v;
}
// N sums at once, step by step: the same operations per value as N gsum8 calls, the chains interleaved (see gsum_n)
template <int N> __device__ __forceinline__ void gsum8_n(real (&v)[N]) {
#pragma clang fp contract(off)
  real t[N];
#pragma unroll
  for (int i = 0; i < N; i++) t[i] = dpp_f64<0x141>(v[i]);
#pragma unroll
  for (int i = 0; i < N; i++) v[i] += t[i];
#pragma unroll
  for (int i = 0; i < N; i++) t[i] = dpp_f64<0xB1>(v[i]);
#pragma unroll
  for (int i = 0; i < N; i++) v[i] += t[i];
#pragma unroll
  for (int i = 0; i < N; i++) t[i] = dpp_f64<0x4E>(v[i]);
#pragma unroll
  for (int i = 0; i < N; i++) v[i] += t[i];
}
__device__ __forceinline__ real gmin8(real v) {
  v = fmin(v, dpp_f64<0x141>(v)); v = fmin(v, dpp_f64<0xB1>(v)); v = fmin(v, dpp_f64<0x4E>(v));
  return v;
}
__device__ __forceinline__ real gmax8(real v) {
  v = fmax(v, dpp_f64<0x141>(v)); v = fmax(v, dpp_f64<0xB1>(v)); v = fmax(v, dpp_f64<0x4E>(v));
  return v;
}
// true iff pred holds on every lane of the 8-lane group (lanes c >= n pass `true`)
__device__ __forceinline__ bool gall8(bool pred) {
  const unsigned long long b = __ballot(pred);
  const int sh = (threadIdx.x & 63) & ~7;
  return ((b >> sh) & 0xFFull) == 0xFFull;
}

// value of lane K (0..7) of this lane's problem on every lane of the problem, registers only (one problem per DPP row)
template <int K> __device__ __forceinline__ real bcast8(real v) { return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + K, 0xF, 0xF, true); }
// acc += bcast8<K>(x) * t with the broadcast folded into the FMA
template <int K> __device__ __forceinline__ void fmac_bcast8(real& acc, real x, real t) { fmac_bcast16<K>(acc, x, t); }

template <int N>
struct CoopCtx {
  const KModelDesc* m;
  const KModelAux* ax;
  int arm, c;                 // arm index, lane index inside the problem (unknown index)
  bool on;                    // c < N
  real goal_pos[3], goal_quat[4];
  real qfix;                  // joint value of the (at most one) chain link beyond the unknowns
  real q_prev, q_home, lb, ub;
  // this lane's chain link (position c of the root -> site chain): constant rotation / offset in its parent
  real cR[9], cp[3];
  int clen, cslide;           // chain length; 1 if this lane's link is a slide joint
  Prof* pf;                   // phase stamps of the diagnostic build (no-ops in the product)
};
// per-problem constants of lane c (call once per problem, after arm / c / m / ax are set)
template <int N>
__device__ __forceinline__ void coop_chain_setup(CoopCtx<N>& P) {
  P.clen = P.ax->chain_len[P.arm];
  const int k = P.c < P.clen ? P.c : 0;
  const int l = P.ax->chain_link[P.arm][k];
#pragma unroll
  for (int i = 0; i < 9; i++) P.cR[i] = P.ax->chain_R[P.arm][k][i];
  P.cp[0] = P.m->link_pos[l][0]; P.cp[1] = P.m->link_pos[l][1]; P.cp[2] = P.m->link_pos[l][2];
  P.cslide = P.m->jnt_type[l] == KM_JNT_SLIDE;
}
// value of lane (c - S) of the same 8-lane problem (garbage for c < S: callers mask); DPP row_shr
template <int S> __device__ __forceinline__ real shr8(real v) { return dpp_f64<0x110 + S>(v); }

// ik_res (+ ik_jac when JAC): residual task part ft[6] (uniform), this lane's Jacobian column Jc[6],
// site position / rotation (uniform).  x = this lane's unknown.
// Kinematics: lane c builds the transform of chain link c in its parent (its own sincos), an inclusive prefix
// product over the 8 lanes (3 DPP row-shift rounds) turns it into the link's world transform -- exactly the
// anchor and axis this lane's Jacobian column needs; the last link's transform is published through LDS.
template <int N, bool JAC>
__device__ __forceinline__ void coop_eval(const CoopCtx<N>& P, real x, real* ft, real* Jc, real* sp_out, real* smat_out) {
  const KModelDesc* m = P.m;
  const int arm = P.arm;
  const int clen = P.clen;
  real R[9], p[3];
  {
    const real q = P.on ? x : P.qfix;                   // (lane N holds the one fixed link behind the unknowns, if any)
    p[0] = P.cp[0]; p[1] = P.cp[1]; p[2] = P.cp[2];
    if (P.c >= clen) {
      R[0] = 1; R[1] = 0; R[2] = 0; R[3] = 0; R[4] = 1; R[5] = 0; R[6] = 0; R[7] = 0; R[8] = 1;
      p[0] = 0; p[1] = 0; p[2] = 0;
    } else if (P.cslide) {
#pragma unroll
      for (int i = 0; i < 9; i++) R[i] = P.cR[i];
      p[0] += R[2] * q; p[1] += R[5] * q; p[2] += R[8] * q;
    } else {
      real sn, cs;
      km_sincos(q, &sn, &cs);
#pragma unroll
      for (int i = 0; i < 3; i++) {
        R[3 * i] = cs * P.cR[3 * i] + sn * P.cR[3 * i + 1];
        R[3 * i + 1] = cs * P.cR[3 * i + 1] - sn * P.cR[3 * i];
        R[3 * i + 2] = P.cR[3 * i + 2];
      }
    }
  }
  static_for<0, 3>([&](auto rc) {
    constexpr int S = 1 << decltype(rc)::value;
    real A[9], pa[3];
#pragma unroll
    for (int i = 0; i < 9; i++) A[i] = shr8<S>(R[i]);
    pa[0] = shr8<S>(p[0]); pa[1] = shr8<S>(p[1]); pa[2] = shr8<S>(p[2]);
    if (P.c >= S) {
      real t[3], Rn[9];
      mat_vec3(t, A, p);
      p[0] = t[0] + pa[0]; p[1] = t[1] + pa[1]; p[2] = t[2] + pa[2];
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) Rn[3 * i + j] = A[3 * i] * R[j] + A[3 * i + 1] * R[3 + j] + A[3 * i + 2] * R[6 + j];
#pragma unroll
      for (int i = 0; i < 9; i++) R[i] = Rn[i];
    }
  });
  // lane GI-1 holds the product over the whole chain (positions >= clen are identities): the site link's world transform reaches
  // the other lanes by row broadcasts (registers only; round 2 published it through LDS: two synchronisations per evaluation)
  real mat[9], pos[3];
#pragma unroll
  for (int i = 0; i < 9; i++) mat[i] = bcast8<GI - 1>(R[i]);
  pos[0] = bcast8<GI - 1>(p[0]); pos[1] = bcast8<GI - 1>(p[1]); pos[2] = bcast8<GI - 1>(p[2]);
  const real anc[3] = {p[0], p[1], p[2]}, axw[3] = {R[2], R[5], R[8]};
  real sp[3], smat[9], cur[4], rq[3];
  real so[3] = {m->arm_site_pos[arm][0], m->arm_site_pos[arm][1], m->arm_site_pos[arm][2]};
  const double* Rs = P.ax->site_R[arm];
  mat_vec3(sp, mat, so);
  sp[0] += pos[0]; sp[1] += pos[1]; sp[2] += pos[2];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) smat[3 * i + j] = mat[3 * i] * Rs[j] + mat[3 * i + 1] * Rs[3 + j] + mat[3 * i + 2] * Rs[6 + j];
  mat2quat(cur, smat);
  real sn_h, ac_h;                                     // sin, |cos| of half the orientation error angle
  sub_quat_sc(rq, P.goal_quat, cur, sn_h, ac_h);
  if (sp_out) { sp_out[0] = sp[0]; sp_out[1] = sp[1]; sp_out[2] = sp[2]; }
  if (smat_out) {
#pragma unroll
    for (int i = 0; i < 9; i++) smat_out[i] = smat[i];
  }
  ft[0] = sp[0] - P.goal_pos[0]; ft[1] = sp[1] - P.goal_pos[1]; ft[2] = sp[2] - P.goal_pos[2];
  ft[3] = rq[0] * m->ik_res_rad; ft[4] = rq[1] * m->ik_res_rad; ft[5] = rq[2] * m->ik_res_rad;
  if (JAC) {
    // mjd_subQuat: Da = I + h K + (1 - h / tan h) K^2 (K = [axs]x, |axs| = 1, so K^2 v = axs (axs . v) - v), D_ee = -Da^T;
    // J_orn column = rad * D_ee^T * site_xmat^T * axis = -rad * Da * (site_xmat^T axis): applied to the one vector this lane
    // needs (27 operations) instead of forming Da and the 3 x 3 product first (~95)
    real axs[3] = {rq[0], rq[1], rq[2]};
    real half = 0.5 * normalize3_fast(axs);
    real coef = 1.0 - (half < 6e-8 ? 1.0 : half * ac_h / sn_h);      // half / tan(half), tan(half) = sn_h / ac_h
    // (P.cslide: the joint type of THIS lane's chain link, read once per problem by coop_chain_setup -- lanes with an unknown
    // have c < clen, so it is their link's; round 5 re-read chain_link -> jnt_type, two dependent global loads, in every evaluation)
    if (!P.on) { Jc[0] = 0; Jc[1] = 0; Jc[2] = 0; Jc[3] = 0; Jc[4] = 0; Jc[5] = 0; }
    else if (P.cslide) { Jc[0] = axw[0]; Jc[1] = axw[1]; Jc[2] = axw[2]; Jc[3] = 0; Jc[4] = 0; Jc[5] = 0; }
    else {
      real r[3] = {sp[0] - anc[0], sp[1] - anc[1], sp[2] - anc[2]}, jp[3];
      cross3(jp, axw, r);
      Jc[0] = jp[0]; Jc[1] = jp[1]; Jc[2] = jp[2];
      real uu[3], cx[3];
      matT_vec3(uu, smat, axw);
      cross3(cx, axs, uu);
      const real du = dot3(axs, uu);
#pragma unroll
      for (int i = 0; i < 3; i++) Jc[3 + i] = -m->ik_jac_rad * ((uu[i] + half * cx[i]) + coef * (axs[i] * du - uu[i]));
    }
  }
}

// cost = 0.5 |f|^2 (task + both regulariser blocks, ik_mujoco.py:48-53)
template <int N>
__device__ __forceinline__ real coop_cost(const CoopCtx<N>& P, real x, const real* ft) {
  real a = P.m->ik_res_reg_prev * (x - P.q_prev), b = P.m->ik_res_reg_home * (x - P.q_home);
  real loc = P.on ? a * a + b * b : 0.0;
  real s = gsum8(loc);
#pragma unroll
  for (int r = 0; r < 6; r++) s += ft[r] * ft[r];
  return 0.5 * s;
}
template <int N>
__device__ __forceinline__ real coop_grad(const CoopCtx<N>& P, real x, const real* ft, const real* Jc) {
  if (!P.on) return 0.0;
  real s = 0;
#pragma unroll
  for (int r = 0; r < 6; r++) s += Jc[r] * ft[r];
  return s + P.m->ik_jac_reg * (P.m->ik_res_reg_prev * (x - P.q_prev) + P.m->ik_res_reg_home * (x - P.q_home));
}

// ---- trust-region subproblem, cooperative over the problem's 8 lanes (round 3; round 2 factorised the 7 x 7 redundantly in
// every lane from an LDS copy: 77 doubles of registers per lane, and the reason k_step needed > 256 of them).
// Lane c holds ROW c of the symmetric normal matrix (arow) and, after a factorisation, row c of L (h[j], j <= c), row c of
// L^T (ut[j] = L[j][c], j > c) and 1 / L[c][c].  Column k of L reaches the other lanes by row broadcasts folded into the FMAs;
// lane k picks its row of L^T up from the same broadcasts (ut[j] += bcast_j(l) * [c == k]), which makes BOTH triangular
// solves column-oriented: one multiply and one broadcast-FMA per pivot, no lane reductions.  Lanes >= N carry zero rows and
// stay inert.
template <int N> struct TrFac { real h[N], ut[N], invd; };
// e[j] = [c == j] as a double (built once per IK call): the diagonal shift, the lane's own 1 / L[c][c] and its row of L^T are
// then FMAs against it instead of a compare and two selects each
template <int N>
__device__ __forceinline__ bool chol_coop(const real (&arow)[N], const real (&e)[N], real alpha, int c, TrFac<N>& F) {
  bool ok = true;
#pragma unroll
  for (int j = 0; j < N; j++) { F.h[j] = __builtin_fma(e[j], alpha, arow[j]); F.ut[j] = 0; }
  F.invd = 0;
  static_for<0, N>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    const real dk = bcast8<k>(F.h[k]);
    ok = ok && dk > 0;                                 // (uniform over the problem: dk is a broadcast; a failed factor is never used)
    const real inv = rsqrt_nr(dk);
    const real lik = c > k ? F.h[k] * inv : 0.0;       // strictly-lower part: rows on and above the pivot hold exact zeros
    F.h[k] = lik;
    F.invd = __builtin_fma(e[k], inv, F.invd);
    static_for<k + 1, N>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      dppfma_pn<j, j == k + 1>(F.ut[j], lik, e[k], F.h[j], lik, lik);     // ut[j] += L[j][k] [c == k];  h[j] -= L[j][k] L[c][k]
    });
    if constexpr (k + 2 >= N && k + 1 < N) dpp_settle(F.h[k + 1]);     // the next pivot's broadcast reads what the last run just wrote
  });
  return ok;
}
// z = L^-1 b (b, z distributed one component per lane)
template <int N>
__device__ __forceinline__ real tr_fwd(const TrFac<N>& F, real b) {
  static_for<0, N>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    const real t = b * F.invd;                         // lane k's t is z_k (its b is final: h[j] = 0 for j >= c)
    fnmac_bcast16<k>(b, t, F.h[k]);
  });
  return b * F.invd;
}
// x = L^-T z
template <int N>
__device__ __forceinline__ real tr_bwd(const TrFac<N>& F, real z) {
  static_for<0, N>([&](auto kc) {
    constexpr int k = N - 1 - decltype(kc)::value;
    const real t = z * F.invd;                         // lane k's t is x_k (ut[j] = 0 for j <= c: later pivots leave it alone)
    fnmac_bcast16<k>(z, t, F.ut[k]);
  });
  return z * F.invd;
}

// ---- Round 5: the same factorisation with 2 x 2 BLOCK pivots (A = L D L^T, L unit block-lower-triangular, D block-diagonal).
// A lone IK problem is a chain of dependent operations on 8 lanes, and a straggler (35-45 evaluations, one wave in ~100, which then
// ends its launch) spends most of them in More's secular iteration: one factorisation and three triangular solves per trial
// alpha.  The scalar Cholesky pays one reciprocal-square-root chain per pivot and two dependent operations per pivot and solve;
// here a block costs ONE reciprocal (of its determinant) and ONE run of two broadcast-FMAs per solve -- 4 sequential steps for
// n = 7 instead of 7.  Lane c keeps row c of L (h[j], j below its block), row c of L^T (ut[j]) and its row of D^-1 of its own
// block (dd on the diagonal, dx towards the block partner c ^ 1).  The block's determinant and the multipliers' numerators are
// compensated differences of products: a block of two nearly dependent Jacobian columns (the Torso's usual case) cancels most of
// their leading digits, and with plain FMAs the worst one-step deviation from the oracle over 0.8 M Torso samples grew from 1e-11
// to 8e-9 rad (still inside the IK bars; with the compensation it is 4e-12 again).  Measured on one box, same run: k_step 0.5942 ms
// with the scalar Cholesky (-DKM_IK_CHOLESKY), 0.5917 ms with the blocks -- the IK is a fifth of a wave and the factorisation a
// fifth of the IK.
template <int N> struct TrFacB { real h[N], ut[N], dd, dx; };
template <int N>
__device__ __forceinline__ bool ldl_coop(const real (&arow)[N], const real (&e)[N], real alpha, int c, TrFacB<N>& F) {
  bool ok = true;
#pragma unroll
  for (int j = 0; j < N; j++) { F.h[j] = __builtin_fma(e[j], alpha, arow[j]); F.ut[j] = 0; }
  F.dd = 0; F.dx = 0;
  static_for<0, N / 2>([&](auto kc) {
    constexpr int p = 2 * decltype(kc)::value, q = p + 1;
    const real a = bcast8<p>(F.h[p]), b = bcast8<q>(F.h[p]), cc = bcast8<q>(F.h[q]);
    // a cc - b^2 by Kahan's difference of products (the rounding error of b * b recovered with one FMA): an ill-conditioned block --
    // two nearly dependent Jacobian columns, the Torso's usual case -- cancels most of the leading digits here
    const real bb = b * b, det = __builtin_fma(a, cc, -bb) - __builtin_fma(b, b, -bb);
    ok = ok && a > 0 && det > 0;                          // (uniform over the problem; a failed factor is never used)
    const real rdet = frcp(det);
    const real ia = cc * rdet, ib = -(b * rdet), ic = a * rdet;            // D^-1 of the block
    F.dd = __builtin_fma(e[p], ia, __builtin_fma(e[q], ic, F.dd));
    F.dx = __builtin_fma(e[p] + e[q], ib, F.dx);
    // rows below the block: their two multipliers (h_p cc - h_q b) / det and (a h_q - b h_p) / det, the numerators again as
    // differences of products with the rounding error of the subtracted product recovered (the entries of D^-1 of a nearly
    // singular block are huge and of opposite sign: h_p ia + h_q ib would cancel); rows on and above the block keep exact zeros
    const real hp = F.h[p], hq = F.h[q];
    const real m1 = hq * b, n1 = __builtin_fma(hp, cc, -m1) - __builtin_fma(hq, b, -m1);
    const real m2 = b * hp, n2 = __builtin_fma(a, hq, -m2) - __builtin_fma(b, hp, -m2);
    const real l1 = c > q ? n1 * rdet : 0.0, l2 = c > q ? n2 * rdet : 0.0;
    static_for<q + 1, N>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      // h[j] -= A[j][p] l1 + A[j][q] l2 (the UNSCALED entries of lane j: h[p], h[q] are overwritten after the runs);
      // ut[j] += L[j][p] [c == p] + L[j][q] [c == q]
      dppfma_acc2n<j, j == q + 1>(F.h[j], F.h[p], l1, F.h[q], l2);
      dppfma_acc2<j, false>(F.ut[j], l1, e[p], l2, e[q]);
    });
    F.h[p] = l1; F.h[q] = l2;
    // the next block's broadcasts (compiler-generated DPP moves) read what the last runs just wrote
    if constexpr (q + 1 < N) dpp_settle(F.h[q + 1]);
    if constexpr (q + 2 < N) dpp_settle(F.h[q + 2]);
  });
  if constexpr (N & 1) {
    constexpr int r = N - 1;
    const real d = bcast8<r>(F.h[r]);
    ok = ok && d > 0;
    F.dd = __builtin_fma(e[r], frcp(d), F.dd);
    F.h[r] = 0;
  }
  return ok;
}
// z = L^-1 b (unit block-lower-triangular: a block's two components are final when the blocks before it are done)
template <int N>
__device__ __forceinline__ real ldl_fwd(const TrFacB<N>& F, real b) {
  static_for<0, N / 2>([&](auto kc) {
    constexpr int p = 2 * decltype(kc)::value, q = p + 1;
    if constexpr (q + 1 < N) { const real zb = b; dppfma_row2n<p, q>(b, zb, F.h[p], F.h[q]); }      // rows below: b -= z_p L[.][p] + z_q L[.][q]
  });
  return b;
}
// w = D^-1 z, all blocks at once: the block partner's component arrives by a quad_perm swap of neighbouring lanes
template <int N>
__device__ __forceinline__ real ldl_mid(const TrFacB<N>& F, real z) {
  const real zp = dpp_f64<0xB1>(z);                      // quad_perm [1,0,3,2]: lane c ^ 1
  return __builtin_fma(F.dd, z, F.dx * zp);
}
// x = L^-T w: from the last pivot up; rows above a block subtract its two components through their row of L^T
template <int N>
__device__ __forceinline__ real ldl_bwd(const TrFacB<N>& F, real x) {
  if constexpr (N & 1) { const real xb = x; dppfma1<true, N - 1>(x, xb, F.ut[N - 1]); }
  static_for<0, N / 2>([&](auto kc) {
    constexpr int k = N / 2 - 1 - decltype(kc)::value, p = 2 * k, q = p + 1;
    if constexpr (k > 0) { const real xb = x; dppfma_row2n<p, q>(x, xb, F.ut[p], F.ut[q]); }
  });
  return x;
}

// scipy common.py solve_lsq_trust_region on the normal matrix: p = argmin of the model in the ball |p| <= Delta by More's
// iteration on the secular equation, phi = |p(alpha)| - Delta, phi' = -|L^-1 p|^2 / |p| (SciPy evaluates the same two
// numbers from its SVD).  g_h, p: this lane's components.
template <int N>
__device__ __forceinline__ void solve_tr_coop(const real (&arow)[N], const real (&e)[N], int c, real g_h, real Delta, real& alpha, real& p) {
  const real ng = -g_h;
#ifdef KM_IK_CHOLESKY      // A/B build: the scalar Cholesky of rounds 3-4 (chol_coop / tr_fwd / tr_bwd above)
  TrFac<N> F;
  auto factor = [&](real al) { return chol_coop<N>(arow, e, al, c, F); };
  auto solve = [&](real rhs) { return tr_bwd<N>(F, tr_fwd<N>(F, rhs)); };
  auto ainv_norm2 = [&](real v) { const real q = tr_fwd<N>(F, v); return gsum8(q * q); };
#else
  TrFacB<N> F;
  auto factor = [&](real al) { return ldl_coop<N>(arow, e, al, c, F); };
  // |L_chol^-1 v|^2 = v^T A^-1 v = z^T D^-1 z with z = L^-1 v (what SciPy reads off its SVD as phi')
  auto solve = [&](real rhs) { return ldl_bwd<N>(F, ldl_mid<N>(F, ldl_fwd<N>(F, rhs))); };
  auto ainv_norm2 = [&](real v) { const real z = ldl_fwd<N>(F, v); return gsum8(z * ldl_mid<N>(F, z)); };
#endif
  const bool full_rank = factor(0.0);
  real pn = 0;
  if (full_rank) {
    p = solve(ng);
    pn = km_sqrt(gsum8(p * p));
    if (pn <= Delta) { alpha = 0.0; return; }
  }
  const real iDelta = frcp(Delta);
  real alpha_upper = km_sqrt(gsum8(g_h * g_h)) * iDelta, alpha_lower = 0.0;
  if (full_rank) {
    const real phi = pn - Delta, phip = -ainv_norm2(p) * frcp(pn);
    alpha_lower = -phi * frcp(phip);
  }
  if (!full_rank && alpha == 0) alpha = fmax(0.001 * alpha_upper, sqrt(alpha_lower * alpha_upper));
  for (int it = 0; it < 10; it++) {
    if (alpha < alpha_lower || alpha > alpha_upper) alpha = fmax(0.001 * alpha_upper, sqrt(alpha_lower * alpha_upper));
    factor(alpha);
    p = solve(ng);
    pn = km_sqrt(gsum8(p * p));
    const real phi = pn - Delta, phip = -ainv_norm2(p) * frcp(pn);
    if (phi < 0) alpha_upper = alpha;
    const real ratio = phi * frcp(phip);
    alpha_lower = fmax(alpha_lower, alpha - ratio);
    alpha -= (phi + Delta) * ratio * iDelta;
    if (fabs(phi) < 0.01 * Delta) break;
  }
  factor(alpha);
  p = solve(ng);
  p *= Delta * rsqrt_nr(gsum8(p * p));
}

__device__ __forceinline__ void min_quad_1d_c(real a, real b, real lo, real hi, real c, real& t_out, real& y_out) {
  real tb = lo, yb = lo * (a * lo + b) + c;
  real y1 = hi * (a * hi + b) + c;
  if (y1 < yb) { yb = y1; tb = hi; }
  if (a != 0) {
    real ex = -0.5 * b / a;
    if (lo < ex && ex < hi) { real y2 = ex * (a * ex + b) + c; if (y2 < yb) { yb = y2; tb = ex; } }
  }
  t_out = tb; y_out = yb;
}

// (A v)_c = sum_j A[c][j] v_j: this lane's row of A (registers) against a vector distributed one component per lane
template <int N>
__device__ __forceinline__ real arow_dot(const real (&arow)[N], real v) {
  real s = 0;
  fmac_rowvec<16, 0, N>(s, bsrc<16>(v), [&](int j) { return arow[j]; });
  return s;
}
// step size to the bound along s for this lane (INF if none)
template <int N>
__device__ __forceinline__ real lane_step_to_bound(const CoopCtx<N>& P, real x, real s) {
  const real is = frcp(s);
  return (P.on && s != 0) ? fmax((P.lb - x) * is, (P.ub - x) * is) : INFINITY;
}

// scipy trf.py select_step in per-lane form.  Inputs: this lane's x, d, p_h, g_h and its row of the normal matrix.
// Output: this lane's step_h (step = d * step_h); returns the predicted reduction (uniform).
template <int N>
__device__ __forceinline__ real coop_select_step(const CoopCtx<N>& P, const real (&arow)[N], real x, real d, real ph, real gh,
                                                 real Delta, real theta, real& step_h) {
  const real Aph = arow_dot<N>(arow, ph);
  const real xp = x + d * ph;
  if (gall8(!P.on || (xp >= P.lb && xp <= P.ub))) {
    real s2[2] = {ph * Aph, gh * ph};
    gsum8_n<2>(s2);
    const real pv = 0.5 * s2[0] + s2[1];
    step_h = ph;
    return -pv;
  }
  const real st = lane_step_to_bound<N>(P, x, d * ph);
  const real p_stride = gmin8(st);
  const bool hit = P.on && (d * ph != 0) && st == p_stride;
  real rh = hit ? -ph : ph;
  const real phs = ph * p_stride;            // trust-region step restricted to hit the bound
  const real xb = x + d * phs;
  real to_tr;
  {
    real s3[3] = {rh * rh, phs * rh, phs * phs};
    gsum8_n<3>(s3);
    const real a = s3[0], b = s3[1], c = s3[2] - Delta * Delta;
    const real dd = sqrt(b * b - a * c);
    const real q = -(b + copysign(dd, b));
    const real t1 = q * frcp(a), t2 = c * frcp(q);
    to_tr = t1 < t2 ? t2 : t1;
  }
  const real to_bound = gmin8(lane_step_to_bound<N>(P, xb, d * rh));
  real r_stride = fmin(to_bound, to_tr), rl, ru;
  if (r_stride > 0) { rl = (1 - theta) * p_stride / r_stride; ru = (r_stride == to_bound) ? theta * to_bound : to_tr; }
  else { rl = 0; ru = -1; }
  real r_value = INFINITY;
  const real Aphs = Aph * p_stride;          // A (p_h * stride) row value
  if (rl <= ru) {
    const real Arh = arow_dot<N>(arow, rh);
    real s5[5] = {rh * Arh, gh * rh, phs * Arh, phs * Aphs, gh * phs};
    gsum8_n<5>(s5);
    const real a = 0.5 * s5[0];
    const real b = s5[1] + s5[2];
    const real c = 0.5 * s5[3] + s5[4];
    min_quad_1d_c(a, b, rl, ru, c, r_stride, r_value);
    rh = rh * r_stride + phs;
  }
  const real pht = phs * theta;              // strictly interior version of the restricted step
  real agh = -gh;
  real s3b[3] = {pht * (Aphs * theta), gh * pht, agh * agh};
  gsum8_n<3>(s3b);
  const real p_value = 0.5 * s3b[0] + s3b[1];
  const real to_tr2 = Delta * rsqrt_nr(s3b[2]);
  const real to_bound2 = gmin8(lane_step_to_bound<N>(P, x, d * agh));
  real ag_stride = (to_bound2 < to_tr2) ? theta * to_bound2 : to_tr2;
  real ag_value;
  {
    const real Aag = arow_dot<N>(arow, agh);
    real s2b[2] = {agh * Aag, gh * agh};
    gsum8_n<2>(s2b);
    const real a = 0.5 * s2b[0], b = s2b[1];
    min_quad_1d_c(a, b, 0, ag_stride, 0, ag_stride, ag_value);
  }
  agh *= ag_stride;
  if (p_value < r_value && p_value < ag_value) { step_h = pht; return -p_value; }
  if (r_value < p_value && r_value < ag_value) { step_h = rh; return -r_value; }
  step_h = agh;
  return -ag_value;
}

// make_strictly_feasible for this lane's component
__device__ __forceinline__ real lane_strictly_feasible(real x, real lb, real ub, real rstep) {
  real xn = x;
  if (rstep == 0) {
    if (x <= lb) xn = nextafter(lb, ub);
    if (x >= ub) xn = nextafter(ub, lb);
  } else {
    real ld = x - lb, ud = ub - x;
    real lt = rstep * fmax(1.0, fabs(lb)), ut = rstep * fmax(1.0, fabs(ub));
    if (ld <= fmin(ud, lt)) xn = lb + lt;
    if (ud <= fmin(ld, ut)) xn = ub - ut;
  }
  if (xn < lb || xn > ub) xn = 0.5 * (lb + ub);
  return xn;
}

// scipy trf.py trf_bounds, cooperative form.  x: this lane's unknown (in: strictly feasible start, out: result.x);
// x_last: the last evaluated point.  Returns status (uniform).
template <int N>
__device__ int coop_trf(const CoopCtx<N>& P, real& x, real& x_last, int* nfev_out) {
  const real ftol = 1e-8, xtol = 1e-8, gtol = 1e-8;
  const int max_nfev = P.m->ik_max_nfev > 0 ? P.m->ik_max_nfev : 100 * N;      // (KModelDesc::ik_max_nfev: opt-in cap, 0 = SciPy's default)
  const real jreg2 = 2 * P.m->ik_jac_reg * P.m->ik_jac_reg;
  real ft[6], Jc[6], ft_new[6], Jn[6], e[N];
#pragma unroll
  for (int j = 0; j < N; j++) e[j] = P.c == j ? 1.0 : 0.0;
  coop_eval<N, true>(P, x, ft, Jc, nullptr, nullptr);
  P.pf->ph(33);
  int nfev = 1;
  real cost = coop_cost<N>(P, x, ft);
  real g = coop_grad<N>(P, x, ft, Jc);
  real v = 1;
  if (g < 0) v = P.ub - x;
  if (g > 0) v = x - P.lb;
  real Delta = km_sqrt(gsum8(P.on ? x * x / v : 0.0));
  if (Delta == 0) Delta = 1.0;
  real alpha = 0.0, cost_new = cost;
  int status = -1;
  x_last = x;
  for (;;) {
    real dv = 0;
    v = 1;
    if (g < 0) { v = P.ub - x; dv = -1; }
    if (g > 0) { v = x - P.lb; dv = 1; }
    const real g_norm = gmax8(P.on ? fabs(g * v) : 0.0);
    if (g_norm < gtol) status = 1;
    if (status != -1 || nfev == max_nfev) break;
    const real d = km_sqrt(v), diag_h = g * dv, g_h = d * g;
    // ---- normal matrix row c: A[c][j] = d_c d_j (J_c . J_j + 2 reg^2 [c==j]) + diag_h [c==j], in registers.  The other lanes'
    // Jacobian columns and scalings arrive by row broadcasts folded into the FMAs (no LDS, no synchronisation); the matrix is
    // symmetric, so the lane's row is also its column -- all the trust-region solve and the step selection need.
    real arow[N];
    static_for<0, N>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      real s = 0;
      dppfma_acc3<j>(s, Jc[0], Jc[0], Jc[1], Jc[1], Jc[2], Jc[2]);
      dppfma_acc3<j>(s, Jc[3], Jc[3], Jc[4], Jc[4], Jc[5], Jc[5]);
      s = __builtin_fma(e[j], jreg2, s);
      s *= d * bcast8<j>(d);
      arow[j] = P.on ? __builtin_fma(e[j], diag_h, s) : 0.0;
    });
    P.pf->ph(34);
    const real theta = fmax(0.995, 1 - g_norm);
    real actual = -1, x_new = x;
    while (actual <= 0 && nfev < max_nfev) {
      real ph = 0;
      solve_tr_coop<N>(arow, e, P.c, g_h, Delta, alpha, ph);
      P.pf->ph(35);
      real step_h;
      const real predicted = coop_select_step<N>(P, arow, x, d, ph, g_h, Delta, theta, step_h);
      if (!P.on) step_h = 0;
      P.pf->ph(36);
      const real step = d * step_h;
      x_new = P.on ? lane_strictly_feasible(x + step, P.lb, P.ub, 0.0) : x;
      // ik_res(x_new) and, in the same kinematics pass, what ik_jac(x_new) recomputes if the step is accepted
      coop_eval<N, true>(P, x_new, ft_new, Jn, nullptr, nullptr);
      P.pf->ph(33);
      x_last = x_new;
      nfev++;
      real n3[3] = {step_h * step_h, step * step, P.on ? x * x : 0.0};
      gsum8_n<3>(n3);
      const real shn = km_sqrt(n3[0]);
      bool fin = true;
#pragma unroll
      for (int r = 0; r < 6; r++) fin = fin && isfinite(ft_new[r]);
      if (!fin) { Delta = 0.25 * shn; continue; }
      cost_new = coop_cost<N>(P, x_new, ft_new);
      actual = cost - cost_new;
      real ratio, Delta_new = Delta;
      if (predicted > 0) ratio = actual * frcp(predicted);
      else if (predicted == 0 && actual == 0) ratio = 1;
      else ratio = 0;
      if (ratio < 0.25) Delta_new = 0.25 * shn;
      else if (ratio > 0.75 && shn > 0.95 * Delta) Delta_new = Delta * 2.0;
      const real sn = km_sqrt(n3[1]), xn = km_sqrt(n3[2]);
      const bool ft_ok = (actual < ftol * cost) && (ratio > 0.25);
      const bool xt_ok = sn < xtol * (xtol + xn);
      if (ft_ok && xt_ok) status = 4; else if (ft_ok) status = 2; else if (xt_ok) status = 3;
      if (status != -1) break;
      alpha *= Delta * frcp(Delta_new);
      Delta = Delta_new;
      P.pf->ph(37);
    }
    if (actual > 0) {
      x = x_new;
      cost = cost_new;
#pragma unroll
      for (int r = 0; r < 6; r++) { ft[r] = ft_new[r]; Jc[r] = Jn[r]; }
      g = coop_grad<N>(P, x, ft, Jc);
    }
  }
  if (status == -1) status = 0;
  *nfev_out = nfev;
  return status;
}

// ik_mujoco.py:100-155 for one problem; returns q_out component of this lane
template <int N>
__device__ __forceinline__ real coop_ik_solve(const CoopCtx<N>& P, real x0, real& x_last, int* nfev, int* status) {
  real x = x0;
  x_last = x0;
  *nfev = 0; *status = -2;
  if (gall8(!P.on || (x >= P.lb && x <= P.ub))) {      // else least_squares raises ValueError -> "IK failed"
    if (P.on) x = lane_strictly_feasible(x, P.lb, P.ub, 1e-10);
    *status = coop_trf<N>(P, x, x_last, nfev);
  }
  return fmin(fmax(x, P.lb), P.ub);                      // :147-152 (:140-145 is a no-op)
}

__device__ __forceinline__ real f32r_c(real x) { return (real)(float)x; }

// ---------------------------------------------------------------------------------------------
// KManipTask.before_step (env_sim.py:38-108) for ONE (env, arm) problem on its 8 lanes: grip decode, EE-delta decode
// + IK, or the joint-delta modes.  IO abstracts where the env's state lives (global SoA columns for the
// stand-alone kernel, the LDS workspace when fused into k_step):
//   real IO::qpos(int i); void IO::set_ctrl(int i, real v); void IO::set_qpos_ik(int i, real v); void IO::set_diag(int arm, int nfev, int status)
template <int N, class IO>
__device__ __forceinline__ void coop_before_step(const KDeviceModel* dm, int arm, int c, const float* a, IO& io, Prof* pf) {
  const KModelDesc* m = &dm->d;
  const int grip_key[2] = {KM_ACT_GRIP_R, KM_ACT_GRIP_L};
  const int pos_key[2] = {KM_ACT_EER_POS, KM_ACT_EEL_POS};
  const int orn_key[2] = {KM_ACT_EER_ORN, KM_ACT_EEL_ORN};
  const int qp_key[2] = {KM_ACT_QPOS_R, KM_ACT_QPOS_L};
  // ---- grip (env_sim.py:41-59): float32 arithmetic exactly as numpy does it
  const int cg = m->act_col[grip_key[arm]];
  if (cg >= 0 && c == 0) {
    int g0 = m->arm_grip_id[arm][0], g1 = m->arm_grip_id[arm][1];
    float g = a[cg] * (float)m->ee_s_delta;
    g = (float)((double)g + io.qpos(g0));
    g = fminf(fmaxf(g, (float)m->ee_s_min), (float)m->ee_s_max);
    io.set_ctrl(g0, (double)g);
    io.set_ctrl(g1, (double)g);
  }
  CoopCtx<N> P;
  P.m = m; P.ax = &dm->x; P.arm = arm; P.c = c; P.on = c < N; P.pf = pf;
  coop_chain_setup<N>(P);
  const int q = m->arm_q_id[arm][P.on ? c : 0];
  const real x0 = P.on ? io.qpos(q) : 0.0;
  P.q_prev = x0; P.q_home = m->q_home[q]; P.lb = m->jnt_range[q][0]; P.ub = m->jnt_range[q][1];
  P.qfix = (P.clen > N) ? io.qpos(dm->x.chain_link[arm][P.clen - 1]) : 0.0;
  const int cp = m->act_col[pos_key[arm]], co = m->act_col[orn_key[arm]], cq = m->act_col[qp_key[arm]];
  if (cp >= 0) {
    // current site pose: one kinematics pass at x0
    P.goal_pos[0] = 0; P.goal_pos[1] = 0; P.goal_pos[2] = 0;
    P.goal_quat[0] = 1; P.goal_quat[1] = 0; P.goal_quat[2] = 0; P.goal_quat[3] = 0;
    real ft0[6], sp[3], smat[9];
    coop_eval<N, false>(P, x0, ft0, nullptr, sp, smat);
    // EE-delta decode (env_sim.py:60-69): euler("xyz", extrinsic) of the site matrix + delta -> quaternion
    real e0 = km_atan2(smat[7], smat[8]);
    real e1 = km_atan2(-smat[6], km_sqrt(smat[7] * smat[7] + smat[8] * smat[8]));
    real e2 = km_atan2(smat[3], smat[0]);
    e0 += (double)a[co] * m->ee_orn_delta[0];
    e1 += (double)a[co + 1] * m->ee_orn_delta[1];
    e2 += (double)a[co + 2] * m->ee_orn_delta[2];
    real s0, c0, s1, c1, s2, c2;
    km_sincos(e0 * 0.5, &s0, &c0); km_sincos(e1 * 0.5, &s1, &c1); km_sincos(e2 * 0.5, &s2, &c2);
    real qx[4] = {c0, s0, 0, 0}, qy[4] = {c1, 0, s1, 0};
    real qz[4] = {c2, 0, 0, s2}, t4[4];
    qmul(t4, qy, qx);
    qmul(P.goal_quat, qz, t4);
    P.goal_pos[0] = (double)a[cp] * m->ee_pos_delta[0] + sp[0];
    P.goal_pos[1] = (double)a[cp + 1] * m->ee_pos_delta[1] + sp[1];
    P.goal_pos[2] = (double)a[cp + 2] * m->ee_pos_delta[2] + sp[2];
    real xl;
    int nfev, status;
    const real qo = coop_ik_solve<N>(P, x0, xl, &nfev, &status);
    if (P.on) { io.set_ctrl(q, f32r_c(qo)); io.set_qpos_ik(q, xl); }
    if (c == 0) io.set_diag(arm, nfev, status);
  } else {
    if (c == 0) io.set_diag(arm, 0, -3);
    if (cq >= 0 && P.on)    // joint-delta modes, env_sim.py:100-103
      io.set_ctrl(q, f32r_c(x0 + (double)(a[cq + c] * (float)m->q_pos_delta)));
  }
}
