// kmanip_ik.hip -- batched action decode + bounded trust-region-reflective IK on gfx950.
//
// Replaces, for all envs at once, the first half of KManipTask.before_step (reference
// gym_kmanip/env_sim.py:38-99): grip decode (:41-59), EE-delta decode (:60-70, :80-90), the IK
// (gym_kmanip/ik_mujoco.py:100-155 -> ik_res :20-53, ik_jac :56-97, scipy least_squares :129-135),
// joint-delta modes (:100-103) and the float32 ctrl round trip (:40,106-108).
//
// Mapping: one lane per (env, arm).  The whole TRF state (7 unknowns, the 6x7 task Jacobian, the 7x7
// normal matrix and its Cholesky factor) lives in registers with compile-time indexing -- no LDS, no
// scratch; state is read/written as coalesced struct-of-arrays columns.  The regulariser rows of the
// reference Jacobian (9e-3 * I twice, ik_mujoco.py:92-97) are never materialised.
// The trust-region subproblem min |J_h p + f|, |p| <= Delta is solved exactly like SciPy's
// solve_lsq_trust_region (More's iteration on the secular equation) but through Cholesky factors of
// (J_h^T J_h + C + alpha I) instead of an SVD: same alpha sequence in exact arithmetic.
#include "kmanip_device.hpp"

#define DBL_EPS 2.220446049250313e-16
// ---- optional phase profiler (diagnostic build only: make prof -> -DKM_PROFILE); see kmanip_dyn.hip
#define KM_NPH_IK 8
#ifdef KM_PROFILE
__device__ unsigned long long g_prof_ik[KM_NPH_IK];
struct ProfIk {
  unsigned long long t0, acc[KM_NPH_IK];
  __device__ __forceinline__ void start() { for (int i = 0; i < KM_NPH_IK; i++) acc[i] = 0; t0 = __builtin_amdgcn_s_memtime(); }
  __device__ __forceinline__ void ph(int i) {
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t = __builtin_amdgcn_s_memtime();
    acc[i] += t - t0; t0 = t;
    __builtin_amdgcn_sched_barrier(0);
  }
  __device__ __forceinline__ void flush() { if ((threadIdx.x & 63) == 0) for (int i = 0; i < KM_NPH_IK; i++) atomicAdd(&g_prof_ik[i], acc[i]); }
};
extern "C" int kmanip_dbg_prof_ik(unsigned long long* out, int reset) {
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof_ik), sizeof(unsigned long long) * KM_NPH_IK) != hipSuccess) return -1;
  if (reset) { unsigned long long z[KM_NPH_IK] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_prof_ik), z, sizeof z) != hipSuccess) return -1; }
  return 0;
}
#else
struct ProfIk {
  __device__ __forceinline__ void start() {}
  // in the product build a phase boundary is just a scheduling fence: it keeps the (fully unrolled) phases from
  // being interleaved, which is what drove this kernel into scratch spills
  __device__ __forceinline__ void ph(int) { __builtin_amdgcn_sched_barrier(0); }
  __device__ __forceinline__ void flush() {}
};
#endif
#ifndef IK_LANE_STRIDE
#define IK_LANE_STRIDE 4
#endif

// R x C matrix living in LDS, one instance per IK problem of the workgroup: element (r, c) of problem `slot`
// is at ((r * C + c) * stride + slot), i.e. consecutive problems hit consecutive banks.
template <int R, int C>
struct LMat {
  real* p;
  int stride;
  __device__ __forceinline__ real& operator()(int r, int c) const { return p[(r * C + c) * stride]; }
};
template <int N>
struct IkLds {          // the big per-problem matrices kept out of the register file
  LMat<6, N> J, J2;     // task Jacobian at x, and at the trial point x_new (swapped in when the step is accepted)
  LMat<N, N> A, L;      // normal matrix J_h^T J_h + C and the Cholesky factor of (A + alpha I)
  static constexpr int WORDS = 12 * N + 2 * N * N;
  __device__ __forceinline__ IkLds(real* base, int slot, int nslots)
      : J{base + slot, nslots}, J2{base + 6 * N * nslots + slot, nslots}, A{base + 12 * N * nslots + slot, nslots},
        L{base + (12 * N + N * N) * nslots + slot, nslots} {}
};

template <int N>
struct IkCtx {
  const KModelDesc* m;
  const KModelAux* ax;
  int arm;
  real qfix[KM_MAX_CHAIN];   // joint values of chain links that are not IK unknowns
  real goal_pos[3], goal_quat[4];
  real q_prev[N], q_home[N], lb[N], ub[N];
};

template <int N>
struct IkEval {
  real ft[6];        // task residual (pos, rad * subQuat)
  LMat<6, N> Jt;     // task Jacobian (LDS view)
  real sp[3], smat[9];  // site position / rotation at the evaluated point
};

// forward kinematics along the chain + residual (+ Jacobian): ik_res / ik_jac
template <int N, bool JAC>
__device__ __forceinline__ void ik_eval(const IkCtx<N>& P, const real* x, IkEval<N>& E) {
  const KModelDesc* m = P.m;
  const int arm = P.arm;
  // Rotation-matrix propagation (all joints rotate/slide about local z): per link one constant 3x3 product
  // and a planar rotation of two columns -- no quaternion normalisations, one sincos per hinge.
  real pos[3] = {0, 0, 0}, mat[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  real anc[N][3], axw[N][3];
  const int clen = P.ax->chain_len[arm];
#pragma unroll
  for (int k = 0; k < KM_MAX_CHAIN; k++) {
    if (k < clen) {
      const int l = P.ax->chain_link[arm][k];
      const double* Rl = P.ax->chain_R[arm][k];
      real lp[3] = {m->link_pos[l][0], m->link_pos[l][1], m->link_pos[l][2]};
      real t[3], R1[9];
      mat_vec3(t, mat, lp);
      pos[0] += t[0]; pos[1] += t[1]; pos[2] += t[2];
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) R1[3 * i + j] = mat[3 * i] * Rl[j] + mat[3 * i + 1] * Rl[3 + j] + mat[3 * i + 2] * Rl[6 + j];
      const real qv = (k < N) ? x[k < N ? k : 0] : P.qfix[k];
      if (m->jnt_type[l] == KM_JNT_SLIDE) {
#pragma unroll
        for (int i = 0; i < 9; i++) mat[i] = R1[i];
        pos[0] += R1[2] * qv; pos[1] += R1[5] * qv; pos[2] += R1[8] * qv;
      } else {
        real sn, cs;
        sincos(qv, &sn, &cs);
#pragma unroll
        for (int i = 0; i < 3; i++) {
          mat[3 * i] = cs * R1[3 * i] + sn * R1[3 * i + 1];
          mat[3 * i + 1] = cs * R1[3 * i + 1] - sn * R1[3 * i];
          mat[3 * i + 2] = R1[3 * i + 2];
        }
      }
      if (JAC && k < N) {
        anc[k < N ? k : 0][0] = pos[0]; anc[k < N ? k : 0][1] = pos[1]; anc[k < N ? k : 0][2] = pos[2];
        axw[k < N ? k : 0][0] = mat[2]; axw[k < N ? k : 0][1] = mat[5]; axw[k < N ? k : 0][2] = mat[8];
      }
    }
  }
  // site pose
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
  sub_quat(rq, P.goal_quat, cur);
  E.sp[0] = sp[0]; E.sp[1] = sp[1]; E.sp[2] = sp[2];
#pragma unroll
  for (int i = 0; i < 9; i++) E.smat[i] = smat[i];
  E.ft[0] = sp[0] - P.goal_pos[0]; E.ft[1] = sp[1] - P.goal_pos[1]; E.ft[2] = sp[2] - P.goal_pos[2];
  E.ft[3] = rq[0] * m->ik_res_rad; E.ft[4] = rq[1] * m->ik_res_rad; E.ft[5] = rq[2] * m->ik_res_rad;
  if (JAC) {
    // mjd_subQuat: Da = I + h K + (1 - h / tan h) K^2, D_ee = Db = -Da^T; mat = rad * D_ee^T * site_xmat^T
    real axs[3] = {rq[0], rq[1], rq[2]};
    real half = 0.5 * normalize3(axs);
    real K[9] = {0, -axs[2], axs[1], axs[2], 0, -axs[0], -axs[1], axs[0], 0};
    real coef = 1.0 - (half < 6e-8 ? 1.0 : half / tan(half));
    real Da[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) {
        real kk = K[3 * i] * K[j] + K[3 * i + 1] * K[3 + j] + K[3 * i + 2] * K[6 + j];
        Da[3 * i + j] = (i == j ? 1.0 : 0.0) + half * K[3 * i + j] + coef * kk;
      }
    real T[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++)
        T[3 * i + j] = -m->ik_jac_rad * (Da[3 * i] * smat[3 * j] + Da[3 * i + 1] * smat[3 * j + 1] + Da[3 * i + 2] * smat[3 * j + 2]);
#pragma unroll
    for (int c = 0; c < N; c++) {
      const int l = P.ax->chain_link[arm][c];
      if (m->jnt_type[l] == KM_JNT_SLIDE) {
        E.Jt(0, c) = axw[c][0]; E.Jt(1, c) = axw[c][1]; E.Jt(2, c) = axw[c][2];
        E.Jt(3, c) = 0; E.Jt(4, c) = 0; E.Jt(5, c) = 0;
      } else {
        real r[3] = {sp[0] - anc[c][0], sp[1] - anc[c][1], sp[2] - anc[c][2]}, jp[3];
        cross3(jp, axw[c], r);
        E.Jt(0, c) = jp[0]; E.Jt(1, c) = jp[1]; E.Jt(2, c) = jp[2];
        E.Jt(3, c) = T[0] * axw[c][0] + T[1] * axw[c][1] + T[2] * axw[c][2];
        E.Jt(4, c) = T[3] * axw[c][0] + T[4] * axw[c][1] + T[5] * axw[c][2];
        E.Jt(5, c) = T[6] * axw[c][0] + T[7] * axw[c][1] + T[8] * axw[c][2];
      }
    }
  }
}

template <int N> __device__ __forceinline__ real vdot(const real* a, const real* b) {
  real s = 0;
#pragma unroll
  for (int i = 0; i < N; i++) s += a[i] * b[i];
  return s;
}
template <int N> __device__ __forceinline__ real vnorm(const real* a) { return sqrt(vdot<N>(a, a)); }

// cost = 0.5 |f|^2 over task + both regulariser blocks (ik_mujoco.py:48-53)
template <int N>
__device__ __forceinline__ real ik_cost(const IkCtx<N>& P, const real* x, const real* ft) {
  real s = 0;
#pragma unroll
  for (int r = 0; r < 6; r++) s += ft[r] * ft[r];
#pragma unroll
  for (int c = 0; c < N; c++) {
    real a = P.m->ik_res_reg_prev * (x[c] - P.q_prev[c]), b = P.m->ik_res_reg_home * (x[c] - P.q_home[c]);
    s += a * a + b * b;
  }
  return 0.5 * s;
}
template <int N>
__device__ __forceinline__ void ik_grad(const IkCtx<N>& P, const real* x, const IkEval<N>& E, real* g) {
#pragma unroll
  for (int c = 0; c < N; c++) {
    real s = 0;
#pragma unroll
    for (int r = 0; r < 6; r++) s += E.Jt(r, c) * E.ft[r];
    s += P.m->ik_jac_reg * (P.m->ik_res_reg_prev * (x[c] - P.q_prev[c]) + P.m->ik_res_reg_home * (x[c] - P.q_home[c]));
    g[c] = s;
  }
}

// lower Cholesky of (A + alpha I) packed full [N][N]; returns false if not numerically SPD
template <int N>
__device__ __forceinline__ bool chol7(LMat<N, N> A, real alpha, LMat<N, N> L) {
  bool ok = true;
#pragma unroll
  for (int j = 0; j < N; j++) {
    real s = A(j, j) + alpha;
#pragma unroll
    for (int k = 0; k < j; k++) s -= L(j, k) * L(j, k);
    if (!(s > 0)) { ok = false; s = 1; }
    real d = sqrt(s), inv = 1.0 / d;
    L(j, j) = d;
#pragma unroll
    for (int i = j + 1; i < N; i++) {
      real t = A(i, j);
#pragma unroll
      for (int k = 0; k < j; k++) t -= L(i, k) * L(j, k);
      L(i, j) = t * inv;
    }
  }
  return ok;
}
template <int N>
__device__ __forceinline__ void chol_solve7(LMat<N, N> L, const real* b, real* x) {
#pragma unroll
  for (int i = 0; i < N; i++) {
    real s = b[i];
#pragma unroll
    for (int k = 0; k < i; k++) s -= L(i, k) * x[k];
    x[i] = s / L(i, i);
  }
#pragma unroll
  for (int i = N - 1; i >= 0; i--) {
    real s = x[i];
#pragma unroll
    for (int k = i + 1; k < N; k++) s -= L(k, i) * x[k];
    x[i] = s / L(i, i);
  }
}
template <int N>
__device__ __forceinline__ real quad_form(LMat<N, N> A, const real* a, const real* b) {
  real s = 0;
#pragma unroll
  for (int i = 0; i < N; i++) {
    real t = 0;
#pragma unroll
    for (int j = 0; j < N; j++) t += A(i, j) * b[j];
    s += a[i] * t;
  }
  return s;
}

// scipy common.py solve_lsq_trust_region restated on the normal matrix A = J_h^T J_h + diag_h
template <int N>
__device__ __forceinline__ void solve_tr(LMat<N, N> A, LMat<N, N> L, const real* g_h, real Delta, real& alpha, real* p) {
  real ng[N], w[N];
#pragma unroll
  for (int i = 0; i < N; i++) ng[i] = -g_h[i];
  bool full_rank = chol7<N>(A, 0.0, L);
  if (full_rank) {
    chol_solve7<N>(L, ng, p);
    if (vnorm<N>(p) <= Delta) { alpha = 0.0; return; }
  }
  real alpha_upper = vnorm<N>(g_h) / Delta, alpha_lower = 0.0;
  if (full_rank) {
    real pn = vnorm<N>(p);
    chol_solve7<N>(L, p, w);
    real phi = pn - Delta, phip = -vdot<N>(p, w) / pn;
    alpha_lower = -phi / phip;
  }
  if (!full_rank && alpha == 0) alpha = fmax(0.001 * alpha_upper, sqrt(alpha_lower * alpha_upper));
  for (int it = 0; it < 10; it++) {
    if (alpha < alpha_lower || alpha > alpha_upper) alpha = fmax(0.001 * alpha_upper, sqrt(alpha_lower * alpha_upper));
    chol7<N>(A, alpha, L);
    chol_solve7<N>(L, ng, p);
    real pn = vnorm<N>(p);
    chol_solve7<N>(L, p, w);
    real phi = pn - Delta, phip = -vdot<N>(p, w) / pn;
    if (phi < 0) alpha_upper = alpha;
    real ratio = phi / phip;
    alpha_lower = fmax(alpha_lower, alpha - ratio);
    alpha -= (phi + Delta) * ratio / Delta;
    if (fabs(phi) < 0.01 * Delta) break;
  }
  chol7<N>(A, alpha, L);
  chol_solve7<N>(L, ng, p);
  real sc = Delta / vnorm<N>(p);
#pragma unroll
  for (int i = 0; i < N; i++) p[i] *= sc;
}

template <int N>
__device__ __forceinline__ bool in_bounds(const real* x, const real* lb, const real* ub) {
  bool ok = true;
#pragma unroll
  for (int i = 0; i < N; i++) ok = ok && (x[i] >= lb[i]) && (x[i] <= ub[i]);
  return ok;
}
template <int N>
__device__ __forceinline__ void make_strictly_feasible(real* x, const real* lb, const real* ub, real rstep) {
#pragma unroll
  for (int i = 0; i < N; i++) {
    real xn = x[i];
    if (rstep == 0) {
      if (x[i] <= lb[i]) xn = nextafter(lb[i], ub[i]);
      if (x[i] >= ub[i]) xn = nextafter(ub[i], lb[i]);
    } else {
      real ld = x[i] - lb[i], ud = ub[i] - x[i];
      real lt = rstep * fmax(1.0, fabs(lb[i])), ut = rstep * fmax(1.0, fabs(ub[i]));
      if (ld <= fmin(ud, lt)) xn = lb[i] + lt;
      if (ud <= fmin(ld, ut)) xn = ub[i] - ut;
    }
    if (xn < lb[i] || xn > ub[i]) xn = 0.5 * (lb[i] + ub[i]);
    x[i] = xn;
  }
}
// returns min step; hits as a bitmask of components attaining it (sign irrelevant for reflection)
template <int N>
__device__ __forceinline__ real step_to_bound(const real* x, const real* s, const real* lb, const real* ub, uint32_t* hits) {
  real st[N], mn = INFINITY;
#pragma unroll
  for (int i = 0; i < N; i++) {
    st[i] = (s[i] != 0) ? fmax((lb[i] - x[i]) / s[i], (ub[i] - x[i]) / s[i]) : INFINITY;
    mn = fmin(mn, st[i]);
  }
  if (hits) {
    uint32_t h = 0;
#pragma unroll
    for (int i = 0; i < N; i++) if (st[i] == mn && s[i] != 0) h |= 1u << i;
    *hits = h;
  }
  return mn;
}
__device__ __forceinline__ void min_quad_1d(real a, real b, real lo, real hi, real c, real& t_out, real& y_out) {
  real tb = lo, yb = lo * (a * lo + b) + c;
  real y1 = hi * (a * hi + b) + c;
  if (y1 < yb) { yb = y1; tb = hi; }
  if (a != 0) {
    real ex = -0.5 * b / a;
    if (lo < ex && ex < hi) { real y2 = ex * (a * ex + b) + c; if (y2 < yb) { yb = y2; tb = ex; } }
  }
  t_out = tb; y_out = yb;
}

// scipy trf.py select_step; quadratic model q(s) = 0.5 s^T A s + g_h^T s
template <int N>
__device__ __forceinline__ real select_step(const real* x, LMat<N, N> A, const real* g_h, real* p, real* p_h,
                                            const real* d, real Delta, const real* lb, const real* ub, real theta,
                                            real* step, real* step_h) {
  real xp[N];
#pragma unroll
  for (int i = 0; i < N; i++) xp[i] = x[i] + p[i];
  if (in_bounds<N>(xp, lb, ub)) {
    real pv = 0.5 * quad_form<N>(A, p_h, p_h) + vdot<N>(g_h, p_h);
#pragma unroll
    for (int i = 0; i < N; i++) { step[i] = p[i]; step_h[i] = p_h[i]; }
    return -pv;
  }
  uint32_t hits;
  real p_stride = step_to_bound<N>(x, p, lb, ub, &hits);
  real r_h[N], r[N], xb[N];
#pragma unroll
  for (int i = 0; i < N; i++) { r_h[i] = ((hits >> i) & 1u) ? -p_h[i] : p_h[i]; r[i] = d[i] * r_h[i]; }
#pragma unroll
  for (int i = 0; i < N; i++) { p[i] *= p_stride; p_h[i] *= p_stride; xb[i] = x[i] + p[i]; }
  real to_tr;
  {
    real a = vdot<N>(r_h, r_h), b = vdot<N>(p_h, r_h), c = vdot<N>(p_h, p_h) - Delta * Delta;
    real dd = sqrt(b * b - a * c);
    real q = -(b + copysign(dd, b));
    real t1 = q / a, t2 = c / q;
    to_tr = t1 < t2 ? t2 : t1;
  }
  real to_bound = step_to_bound<N>(xb, r, lb, ub, nullptr);
  real r_stride = fmin(to_bound, to_tr), rl, ru;
  if (r_stride > 0) { rl = (1 - theta) * p_stride / r_stride; ru = (r_stride == to_bound) ? theta * to_bound : to_tr; }
  else { rl = 0; ru = -1; }
  real r_value;
  if (rl <= ru) {
    real a = 0.5 * quad_form<N>(A, r_h, r_h);
    real b = vdot<N>(g_h, r_h) + quad_form<N>(A, p_h, r_h);
    real c = 0.5 * quad_form<N>(A, p_h, p_h) + vdot<N>(g_h, p_h);
    min_quad_1d(a, b, rl, ru, c, r_stride, r_value);
#pragma unroll
    for (int i = 0; i < N; i++) { r_h[i] = r_h[i] * r_stride + p_h[i]; r[i] = r_h[i] * d[i]; }
  } else r_value = INFINITY;
#pragma unroll
  for (int i = 0; i < N; i++) { p[i] *= theta; p_h[i] *= theta; }
  real p_value = 0.5 * quad_form<N>(A, p_h, p_h) + vdot<N>(g_h, p_h);
  real ag_h[N], ag[N];
#pragma unroll
  for (int i = 0; i < N; i++) { ag_h[i] = -g_h[i]; ag[i] = d[i] * ag_h[i]; }
  real to_tr2 = Delta / vnorm<N>(ag_h);
  real to_bound2 = step_to_bound<N>(x, ag, lb, ub, nullptr);
  real ag_stride = (to_bound2 < to_tr2) ? theta * to_bound2 : to_tr2;
  real ag_value;
  {
    real a = 0.5 * quad_form<N>(A, ag_h, ag_h), b = vdot<N>(g_h, ag_h);
    min_quad_1d(a, b, 0, ag_stride, 0, ag_stride, ag_value);
  }
#pragma unroll
  for (int i = 0; i < N; i++) { ag_h[i] *= ag_stride; ag[i] *= ag_stride; }
  if (p_value < r_value && p_value < ag_value) {
#pragma unroll
    for (int i = 0; i < N; i++) { step[i] = p[i]; step_h[i] = p_h[i]; }
    return -p_value;
  } else if (r_value < p_value && r_value < ag_value) {
#pragma unroll
    for (int i = 0; i < N; i++) { step[i] = r[i]; step_h[i] = r_h[i]; }
    return -r_value;
  }
#pragma unroll
  for (int i = 0; i < N; i++) { step[i] = ag[i]; step_h[i] = ag_h[i]; }
  return -ag_value;
}

// scipy trf.py trf_bounds (tr_solver='exact', x_scale=1, loss='linear', ftol=xtol=gtol=1e-8, max_nfev=100n).
// x: in = strictly feasible start, out = result.x; x_last = last point the residual/Jacobian was evaluated at.
template <int N>
__device__ int trf_bounds(const IkCtx<N>& P, const IkLds<N>& S, real* x, real* x_last, int* nfev_out, ProfIk& pf) {
  const real ftol = 1e-8, xtol = 1e-8, gtol = 1e-8;
  const int max_nfev = 100 * N;
  const real jreg2 = 2 * P.m->ik_jac_reg * P.m->ik_jac_reg;
  IkEval<N> E, En;
  E.Jt = S.J;
  LMat<6, N> Jtrial = S.J2;
  const LMat<N, N> A = S.A;
  real g[N], v[N], dv[N], d[N], diag_h[N], g_h[N];
  real x_new[N], step[N], step_h[N], p[N], p_h[N], ft_new[6];
  ik_eval<N, true>(P, x, E);
  int nfev = 1;
  real cost = ik_cost<N>(P, x, E.ft);
  ik_grad<N>(P, x, E, g);
#pragma unroll
  for (int i = 0; i < N; i++) {
    v[i] = 1;
    if (g[i] < 0) v[i] = P.ub[i] - x[i];
    if (g[i] > 0) v[i] = x[i] - P.lb[i];
  }
  real Delta;
  { real s = 0;
#pragma unroll
    for (int i = 0; i < N; i++) s += x[i] * x[i] / v[i];
    Delta = sqrt(s); if (Delta == 0) Delta = 1.0; }
  real alpha = 0.0, cost_new = cost;
  int status = -1;
  pf.ph(0);
#pragma unroll
  for (int i = 0; i < N; i++) x_last[i] = x[i];
  for (;;) {
    real g_norm = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
      v[i] = 1; dv[i] = 0;
      if (g[i] < 0) { v[i] = P.ub[i] - x[i]; dv[i] = -1; }
      if (g[i] > 0) { v[i] = x[i] - P.lb[i]; dv[i] = 1; }
      g_norm = fmax(g_norm, fabs(g[i] * v[i]));
    }
    if (g_norm < gtol) status = 1;
    if (status != -1 || nfev == max_nfev) break;
#pragma unroll
    for (int i = 0; i < N; i++) { d[i] = sqrt(v[i]); diag_h[i] = g[i] * dv[i]; g_h[i] = d[i] * g[i]; }
#pragma unroll
    for (int i = 0; i < N; i++)
#pragma unroll
      for (int j = 0; j <= i; j++) {
        real s = 0;
#pragma unroll
        for (int r = 0; r < 6; r++) s += E.Jt(r, i) * E.Jt(r, j);
        if (i == j) s += jreg2;
        s *= d[i] * d[j];
        if (i == j) s += diag_h[i];
        A(i, j) = s; A(j, i) = s;
      }
    real theta = fmax(0.995, 1 - g_norm);
    real actual = -1;
    pf.ph(1);
    while (actual <= 0 && nfev < max_nfev) {
      solve_tr<N>(A, S.L, g_h, Delta, alpha, p_h);
      pf.ph(2);
#pragma unroll
      for (int i = 0; i < N; i++) p[i] = d[i] * p_h[i];
      real predicted = select_step<N>(x, A, g_h, p, p_h, d, Delta, P.lb, P.ub, theta, step, step_h);
      pf.ph(3);
#pragma unroll
      for (int i = 0; i < N; i++) x_new[i] = x[i] + step[i];
      make_strictly_feasible<N>(x_new, P.lb, P.ub, 0.0);
      // ik_res(x_new) -- and, in the same kinematics pass, what ik_jac(x_new) would recompute if the step is
      // accepted (the reference evaluates both at the same point, ik_mujoco.py:20-97)
      En.Jt = Jtrial;
      ik_eval<N, true>(P, x_new, En);
      pf.ph(4);
#pragma unroll
      for (int r = 0; r < 6; r++) ft_new[r] = En.ft[r];
#pragma unroll
      for (int i = 0; i < N; i++) x_last[i] = x_new[i];
      nfev++;
      real shn = vnorm<N>(step_h);
      bool fin = true;
#pragma unroll
      for (int r = 0; r < 6; r++) fin = fin && isfinite(ft_new[r]);
      if (!fin) { Delta = 0.25 * shn; continue; }
      cost_new = ik_cost<N>(P, x_new, ft_new);
      actual = cost - cost_new;
      real ratio, Delta_new = Delta;
      if (predicted > 0) ratio = actual / predicted;
      else if (predicted == 0 && actual == 0) ratio = 1;
      else ratio = 0;
      if (ratio < 0.25) Delta_new = 0.25 * shn;
      else if (ratio > 0.75 && shn > 0.95 * Delta) Delta_new = Delta * 2.0;
      real sn = vnorm<N>(step), xn = vnorm<N>(x);
      bool ft_ok = (actual < ftol * cost) && (ratio > 0.25);
      bool xt_ok = sn < xtol * (xtol + xn);
      if (ft_ok && xt_ok) status = 4; else if (ft_ok) status = 2; else if (xt_ok) status = 3;
      if (status != -1) break;
      alpha *= Delta / Delta_new;
      Delta = Delta_new;
    }
    if (actual > 0) {
#pragma unroll
      for (int i = 0; i < N; i++) x[i] = x_new[i];
      cost = cost_new;
      const LMat<6, N> tmpJ = E.Jt;        // accept: the trial Jacobian becomes current, the old slot becomes trial
      E.Jt = En.Jt; Jtrial = tmpJ;
#pragma unroll
      for (int r = 0; r < 6; r++) E.ft[r] = ft_new[r];
      ik_grad<N>(P, x, E, g);
    }
    pf.ph(5);
  }
  if (status == -1) status = 0;
  *nfev_out = nfev;
  return status;
}

// ik_mujoco.py:100-155 for one (env, arm); x0 = current arm joints
template <int N>
__device__ __forceinline__ void ik_solve(IkCtx<N>& P, const IkLds<N>& S, const real* x0, real* q_out, real* x_last, int* nfev, int* status, ProfIk& pf) {
  real x[N];
#pragma unroll
  for (int i = 0; i < N; i++) { x[i] = x0[i]; x_last[i] = x0[i]; }
  *nfev = 0; *status = -2;
  if (in_bounds<N>(x, P.lb, P.ub)) {            // else least_squares raises ValueError -> "IK failed"
    make_strictly_feasible<N>(x, P.lb, P.ub, 1e-10);
    *status = trf_bounds<N>(P, S, x, x_last, nfev, pf);
  }
#pragma unroll
  for (int i = 0; i < N; i++) q_out[i] = fmin(fmax(x[i], P.lb[i]), P.ub[i]);   // :147-152 (:140-145 is a no-op)
}

template <int N>
__device__ __forceinline__ void ik_ctx_init(IkCtx<N>& P, const KDeviceModel* dm, int arm) {
  P.m = &dm->d; P.ax = &dm->x; P.arm = arm;
#pragma unroll
  for (int i = 0; i < N; i++) {
    int q = dm->d.arm_q_id[arm][i];
    P.q_home[i] = dm->d.q_home[q];
    P.lb[i] = dm->d.jnt_range[q][0]; P.ub[i] = dm->d.jnt_range[q][1];
  }
}

__device__ __forceinline__ real f32r(real x) { return (real)(float)x; }

// ---------------------------------------------------------------------------------------------
// before_step for every env: decode + IK + ctrl.  One lane per (env, arm slot); slot 0 = right arm.
// Writes ctrl (float32-rounded, env_sim.py:40) and qpos_ik (arm joints left at the IK's last
// evaluated point).  qpos itself is NOT modified: the dynamics kernel still needs the pre-IK
// configuration for the stale mj_step2 (dm_control legacy step order).
template <int N>
__global__ __launch_bounds__(64) void k_before_step(const KDeviceModel* __restrict__ dm, KDeviceState st,
                                                    const float* __restrict__ act) {
  const KModelDesc* m = &dm->d;
  const int NE = st.num_envs;
  constexpr int NSLOT = 64 / IK_LANE_STRIDE;
  __shared__ real ik_lds[IkLds<N>::WORDS * NSLOT];
  const IkLds<N> S(ik_lds, threadIdx.x / IK_LANE_STRIDE, NSLOT);
  // The TRF is a long data-dependent serial program: a wave runs for as long as its slowest lane.  With
  // only num_envs * narm problems (4096..16384) against 65,536 lanes, problems are spread IK_LANE_STRIDE
  // lanes apart so that a wave carries 64 / stride of them (shorter divergence tail, all SIMDs busy).
  const int gtid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gtid % IK_LANE_STRIDE) return;
  const int tid = gtid / IK_LANE_STRIDE;
  const int env = tid % NE, arm = tid / NE;
  if (arm >= KM_MAX_ARMS || !m->arm_present[arm]) return;
  const int nl = m->nlink;
  const float* a = act + (size_t)env * m->act_dim;
  static const int grip_key[2] = {KM_ACT_GRIP_R, KM_ACT_GRIP_L};
  static const int pos_key[2] = {KM_ACT_EER_POS, KM_ACT_EEL_POS};
  static const int orn_key[2] = {KM_ACT_EER_ORN, KM_ACT_EEL_ORN};
  static const int qp_key[2] = {KM_ACT_QPOS_R, KM_ACT_QPOS_L};
  // ---- grip (env_sim.py:41-59): float32 arithmetic exactly as numpy does it
  int cg = m->act_col[grip_key[arm]];
  if (cg >= 0) {
    int g0 = m->arm_grip_id[arm][0], g1 = m->arm_grip_id[arm][1];
    float g = a[cg] * (float)m->ee_s_delta;
    g = (float)((double)g + st.qpos[(size_t)g0 * NE + env]);
    g = fminf(fmaxf(g, (float)m->ee_s_min), (float)m->ee_s_max);
    st.ctrl[(size_t)g0 * NE + env] = (double)g;
    st.ctrl[(size_t)g1 * NE + env] = (double)g;
  }
  real x0[N];
#pragma unroll
  for (int i = 0; i < N; i++) x0[i] = st.qpos[(size_t)m->arm_q_id[arm][i] * NE + env];
  int cp = m->act_col[pos_key[arm]], co = m->act_col[orn_key[arm]], cq = m->act_col[qp_key[arm]];
  if (cp >= 0) {
    IkCtx<N> P;
    ik_ctx_init<N>(P, dm, arm);
#pragma unroll
    for (int k = 0; k < KM_MAX_CHAIN; k++) {
      int l = dm->x.chain_link[arm][k];
      P.qfix[k] = (k < dm->x.chain_len[arm] && k >= N) ? st.qpos[(size_t)l * NE + env] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < N; i++) P.q_prev[i] = x0[i];
    // current site pose = goal-free evaluation at x0: reuse ik_eval with goal = 0 / identity
    P.goal_pos[0] = 0; P.goal_pos[1] = 0; P.goal_pos[2] = 0;
    P.goal_quat[0] = 1; P.goal_quat[1] = 0; P.goal_quat[2] = 0; P.goal_quat[3] = 0;
    IkEval<N> E0;
    E0.Jt = S.J;
    ik_eval<N, false>(P, x0, E0);
    const real* smat = E0.smat;
    // EE-delta decode (env_sim.py:60-69): euler("xyz", extrinsic) of the site matrix + delta -> quaternion
    real e0 = atan2(smat[7], smat[8]);
    real e1 = atan2(-smat[6], sqrt(smat[7] * smat[7] + smat[8] * smat[8]));
    real e2 = atan2(smat[3], smat[0]);
    e0 += (double)a[co] * m->ee_orn_delta[0];
    e1 += (double)a[co + 1] * m->ee_orn_delta[1];
    e2 += (double)a[co + 2] * m->ee_orn_delta[2];
    real qx[4] = {cos(e0 * 0.5), sin(e0 * 0.5), 0, 0}, qy[4] = {cos(e1 * 0.5), 0, sin(e1 * 0.5), 0};
    real qz[4] = {cos(e2 * 0.5), 0, 0, sin(e2 * 0.5)}, t4[4];
    qmul(t4, qy, qx);
    qmul(P.goal_quat, qz, t4);
    P.goal_pos[0] = (double)a[cp] * m->ee_pos_delta[0] + E0.sp[0];
    P.goal_pos[1] = (double)a[cp + 1] * m->ee_pos_delta[1] + E0.sp[1];
    P.goal_pos[2] = (double)a[cp + 2] * m->ee_pos_delta[2] + E0.sp[2];
    real qo[N], xl[N];
    int nfev, status;
    ProfIk pf;
    pf.start();
    ik_solve<N>(P, S, x0, qo, xl, &nfev, &status, pf);
    pf.ph(6);
    pf.flush();
#pragma unroll
    for (int i = 0; i < N; i++) {
      int q = m->arm_q_id[arm][i];
      st.ctrl[(size_t)q * NE + env] = f32r(qo[i]);
      st.qpos_ik[(size_t)q * NE + env] = xl[i];
    }
    st.ik_nfev[(size_t)arm * NE + env] = nfev;
    st.ik_status[(size_t)arm * NE + env] = status;
  } else {
    st.ik_nfev[(size_t)arm * NE + env] = 0;
    st.ik_status[(size_t)arm * NE + env] = -3;
    if (cq >= 0) {   // joint-delta modes, env_sim.py:100-103
#pragma unroll
      for (int i = 0; i < N; i++) {
        int q = m->arm_q_id[arm][i];
        st.ctrl[(size_t)q * NE + env] = f32r(x0[i] + (double)(a[cq + i] * (float)m->q_pos_delta));
      }
    }
  }
}

// ctrl = data.ctrl.astype(float32) for entries no action key touches, and qpos_ik = qpos for the
// robot joints (the IK kernel then overwrites the arm entries).
__global__ void k_prepare(const KDeviceModel* __restrict__ dm, KDeviceState st) {
  const int NE = st.num_envs, nl = dm->d.nlink;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= NE * nl) return;
  st.ctrl[i] = f32r(st.ctrl[i]);
  st.qpos_ik[i] = st.qpos[i];
}

void kmanip_launch_ik(const KDeviceModel* dm, const KModelDesc& hd, const KDeviceState& st, const float* act,
                      hipStream_t stream) {
  int n0 = st.num_envs * hd.nlink;
  hipLaunchKernelGGL(k_prepare, dim3((n0 + 255) / 256), dim3(256), 0, stream, dm, st);
  int narm_slots = (hd.arm_present[1]) ? 2 : 1;
  int nt = st.num_envs * narm_slots * IK_LANE_STRIDE;
  int nik = hd.arm_nq[0] ? hd.arm_nq[0] : hd.arm_nq[1];
  if (nik == 7) hipLaunchKernelGGL(k_before_step<7>, dim3((nt + 63) / 64), dim3(64), 0, stream, dm, st, act);
  else hipLaunchKernelGGL(k_before_step<6>, dim3((nt + 63) / 64), dim3(64), 0, stream, dm, st, act);
}

// ---------------------------------------------------------------------------------------------
// standalone batched ik() for parity tests: qpos env-major [n][nq] (mutated like the reference)
template <int N>
__global__ __launch_bounds__(64) void k_ik_standalone(const KDeviceModel* __restrict__ dm, int arm, int n, int nq,
                                                      double* qpos, const double* goal_pos, const double* goal_quat,
                                                      double* q_out, int32_t* nfev_o, int32_t* status_o) {
  __shared__ real ik_lds[IkLds<N>::WORDS * 64];
  const IkLds<N> S(ik_lds, threadIdx.x, 64);
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  IkCtx<N> P;
  ik_ctx_init<N>(P, dm, arm);
  double* qp = qpos + (size_t)e * nq;
  real x0[N];
#pragma unroll
  for (int i = 0; i < N; i++) { x0[i] = qp[dm->d.arm_q_id[arm][i]]; P.q_prev[i] = x0[i]; }
#pragma unroll
  for (int k = 0; k < KM_MAX_CHAIN; k++) {
    int l = dm->x.chain_link[arm][k];
    P.qfix[k] = (k < dm->x.chain_len[arm] && k >= N) ? qp[l] : 0.0;
  }
  for (int c = 0; c < 3; c++) P.goal_pos[c] = goal_pos[3 * e + c];
  for (int c = 0; c < 4; c++) P.goal_quat[c] = goal_quat[4 * e + c];
  real qo[N], xl[N];
  int nfev, status;
  ProfIk pf;
  pf.start();
  ik_solve<N>(P, S, x0, qo, xl, &nfev, &status, pf);
#pragma unroll
  for (int i = 0; i < N; i++) { q_out[(size_t)e * N + i] = qo[i]; qp[dm->d.arm_q_id[arm][i]] = xl[i]; }
  nfev_o[e] = nfev; status_o[e] = status;
}

void kmanip_launch_ik_standalone(const KDeviceModel* dm, const KModelDesc& hd, int arm, int n, double* qpos,
                                 const double* goal_pos, const double* goal_quat, double* q_out, int32_t* nfev,
                                 int32_t* status, hipStream_t stream) {
  int nq = hd.nlink + 7;
  if (hd.arm_nq[arm] == 7)
    hipLaunchKernelGGL(k_ik_standalone<7>, dim3((n + 63) / 64), dim3(64), 0, stream, dm, arm, n, nq, qpos, goal_pos, goal_quat, q_out, nfev, status);
  else
    hipLaunchKernelGGL(k_ik_standalone<6>, dim3((n + 63) / 64), dim3(64), 0, stream, dm, arm, n, nq, qpos, goal_pos, goal_quat, q_out, nfev, status);
}
