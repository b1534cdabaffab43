// kmanip_ik_coop.hip -- stand-alone launches of the cooperative decode + IK device code (kmanip_ik_coop.hpp):
// the batched ik() entry point used by parity tests and the unfused before_step kernel kept for A/B runs.
#include "kmanip_ik_coop.hpp"
#include <stdlib.h>

// ---------------------------------------------------------------------------------------------
// before_step for every env, stand-alone launch (A/B path: KMANIP_IK_UNFUSED=1; the product path runs the same
// device code inside k_step): 8 lanes per (env, arm) problem
struct GlobalIO {
  KDeviceState st; int env;
  __device__ __forceinline__ real qpos(int i) const { return st.qpos[(size_t)i * st.num_envs + env]; }
  __device__ __forceinline__ void set_ctrl(int i, real v) { st.ctrl[(size_t)i * st.num_envs + env] = v; }
  __device__ __forceinline__ void set_qpos_ik(int i, real v) { st.qpos_ik[(size_t)i * st.num_envs + env] = v; }
  __device__ __forceinline__ void set_diag(int arm, int nfev, int status) {
    st.ik_nfev[(size_t)arm * st.num_envs + env] = nfev; st.ik_status[(size_t)arm * st.num_envs + env] = status;
  }
};
// PPB = problems per single-wave workgroup (<= PPW): the kernel needs > 256 registers, i.e. one wave per SIMD, so a
// batch with fewer than 1024 full waves is spread over more, emptier waves (less lock-step divergence too).
template <int N, int PPB>
__global__ __launch_bounds__(64) void k_before_step_coop(const KDeviceModel* __restrict__ dm, KDeviceState st,
                                                         const float* __restrict__ act, int nprob) {
  const KModelDesc* m = &dm->d;
  const int NE = st.num_envs;
  const int slot = threadIdx.x / GS, c = threadIdx.x % GS;
  const int prob = blockIdx.x * PPB + slot;
  if (c >= GI || slot >= PPB || prob >= nprob) return;      // whole problem exits together (lanes 8..15 of a row stay idle)
  const int env = prob % NE, arm = prob / NE;
  if (!m->arm_present[arm]) return;
  GlobalIO io{st, env};
  Prof pf;
  coop_before_step<N>(dm, arm, c, act + (size_t)env * m->act_dim, io, &pf);
}

__global__ void k_prepare_coop(const KDeviceModel* __restrict__ dm, KDeviceState st) {
  const int NE = st.num_envs, nl = dm->d.nlink;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= NE * nl) return;
  st.ctrl[i] = f32r_c(st.ctrl[i]);
  st.qpos_ik[i] = st.qpos[i];
}

void kmanip_launch_ik_coop(const KDeviceModel* dm, const KModelDesc& hd, const KDeviceState& st, const float* act,
                           hipStream_t stream) {
  int n0 = st.num_envs * hd.nlink;
  hipLaunchKernelGGL(k_prepare_coop, dim3((n0 + 255) / 256), dim3(256), 0, stream, dm, st);
  int narm_slots = (hd.arm_present[1]) ? 2 : 1;
  int nprob = st.num_envs * narm_slots;
  int nik = hd.arm_nq[0] ? hd.arm_nq[0] : hd.arm_nq[1];
  static int forced = -1;
  if (forced < 0) { const char* e = getenv("KMANIP_IK_PPB"); forced = e ? atoi(e) : 0; }
  int ppb = PPW;
  if (forced > 0) ppb = forced;
  else while (ppb > 2 && (nprob + ppb - 1) / ppb < 1024) ppb >>= 1;   // 1024 = SIMDs on the chip
#define KM_IK_LAUNCH(NN, PP) hipLaunchKernelGGL((k_before_step_coop<NN, PP>), dim3((nprob + PP - 1) / PP), dim3(64), 0, stream, dm, st, act, nprob)
  if (nik == 7) { if (ppb >= 4) KM_IK_LAUNCH(7, 4); else KM_IK_LAUNCH(7, 2); }
  else { if (ppb >= 4) KM_IK_LAUNCH(6, 4); else KM_IK_LAUNCH(6, 2); }
#undef KM_IK_LAUNCH
}

// standalone batched ik() for parity tests: qpos env-major [n][nq] (mutated like the reference)
template <int N>
__global__ __launch_bounds__(64) void k_ik_coop_standalone(const KDeviceModel* __restrict__ dm, int arm, int n, int nq,
                                                           double* qpos, const double* goal_pos, const double* goal_quat,
                                                           double* q_out, int32_t* nfev_o, int32_t* status_o) {
  const KModelDesc* m = &dm->d;
  const int slot = threadIdx.x / GS, c = threadIdx.x % GS;
  const int e = blockIdx.x * PPW + slot;
  if (c >= GI || e >= n) return;
  double* qp = qpos + (size_t)e * nq;
  CoopCtx<N> P;
  Prof pf;
  P.m = m; P.ax = &dm->x; P.arm = arm; P.c = c; P.on = c < N; P.pf = &pf;
  coop_chain_setup<N>(P);
  const int q = m->arm_q_id[arm][P.on ? c : 0];
  const real x0 = P.on ? qp[q] : 0.0;
  P.q_prev = x0; P.q_home = m->q_home[q]; P.lb = m->jnt_range[q][0]; P.ub = m->jnt_range[q][1];
  P.qfix = (P.clen > N) ? qp[dm->x.chain_link[arm][P.clen - 1]] : 0.0;
  for (int k = 0; k < 3; k++) P.goal_pos[k] = goal_pos[3 * e + k];
  for (int k = 0; k < 4; k++) P.goal_quat[k] = goal_quat[4 * e + k];
  real xl;
  int nfev, status;
  const real qo = coop_ik_solve<N>(P, x0, xl, &nfev, &status);
  KM_GSYNC();
  if (P.on) { q_out[(size_t)e * N + c] = qo; qp[q] = xl; }
  if (c == 0) { nfev_o[e] = nfev; status_o[e] = status; }
}

void kmanip_launch_ik_coop_standalone(const KDeviceModel* dm, const KModelDesc& hd, int arm, int n, double* qpos,
                                      const double* goal_pos, const double* goal_quat, double* q_out, int32_t* nfev,
                                      int32_t* status, hipStream_t stream) {
  int nq = hd.nlink + 7;
  dim3 grid((n + PPW - 1) / PPW);
  if (hd.arm_nq[arm] == 7)
    hipLaunchKernelGGL(k_ik_coop_standalone<7>, grid, dim3(64), 0, stream, dm, arm, n, nq, qpos, goal_pos, goal_quat, q_out, nfev, status);
  else
    hipLaunchKernelGGL(k_ik_coop_standalone<6>, grid, dim3(64), 0, stream, dm, arm, n, nq, qpos, goal_pos, goal_quat, q_out, nfev, status);
}

// ik_res / ik_jac (ik_mujoco.py:20-97) of the cooperative IK's own evaluation code at x = qpos[q_mask], q_pos_prev = x:
// res [n][6+2N], jac [n][(6+2N) x N] row-major -- for parity tests against the NumPy/SciPy fixtures (res0 / jac0).
template <int N>
__global__ __launch_bounds__(64) void k_ik_eval_coop(const KDeviceModel* __restrict__ dm, int arm, int n, int nq, const double* qpos,
                                                     const double* goal_pos, const double* goal_quat, double* res, double* jac) {
  const KModelDesc* m = &dm->d;
  const int slot = threadIdx.x / GS, c = threadIdx.x % GS;
  const int e = blockIdx.x * PPW + slot;
  if (c >= GI || e >= n) return;
  const double* qp = qpos + (size_t)e * nq;
  CoopCtx<N> P;
  Prof pf;
  P.m = m; P.ax = &dm->x; P.arm = arm; P.c = c; P.on = c < N; P.pf = &pf;
  coop_chain_setup<N>(P);
  const int q = m->arm_q_id[arm][P.on ? c : 0];
  const real x0 = P.on ? qp[q] : 0.0;
  P.q_prev = x0; P.q_home = m->q_home[q]; P.lb = m->jnt_range[q][0]; P.ub = m->jnt_range[q][1];
  P.qfix = (P.clen > N) ? qp[dm->x.chain_link[arm][P.clen - 1]] : 0.0;
  for (int k = 0; k < 3; k++) P.goal_pos[k] = goal_pos[3 * e + k];
  for (int k = 0; k < 4; k++) P.goal_quat[k] = goal_quat[4 * e + k];
  real ft[6], Jc[6];
  coop_eval<N, true>(P, x0, ft, Jc, nullptr, nullptr);
  constexpr int M = 6 + 2 * N;
  double* r = res + (size_t)e * M;
  double* J = jac + (size_t)e * M * N;
  if (c == 0) for (int k = 0; k < 6; k++) r[k] = ft[k];
  if (P.on) {
    r[6 + c] = m->ik_res_reg_prev * (x0 - P.q_prev);
    r[6 + N + c] = m->ik_res_reg_home * (x0 - P.q_home);
    for (int k = 0; k < M; k++) J[k * N + c] = 0;
    for (int k = 0; k < 6; k++) J[k * N + c] = Jc[k];
    J[(6 + c) * N + c] = m->ik_jac_reg;          // both regulariser blocks use IK_JAC_REG (ik_mujoco.py:92,97)
    J[(6 + N + c) * N + c] = m->ik_jac_reg;
  }
}

void kmanip_launch_ik_eval_coop(const KDeviceModel* dm, const KModelDesc& hd, int arm, int n, const double* qpos,
                                const double* goal_pos, const double* goal_quat, double* res, double* jac, hipStream_t stream) {
  int nq = hd.nlink + 7;
  dim3 grid((n + PPW - 1) / PPW);
  if (hd.arm_nq[arm] == 7) hipLaunchKernelGGL(k_ik_eval_coop<7>, grid, dim3(64), 0, stream, dm, arm, n, nq, qpos, goal_pos, goal_quat, res, jac);
  else hipLaunchKernelGGL(k_ik_eval_coop<6>, grid, dim3(64), 0, stream, dm, arm, n, nq, qpos, goal_pos, goal_quat, res, jac);
}
