// kmanip_dyn.hip -- the physics half of one control step, all envs, one launch (gfx950, wave64).
//
// Replaces, for every env, `physics.step(n_sub_steps)` as dm_control runs it for KManipEnvSim.k_step
// (reference gym_kmanip/env_sim.py:196-200, control_timestep :210): legacy order
//   mj_step2 (on products of the PRE-IK state) ; (n-1) x mj_step ; mj_step1
// followed by KManipTask.get_reward (env_sim.py:148-179) and get_observation (env_sim.py:110-146), the
// TimeLimit done flag (__init__.py:28,247) and, when enabled, the auto-reset
// (KManipTask.initialize_episode env_sim.py:23-36 + dm_control's mj_forward without actuation).
//
// Execution model ("many envs per wavefront"): a workgroup is ONE wave of 64 lanes holding 64/G envs;
// each env is owned by a group of G lanes (G = 16 for nv = 16, G = 32 for nv = 26), lane i of the group
// owning dof i (its acceleration / velocity component lives in that lane's registers).  Every per-env
// intermediate (body frames, joint-space inertia and its inverse, contact Jacobian bases, constraint
// rows) is staged in LDS; HBM is touched once on entry and once on exit with struct-of-arrays
// coalesced columns.  Small per-env reductions (constraint-row dot products, norms) use DPP/bpermute
// wave shuffles of width G.  Groups never need s_barrier: all lanes of a group sit in the same wave.
//
// Formulations deliberately differ from the oracle's (so parity is a cross-check, not a re-run):
//   mass matrix      : sum over bodies of COM-Jacobian outer products (oracle: composite-body CRBA)
//   bias forces      : per-body bias wrenches projected with J^T   (oracle: RNE backward recursion)
//   M^-1             : explicit inverse via cooperative Cholesky    (oracle: factor + solves)
//   constraint rows  : compact (single-dof rows + 4-vector contact bases, edges expanded on the fly)
//   PGS              : per-contact block form with a 4x4 Gram matrix (algebraically the same row order)
#include "kmanip_device.hpp"

template <int NL> struct Dim {
  static constexpr int NV = NL + 6;
  static constexpr int NQ = NL + 7;
  static constexpr int NS = NV + NL;              // single-dof rows: friction loss (<= nv) + limits (<= nl)
  static constexpr int NC = 4 + 2 * (NL / 5);     // cube-table corners + 2 pairs per finger sphere
};

// Solver view of one pyramidal contact (all group-uniform scalars).  Basis index 0 = normal, 1..2 =
// tangents, 3 = torsion.  Edge e = 2*(k-1) + s uses J_0 + sm J_k with sm = (s ? -mu[k-1] : mu[k-1]).
struct ConRec {
  real G[10];     // symmetric Gram J_k M^-1 J_l^T, packed (0,0)(0,1)(0,2)(0,3)(1,1)(1,2)(1,3)(2,2)(2,3)(3,3)
  real mu[3];
  real R;         // regulariser shared by all edges (MuJoCo pyramidal rule)
  real A[4];      // reference-acceleration basis: aref_e = A[0] + sm * A[k]
  real inv[6];    // 1 / (A_ee + R)
  real f[6];      // edge forces
};
__device__ __forceinline__ constexpr int gidx(int k, int l) {
  const int a = k < l ? k : l, b = k < l ? l : k;
  return a * 4 - a * (a - 1) / 2 + (b - a);
}

template <int NL>
struct Ws {
  static constexpr int NV = Dim<NL>::NV, NQ = Dim<NL>::NQ, NS = Dim<NL>::NS, NC = Dim<NL>::NC;
  real qpos[NQ], qvel[NV], ctrl[NL], warm[NV], qpos_ik[NL];
  // kinematics
  real xpos[NL][3], xquat[NL][4], xmat[NL][9], axis[NL][3], cpos[NL][3];
  real cube_mat[9];
  // smooth dynamics
  real Minv[NL][NL];      // joint-space inertia, overwritten by its inverse
  real bias[NV], as[NV], tmp[NV];
  real Mtrace;
  // single-dof constraint rows (friction loss, joint limits)
  int ns;
  int s_dof[NS], s_type[NS];
  real s_sign[NS], s_pos[NS], s_f[NS], s_R[NS], s_aref[NS], s_den[NS], s_inv[NS], s_floss[NS];
  // contacts
  int ncon;
  int c_b1[NC], c_b2[NC], c_dim[NC], c_cube[NC];
  real c_pos[NC][3], c_frame[NC][9], c_dist[NC];
  real c_Jb[NC][4][NV];
  union {
    real c_Bb[NC][4][NV];                          // M^-1 J^T of the contact bases (built after M^-1)
    struct { real Lw[NL][NL]; real FN[NL][6]; };   // Cholesky workspace / per-body bias wrenches (dead by then)
  };
  ConRec c_rec[NC];        // per-contact solver scalars, read as one block per contact per sweep
  uint32_t contact_mask;
  int touch_fc, touch_ct, bad;
};

#define GSYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

// cross-lane double move with a DPP control word (row = 16 lanes)
template <int CTRL> __device__ __forceinline__ real dpp_f64(real v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
// sum over the G lanes of a group, result identical (bitwise) in every lane.  16-lane rows use four DPP
// steps (row_mirror, row_half_mirror, two quad_perms) instead of ds_bpermute; G = 32 adds one swizzle.
template <int G> __device__ __forceinline__ real gsum(real v) {
  static_assert(G == 16 || G == 32, "lane group must be one or two DPP rows");
  v += dpp_f64<0x140>(v);   // row_mirror:      i <-> 15 - i
  v += dpp_f64<0x141>(v);   // row_half_mirror: i <-> 7 - i within each half row
  v += dpp_f64<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_f64<0x4E>(v);    // quad_perm [2,3,0,1]
  if (G == 32) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_ds_swizzle(lo, 0x401F);   // bit mode: xor lane id with 16
    hi = __builtin_amdgcn_ds_swizzle(hi, 0x401F);
    v += __hiloint2double(hi, lo);
  }
  return v;
}
template <int G> __device__ __forceinline__ int gor(int v) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) v |= __shfl_xor(v, o, G);
  return v;
}

// ---------------------------------------------------------------------------------------------
// mj_kinematics (+ link com): serial over the tree on the group's lane 0
template <int NL>
__device__ __forceinline__ void fk_serial(Ws<NL>& w, const KModelDesc* m) {
  const int nl = m->nlink;
  for (int i = 0; i < nl; i++) {
    const int p = m->link_parent[i];
    real pos[3], quat[4];
    real lp[3] = {m->link_pos[i][0], m->link_pos[i][1], m->link_pos[i][2]};
    real lq[4] = {m->link_quat[i][0], m->link_quat[i][1], m->link_quat[i][2], m->link_quat[i][3]};
    real ja[3] = {m->jnt_axis[i][0], m->jnt_axis[i][1], m->jnt_axis[i][2]};
    if (p < 0) {
      pos[0] = lp[0]; pos[1] = lp[1]; pos[2] = lp[2];
      quat[0] = lq[0]; quat[1] = lq[1]; quat[2] = lq[2]; quat[3] = lq[3];
    } else {
      mat_vec3(pos, w.xmat[p], lp);
      pos[0] += w.xpos[p][0]; pos[1] += w.xpos[p][1]; pos[2] += w.xpos[p][2];
      qmul(quat, w.xquat[p], lq);
    }
    const real q = w.qpos[i];
    real mat[9], aw[3];
    if (m->jnt_type[i] == KM_JNT_SLIDE) {
      normalize4(quat);
      quat2mat(mat, quat);
      mat_vec3(aw, mat, ja);
      pos[0] += aw[0] * q; pos[1] += aw[1] * q; pos[2] += aw[2] * q;
    } else {
      real ql[4], qn[4];
      axis_angle2quat(ql, ja, q);
      qmul(qn, quat, ql);
      quat[0] = qn[0]; quat[1] = qn[1]; quat[2] = qn[2]; quat[3] = qn[3];
      normalize4(quat);
      quat2mat(mat, quat);
      mat_vec3(aw, mat, ja);
    }
    real cl[3] = {m->com[i][0], m->com[i][1], m->com[i][2]}, cw[3];
    mat_vec3(cw, mat, cl);
#pragma unroll
    for (int c = 0; c < 3; c++) { w.xpos[i][c] = pos[c]; w.axis[i][c] = aw[c]; w.cpos[i][c] = pos[c] + cw[c]; }
#pragma unroll
    for (int c = 0; c < 4; c++) w.xquat[i][c] = quat[c];
#pragma unroll
    for (int c = 0; c < 9; c++) w.xmat[i][c] = mat[c];
  }
  real cq[4] = {w.qpos[nl + 3], w.qpos[nl + 4], w.qpos[nl + 5], w.qpos[nl + 6]}, cm[9];
  normalize4(cq);
  quat2mat(cm, cq);
#pragma unroll
  for (int c = 0; c < 9; c++) w.cube_mat[c] = cm[c];
}

// column j of the com Jacobian of body b (world frame): linear part jv, angular part jw
template <int NL>
__device__ __forceinline__ void com_jac_col(const Ws<NL>& w, const KModelDesc* m, int b, int j, real* jv, real* jw) {
  if (m->jnt_type[j] == KM_JNT_SLIDE) {
    jv[0] = w.axis[j][0]; jv[1] = w.axis[j][1]; jv[2] = w.axis[j][2];
    jw[0] = 0; jw[1] = 0; jw[2] = 0;
  } else {
    real r[3] = {w.cpos[b][0] - w.xpos[j][0], w.cpos[b][1] - w.xpos[j][1], w.cpos[b][2] - w.xpos[j][2]};
    real ax[3] = {w.axis[j][0], w.axis[j][1], w.axis[j][2]};
    cross3(jv, ax, r);
    jw[0] = ax[0]; jw[1] = ax[1]; jw[2] = ax[2];
  }
}

// M_ij = sum over bodies b below both i and j of  m_b Jv_bi . Jv_bj + Jw_bi . I_b Jw_bj
template <int NL, int G>
__device__ __forceinline__ void mass_matrix(Ws<NL>& w, const KDeviceModel* dm, int sub) {
  const KModelDesc* m = &dm->d;
  const int nl = m->nlink;
  for (int idx = sub; idx < nl * nl; idx += G) {
    const int i = idx / nl, j = idx % nl;
    if (i > j) continue;
    real s = 0;
    if ((dm->x.anc_mask[j] >> i) & 1u) {
      for (int b = j; b < nl; b++) {
        if (!((dm->x.anc_mask[b] >> j) & 1u)) continue;
        real jvi[3], jwi[3], jvj[3], jwj[3];
        com_jac_col<NL>(w, m, b, i, jvi, jwi);
        com_jac_col<NL>(w, m, b, j, jvj, jwj);
        s += m->mass[b] * dot3(jvi, jvj);
        real li[3], lj[3];
        matT_vec3(li, w.xmat[b], jwi);
        matT_vec3(lj, w.xmat[b], jwj);
        s += m->inertia[b][0] * li[0] * lj[0] + m->inertia[b][1] * li[1] * lj[1] + m->inertia[b][2] * li[2] * lj[2];
      }
    }
    w.Minv[i][j] = s; w.Minv[j][i] = s;
  }
}

// velocity-product + gravity wrenches per body (serial forward pass), then bias_j = sum_b J_bj^T wrench_b
template <int NL>
__device__ __forceinline__ void bias_bodies_serial(Ws<NL>& w, const KModelDesc* m) {
  const int nl = m->nlink;
  // reuse Lw rows as scratch for (omega, alpha, a_origin) of each link: 9 numbers per link
  real (*kinv)[NL] = w.Lw;   // flat scratch view
  real* scratch = &kinv[0][0];
  for (int i = 0; i < nl; i++) {
    const int p = m->link_parent[i];
    real wp[3] = {0, 0, 0}, alp[3] = {0, 0, 0}, ap[3] = {-m->gravity[0], -m->gravity[1], -m->gravity[2]}, op[3] = {0, 0, 0};
    if (p >= 0) {
#pragma unroll
      for (int c = 0; c < 3; c++) { wp[c] = scratch[9 * p + c]; alp[c] = scratch[9 * p + 3 + c]; ap[c] = scratch[9 * p + 6 + c]; op[c] = w.xpos[p][c]; }
    }
    real r[3] = {w.xpos[i][0] - op[0], w.xpos[i][1] - op[1], w.xpos[i][2] - op[2]};
    real t1[3], t2[3], ai[3], wi[3], ali[3];
    cross3(t1, alp, r);
    cross3(t2, wp, r); cross3(t2, wp, t2);
    real ax[3] = {w.axis[i][0] * w.qvel[i], w.axis[i][1] * w.qvel[i], w.axis[i][2] * w.qvel[i]}, cz[3];
    cross3(cz, wp, ax);
#pragma unroll
    for (int c = 0; c < 3; c++) { ai[c] = ap[c] + t1[c] + t2[c]; wi[c] = wp[c]; ali[c] = alp[c]; }
    if (m->jnt_type[i] == KM_JNT_SLIDE) {
#pragma unroll
      for (int c = 0; c < 3; c++) ai[c] += 2 * cz[c];
    } else {
#pragma unroll
      for (int c = 0; c < 3; c++) { wi[c] += ax[c]; ali[c] += cz[c]; }
    }
#pragma unroll
    for (int c = 0; c < 3; c++) { scratch[9 * i + c] = wi[c]; scratch[9 * i + 3 + c] = ali[c]; scratch[9 * i + 6 + c] = ai[c]; }
    real cr[3] = {w.cpos[i][0] - w.xpos[i][0], w.cpos[i][1] - w.xpos[i][1], w.cpos[i][2] - w.xpos[i][2]};
    cross3(t1, ali, cr);
    cross3(t2, wi, cr); cross3(t2, wi, t2);
    real wl[3], all[3], Iw[3], nl3[3], nw[3];
    matT_vec3(wl, w.xmat[i], wi);
    matT_vec3(all, w.xmat[i], ali);
#pragma unroll
    for (int c = 0; c < 3; c++) Iw[c] = m->inertia[i][c] * wl[c];
    cross3(nl3, wl, Iw);
#pragma unroll
    for (int c = 0; c < 3; c++) nl3[c] += m->inertia[i][c] * all[c];
    mat_vec3(nw, w.xmat[i], nl3);
#pragma unroll
    for (int c = 0; c < 3; c++) { w.FN[i][c] = m->mass[i] * (ai[c] + t1[c] + t2[c]); w.FN[i][3 + c] = nw[c]; }
  }
  // cube (free joint, qvel = [v_world, w_body]): bias = [-m g, w x I w]
  real wv[3] = {w.qvel[nl + 3], w.qvel[nl + 4], w.qvel[nl + 5]};
  real Iw[3] = {m->cube_inertia[0] * wv[0], m->cube_inertia[1] * wv[1], m->cube_inertia[2] * wv[2]}, t[3];
  cross3(t, wv, Iw);
#pragma unroll
  for (int c = 0; c < 3; c++) { w.bias[nl + c] = -m->cube_mass * m->gravity[c]; w.bias[nl + 3 + c] = t[c]; }
}
template <int NL, int G>
__device__ __forceinline__ void bias_project(Ws<NL>& w, const KDeviceModel* dm, int sub) {
  const KModelDesc* m = &dm->d;
  const int nl = m->nlink;
  for (int j = sub; j < nl; j += G) {
    real s = 0;
    for (int b = j; b < nl; b++) {
      if (!((dm->x.anc_mask[b] >> j) & 1u)) continue;
      real jv[3], jw[3];
      com_jac_col<NL>(w, m, b, j, jv, jw);
      s += jv[0] * w.FN[b][0] + jv[1] * w.FN[b][1] + jv[2] * w.FN[b][2];
      s += jw[0] * w.FN[b][3] + jw[1] * w.FN[b][4] + jw[2] * w.FN[b][5];
    }
    w.bias[j] = s;
  }
}

// Minv <- inverse of the SPD joint-space inertia held in Minv: cooperative left-looking Cholesky
// (lane i owns row i), L^-1 by forward substitution (lane j owns column j), M^-1 = L^-T L^-1.
template <int NL, int G>
__device__ __forceinline__ void invert_mass(Ws<NL>& w, int nl, int sub) {
  real tr = 0;
  for (int i = 0; i < nl; i++) tr += w.Minv[i][i];
  if (sub == 0) w.Mtrace = tr;
  for (int k = 0; k < nl; k++) {
    if (sub >= k && sub < nl) {
      real s = w.Minv[sub][k];
      for (int t = 0; t < k; t++) s -= w.Lw[sub][t] * w.Lw[k][t];
      w.Lw[sub][k] = s;
    }
    GSYNC();
    real dk = w.Lw[k][k];
    if (!(dk > 0)) { w.bad = 1; dk = 1; }
    real d = sqrt(dk);
    GSYNC();
    if (sub == k) w.Lw[k][k] = d;
    else if (sub > k && sub < nl) w.Lw[sub][k] = w.Lw[sub][k] / d;
    GSYNC();
  }
  // lane j: column j of L^-1 into Minv (lower part), x_j = 1/L_jj, x_i = -(sum_{t=j}^{i-1} L_it x_t) / L_ii
  if (sub < nl) {
    const int j = sub;
    for (int i = 0; i < nl; i++) {
      real x = 0;
      if (i == j) x = 1.0 / w.Lw[j][j];
      else if (i > j) {
        real s = 0;
        for (int t = j; t < i; t++) s += w.Lw[i][t] * w.Minv[t][j];
        x = -s / w.Lw[i][i];
      }
      w.Minv[i][j] = x;
    }
  }
  GSYNC();
  // copy L^-1 to Lw, then Minv[i][j] = sum_{t >= max(i,j)} Linv[t][i] Linv[t][j]
  if (sub < nl) for (int i = 0; i < nl; i++) w.Lw[i][sub] = w.Minv[i][sub];
  GSYNC();
  if (sub < nl) {
    const int j = sub;
    for (int i = 0; i < nl; i++) {
      real s = 0;
      for (int t = (i > j ? i : j); t < nl; t++) s += w.Lw[t][i] * w.Lw[t][j];
      w.Minv[i][j] = s;
    }
  }
  GSYNC();
}

// mju_makeFrame
__device__ __forceinline__ void make_frame(real* fr) {
  normalize3(fr);
  real y[3] = {0, 0, 0};
  if (fr[1] < 0.5 && fr[1] > -0.5) y[1] = 1; else y[2] = 1;
  real t = dot3(fr, y);
  y[0] -= t * fr[0]; y[1] -= t * fr[1]; y[2] -= t * fr[2];
  normalize3(y);
  fr[3] = y[0]; fr[4] = y[1]; fr[5] = y[2];
  cross3(fr + 6, fr, fr + 3);
}

// narrow phase for the fixed candidate set: plane-box (first 4 corners below the table), sphere-box, plane-sphere
template <int NL>
__device__ __forceinline__ void collide_serial(Ws<NL>& w, const KModelDesc* m) {
  const int nl = m->nlink;
  int n = 0, cnt = 0;
  uint32_t mask = 0;
  int tfc = 0, tct = 0;
  real cp[3] = {w.qpos[nl], w.qpos[nl + 1], w.qpos[nl + 2]};
  for (int i = 0; i < 8 && cnt < 4; i++) {
    real loc[3] = {(i & 1 ? 1 : -1) * m->cube_half[0], (i & 2 ? 1 : -1) * m->cube_half[1], (i & 4 ? 1 : -1) * m->cube_half[2]}, c[3];
    mat_vec3(c, w.cube_mat, loc);
    c[0] += cp[0]; c[1] += cp[1]; c[2] += cp[2];
    real dist = c[2] - m->table_z;
    if (dist < 0) {
      real fr[9] = {0, 0, 1, 0, 0, 0, 0, 0, 0};
      make_frame(fr);
#pragma unroll
      for (int k = 0; k < 9; k++) w.c_frame[n][k] = fr[k];
      w.c_dist[n] = dist;
      w.c_pos[n][0] = c[0]; w.c_pos[n][1] = c[1]; w.c_pos[n][2] = c[2] - 0.5 * dist;
      w.c_b1[n] = -1; w.c_b2[n] = nl; w.c_dim[n] = 4; w.c_cube[n] = 1;
      mask |= KM_CON_CUBE_TABLE(i); tct = 1; cnt++; n++;
    }
  }
  for (int s = 0; s < m->nsphere; s++) {
    const int l = m->sphere_link[s];
    real sl[3] = {m->sphere_pos[s][0], m->sphere_pos[s][1], m->sphere_pos[s][2]}, ctr[3], rel[3], loc[3], cl[3];
    mat_vec3(ctr, w.xmat[l], sl);
#pragma unroll
    for (int a = 0; a < 3; a++) { ctr[a] += w.xpos[l][a]; rel[a] = ctr[a] - cp[a]; }
    matT_vec3(loc, w.cube_mat, rel);
    bool inside = true;
#pragma unroll
    for (int a = 0; a < 3; a++) { cl[a] = fmin(fmax(loc[a], -m->cube_half[a]), m->cube_half[a]); if (cl[a] != loc[a]) inside = false; }
    real nloc[3], dist;
    const real rad = m->sphere_radius[s];
    if (!inside) {
      nloc[0] = cl[0] - loc[0]; nloc[1] = cl[1] - loc[1]; nloc[2] = cl[2] - loc[2];
      real dn = normalize3(nloc);
      dist = dn - rad;
    } else {
      int best = 0; real bd = INFINITY;
#pragma unroll
      for (int a = 0; a < 3; a++) { real dd = m->cube_half[a] - fabs(loc[a]); if (dd < bd) { bd = dd; best = a; } }
      nloc[0] = 0; nloc[1] = 0; nloc[2] = 0;
      real sg = loc[best] >= 0 ? -1.0 : 1.0;
      if (best == 0) nloc[0] = sg; else if (best == 1) nloc[1] = sg; else nloc[2] = sg;
      dist = -bd - rad;
    }
    if (dist < 0) {
      real fr[9];
      mat_vec3(fr, w.cube_mat, nloc);
      make_frame(fr);
#pragma unroll
      for (int k = 0; k < 9; k++) w.c_frame[n][k] = fr[k];
      w.c_dist[n] = dist;
#pragma unroll
      for (int a = 0; a < 3; a++) w.c_pos[n][a] = ctr[a] + fr[a] * (rad + 0.5 * dist);
      w.c_b1[n] = l; w.c_b2[n] = nl; w.c_dim[n] = 4; w.c_cube[n] = 1;
      mask |= KM_CON_FINGER_CUBE(s); tfc = 1; n++;
    }
  }
  for (int s = 0; s < m->nsphere; s++) {
    const int l = m->sphere_link[s];
    real sl[3] = {m->sphere_pos[s][0], m->sphere_pos[s][1], m->sphere_pos[s][2]}, ctr[3];
    mat_vec3(ctr, w.xmat[l], sl);
#pragma unroll
    for (int a = 0; a < 3; a++) ctr[a] += w.xpos[l][a];
    const real rad = m->sphere_radius[s];
    real dist = ctr[2] - m->table_z - rad;
    if (dist < 0) {
      real fr[9] = {0, 0, 1, 0, 0, 0, 0, 0, 0};
      make_frame(fr);
#pragma unroll
      for (int k = 0; k < 9; k++) w.c_frame[n][k] = fr[k];
      w.c_dist[n] = dist;
      w.c_pos[n][0] = ctr[0]; w.c_pos[n][1] = ctr[1]; w.c_pos[n][2] = ctr[2] - (rad + 0.5 * dist);
      w.c_b1[n] = -1; w.c_b2[n] = l; w.c_dim[n] = 3; w.c_cube[n] = 0;
      mask |= KM_CON_FINGER_TABLE(s); n++;
    }
  }
  w.ncon = n; w.contact_mask = mask; w.touch_fc = tfc; w.touch_ct = tct;
}

// MuJoCo impedance / reference acceleration parameters
__device__ __forceinline__ real impedance(const real* si, real pos) {
  real d0 = fmin(fmax(si[0], MJ_MINIMP), MJ_MAXIMP), dw = fmin(fmax(si[1], MJ_MINIMP), MJ_MAXIMP);
  real width = fmax(MJ_MINVAL, si[2]), mid = fmin(fmax(si[3], MJ_MINIMP), MJ_MAXIMP), power = fmax(1.0, si[4]);
  if (d0 == dw || width <= MJ_MINVAL) return 0.5 * (d0 + dw);
  real x = fabs(pos) / width, y;
  if (x >= 1) return dw;
  if (x <= 0) return d0;
  if (power == 1) y = x;
  else if (x <= mid) y = pow(x, power) / pow(mid, power - 1);
  else y = 1 - pow(1 - x, power) / pow(1 - mid, power - 1);
  return d0 + y * (dw - d0);
}
__device__ __forceinline__ void get_kb(const KModelDesc* m, const real* sr, const real* si, real& kk, real& bb) {
  real tc = fmax(sr[0], 2 * m->timestep), dr = sr[1];
  real dmax = fmin(fmax(si[1], MJ_MINIMP), MJ_MAXIMP);
  bb = 2 / (dmax * tc);
  kk = 1 / (dmax * dmax * tc * tc * dr * dr);
}

// linear/angular velocity Jacobian column of dof j for a world point `pt` fixed to body `body`
template <int NL>
__device__ __forceinline__ void point_jac_col(const Ws<NL>& w, const KDeviceModel* dm, int body, int j, const real* pt,
                                              real* jp, real* jr) {
  const KModelDesc* m = &dm->d;
  const int nl = m->nlink;
  jp[0] = 0; jp[1] = 0; jp[2] = 0; jr[0] = 0; jr[1] = 0; jr[2] = 0;
  if (body < 0) return;
  if (body < nl) {
    if (j >= nl || !((dm->x.anc_mask[body] >> j) & 1u)) return;
    if (m->jnt_type[j] == KM_JNT_SLIDE) { jp[0] = w.axis[j][0]; jp[1] = w.axis[j][1]; jp[2] = w.axis[j][2]; }
    else {
      real ax[3] = {w.axis[j][0], w.axis[j][1], w.axis[j][2]};
      real r[3] = {pt[0] - w.xpos[j][0], pt[1] - w.xpos[j][1], pt[2] - w.xpos[j][2]};
      cross3(jp, ax, r);
      jr[0] = ax[0]; jr[1] = ax[1]; jr[2] = ax[2];
    }
    return;
  }
  if (j < nl) return;
  const int e = j - nl;
  if (e < 3) { jp[e] = 1; return; }
  const int k = e - 3;
  real col[3] = {w.cube_mat[k], w.cube_mat[3 + k], w.cube_mat[6 + k]};
  real r[3] = {pt[0] - w.qpos[nl], pt[1] - w.qpos[nl + 1], pt[2] - w.qpos[nl + 2]};
  cross3(jp, col, r);
  jr[0] = col[0]; jr[1] = col[1]; jr[2] = col[2];
}

// single-dof rows in mj_makeConstraint order (friction loss, then limits), enumerated by lane 0
template <int NL>
__device__ __forceinline__ void scalar_rows_serial(Ws<NL>& w, const KModelDesc* m) {
  const int nl = m->nlink, nv = nl + 6;
  int n = 0;
  for (int j = 0; j < nv; j++) {
    real fl = j < nl ? m->frictionloss[j] : m->cube_frictionloss;
    if (fl > 0) { w.s_dof[n] = j; w.s_type[n] = 0; w.s_sign[n] = 1; w.s_pos[n] = 0; w.s_floss[n] = fl; n++; }
  }
  for (int j = 0; j < nl; j++) {
    real dl = w.qpos[j] - m->jnt_range[j][0], du = m->jnt_range[j][1] - w.qpos[j];
    if (dl < 0) { w.s_dof[n] = j; w.s_type[n] = 1; w.s_sign[n] = 1; w.s_pos[n] = dl; w.s_floss[n] = 0; n++; }
    if (du < 0) { w.s_dof[n] = j; w.s_type[n] = 1; w.s_sign[n] = -1; w.s_pos[n] = du; w.s_floss[n] = 0; n++; }
  }
  w.ns = n;
}

template <int NL, int G>
__device__ __forceinline__ void build_constraints(Ws<NL>& w, const KDeviceModel* dm, int sub) {
  constexpr int NV = Dim<NL>::NV;
  const KModelDesc* m = &dm->d;
  const int nl = m->nlink, nv = nl + 6;
  // single-dof rows: parameters in parallel over rows
  for (int r = sub; r < w.ns; r += G) {
    const int j = w.s_dof[r];
    real Ad = j < nl ? w.Minv[j][j] : (j < nl + 3 ? 1.0 / m->cube_mass : 1.0 / m->cube_inertia[j - nl - 3]);
    real pos = w.s_pos[r];
    real imp = impedance(m->con_def_solimp, pos), kk, bb;
    get_kb(m, m->con_def_solref, m->con_def_solimp, kk, bb);
    const real R = fmax(MJ_MINVAL, (1 - imp) / imp * Ad);
    w.s_R[r] = R;
    w.s_den[r] = Ad + R;
    w.s_inv[r] = 1.0 / (Ad + R);
    w.s_aref[r] = -bb * (w.s_sign[r] * w.qvel[j]) - kk * imp * pos;
  }
  // contact bases J (normal, 2 tangents, torsion): lane j builds column j of every contact
  const int nc = w.ncon;
  for (int j = sub; j < nv; j += G) {
    for (int c = 0; c < nc; c++) {
      real pt[3] = {w.c_pos[c][0], w.c_pos[c][1], w.c_pos[c][2]};
      real p1[3], r1[3], p2[3], r2[3];
      point_jac_col<NL>(w, dm, w.c_b1[c], j, pt, p1, r1);
      point_jac_col<NL>(w, dm, w.c_b2[c], j, pt, p2, r2);
      real dl[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]}, dr[3] = {r2[0] - r1[0], r2[1] - r1[1], r2[2] - r1[2]};
      w.c_Jb[c][0][j] = dot3(w.c_frame[c], dl);
      w.c_Jb[c][1][j] = dot3(w.c_frame[c] + 3, dl);
      w.c_Jb[c][2][j] = dot3(w.c_frame[c] + 6, dl);
      w.c_Jb[c][3][j] = dot3(w.c_frame[c], dr);
    }
  }
  GSYNC();
  // B = M^-1 J^T per basis row: lane i builds component i
  for (int i = sub; i < nv; i += G) {
    for (int c = 0; c < nc; c++) {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        real s;
        if (i < nl) { s = 0; for (int j = 0; j < nl; j++) s += w.Minv[i][j] * w.c_Jb[c][k][j]; }
        else if (i < nl + 3) s = w.c_Jb[c][k][i] / m->cube_mass;
        else s = w.c_Jb[c][k][i] / m->cube_inertia[i - nl - 3];
        w.c_Bb[c][k][i] = s;
      }
    }
  }
  GSYNC();
  // per-contact Gram matrix, edge parameters (pyramidal cone, MuJoCo mj_makeImpedance)
  for (int c = sub; c < nc; c += G) {
    real Gm[4][4], vb[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      real s = 0;
      for (int j = 0; j < nv; j++) s += w.c_Jb[c][k][j] * w.qvel[j];
      vb[k] = s;
#pragma unroll
      for (int l = 0; l < 4; l++) {
        real g = 0;
        for (int j = 0; j < nv; j++) g += w.c_Jb[c][k][j] * w.c_Bb[c][l][j];
        Gm[k][l] = g;
      }
    }
    const real* fr = w.c_cube[c] ? m->con_cube_friction : m->con_def_friction;
    const real* sr = w.c_cube[c] ? m->con_cube_solref : m->con_def_solref;
    const real* si = w.c_cube[c] ? m->con_cube_solimp : m->con_def_solimp;
    real mu[3] = {fr[0], fr[0], fr[1]};
    const real dist = w.c_dist[c];
    real imp = impedance(si, dist), kk, bb;
    get_kb(m, sr, si, kk, bb);
    const int ne = 2 * (w.c_dim[c] - 1);
    ConRec& rc = w.c_rec[c];
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
      for (int l = k; l < 4; l++) rc.G[gidx(k, l)] = 0.5 * (Gm[k][l] + Gm[l][k]);
    rc.mu[0] = mu[0]; rc.mu[1] = mu[1]; rc.mu[2] = mu[2];
    rc.A[0] = -bb * vb[0] - kk * imp * dist;
    rc.A[1] = -bb * vb[1]; rc.A[2] = -bb * vb[2]; rc.A[3] = -bb * vb[3];
    real R = 0;
#pragma unroll
    for (int e = 0; e < 6; e++) {
      const int k = e / 2 + 1;
      const real sm = (e & 1) ? -mu[k - 1] : mu[k - 1];
      real Ad = Gm[0][0] + sm * (Gm[0][k] + Gm[k][0]) + sm * sm * Gm[k][k];
      if (e == 0) { R = 2 * fr[0] * fr[0] * fmax(MJ_MINVAL, (1 - imp) / imp * Ad); rc.R = R; }
      rc.inv[e] = e < ne ? 1.0 / (Ad + R) : 0.0;
      rc.f[e] = 0;
    }
  }
  GSYNC();
  (void)NV;
}

// one Gauss-Seidel update of a non-negative / box-bounded row (returns delta f); inv = 1 / den
__device__ __forceinline__ real pgs_row(real Ja, real aref, real R, real den, real inv, real f, int type, real floss,
                                        real& improvement) {
  const real res = Ja - aref + R * f;
  real fn = f - res * inv;
  if (type == 0) fn = fmin(fmax(fn, -floss), floss);
  else fn = fmax(fn, 0.0);
  const real dlt = fn - f;
  improvement -= dlt * (res + 0.5 * den * dlt);
  return dlt;
}

// mj_step2 up to (not including) integration: actuation, qacc_smooth, warm start, PGS.  Returns this
// lane's component of qacc (lane `sub` owns dof `sub`).
template <int NL, int G>
__device__ __forceinline__ real solve_accel(Ws<NL>& w, const KDeviceModel* dm, int sub, int actuation) {
  const KModelDesc* m = &dm->d;
  const int nl = m->nlink, nv = nl + 6;
  // ---- actuation (position servos on actuator_length = q at mj_step1 time) and smooth acceleration
  real invm = 0;                                  // diagonal of M^-1 for the cube dofs (lane-local)
  if (sub >= nl && sub < nv) invm = sub < nl + 3 ? 1.0 / m->cube_mass : 1.0 / m->cube_inertia[sub - nl - 3];
  if (sub < nv) {
    real rhs = -w.bias[sub];
    if (actuation && sub < nl) {
      real c = fmin(fmax(w.ctrl[sub], m->ctrlrange[sub][0]), m->ctrlrange[sub][1]);
      real force = m->kp[sub] * c - m->kp[sub] * w.qpos[sub];
      if (m->forcelimited[sub]) force = fmin(fmax(force, m->forcerange[sub][0]), m->forcerange[sub][1]);
      rhs += force;
    }
    w.tmp[sub] = rhs;
  }
  GSYNC();
  real a_s = 0;
  if (sub < nl) { for (int j = 0; j < nl; j++) a_s += w.Minv[sub][j] * w.tmp[j]; }
  else if (sub < nv) a_s = w.tmp[sub] * invm;
  if (sub < nv) w.as[sub] = a_s;
  GSYNC();
  const int ns = w.ns, nc = w.ncon;
  // ---- warm start: forces implied by qacc_warmstart, kept only if the dual cost is negative
  real cost_rows = 0;
  for (int r = sub; r < ns; r += G) {
    const int j = w.s_dof[r];
    const real sg = w.s_sign[r], R = w.s_R[r], aref = w.s_aref[r];
    real jar = sg * w.warm[j] - aref, f;
    if (w.s_type[r] == 0) { const real fl = w.s_floss[r]; f = (jar <= -R * fl) ? fl : ((jar >= R * fl) ? -fl : -jar / R); }
    else f = jar < 0 ? -jar / R : 0.0;
    w.s_f[r] = f;
    cost_rows += 0.5 * R * f * f + f * (sg * w.as[j] - aref);
  }
  for (int c = sub; c < nc; c += G) {
    real wk[4], ak[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      real s1 = 0, s2 = 0;
      for (int j = 0; j < nv; j++) { s1 += w.c_Jb[c][k][j] * w.warm[j]; s2 += w.c_Jb[c][k][j] * w.as[j]; }
      wk[k] = s1; ak[k] = s2;
    }
    const int ne = 2 * (w.c_dim[c] - 1);
    ConRec& rc = w.c_rec[c];
    const real R = rc.R;
#pragma unroll
    for (int e = 0; e < 6; e++) {
      const int k = e / 2 + 1;
      const real sm = (e & 1) ? -rc.mu[k - 1] : rc.mu[k - 1];
      const real aref = rc.A[0] + sm * rc.A[k];
      real jar = wk[0] + sm * wk[k] - aref;
      real f = (e < ne && jar < 0) ? -jar / R : 0.0;
      rc.f[e] = f;
      cost_rows += 0.5 * R * f * f + f * (ak[0] + sm * ak[k] - aref);
    }
  }
  GSYNC();
  // y = J^T f (lane j), z = M^-1 y
  real y = 0;
  if (sub < nv) {
    for (int r = 0; r < ns; r++) if (w.s_dof[r] == sub) y += w.s_sign[r] * w.s_f[r];
    for (int c = 0; c < nc; c++) {
      const ConRec& rc = w.c_rec[c];
      real F[4] = {0, 0, 0, 0};
#pragma unroll
      for (int e = 0; e < 6; e++) {
        const int k = e / 2 + 1;
        const real f = rc.f[e];
        F[0] += f;
        F[k] += ((e & 1) ? -rc.mu[k - 1] : rc.mu[k - 1]) * f;
      }
#pragma unroll
      for (int k = 0; k < 4; k++) y += w.c_Jb[c][k][sub] * F[k];
    }
    w.tmp[sub] = y;
  }
  GSYNC();
  real z = 0;
  if (sub < nl) { for (int j = 0; j < nl; j++) z += w.Minv[sub][j] * w.tmp[j]; }
  else if (sub < nv) z = y * invm;
  const real cost = gsum<G>(0.5 * y * z + cost_rows);
  real a = a_s;
  if (cost > 0) {
    for (int r = sub; r < ns; r += G) w.s_f[r] = 0;
    for (int c = sub; c < nc; c += G) for (int e = 0; e < 6; e++) w.c_rec[c].f[e] = 0;
  } else a += z;
  GSYNC();
  // ---- projected Gauss-Seidel in acceleration space: a = a_s + M^-1 J^T f kept distributed (lane = dof).
  // Row order = mj_makeConstraint order.  Rows on the cube's own dofs (its friction loss) touch only the
  // diagonal block of M^-1, so the owning lane updates them locally -- identical to processing them one
  // after another, and no cross-lane traffic.  Rows on arm dofs need one broadcast each; a contact needs
  // four DPP row reductions (its basis projections), then its 4-6 pyramid edges run on the 4x4 Gram form.
  const real scale = 1.0 / (w.Mtrace + 3 * m->cube_mass + m->cube_inertia[0] + m->cube_inertia[1] + m->cube_inertia[2]);
  // the (at most one) cube friction-loss row owned by this lane, kept in registers across sweeps
  int my_row = -1;
  for (int r = 0; r < ns; r++) if (w.s_dof[r] == sub && sub >= nl && w.s_type[r] == 0) my_row = r;
  real my_f = 0, my_aref = 0, my_R = 0, my_den = 1, my_inv = 0, my_fl = 0;
  if (my_row >= 0) {
    my_f = w.s_f[my_row]; my_aref = w.s_aref[my_row]; my_R = w.s_R[my_row]; my_den = w.s_den[my_row];
    my_inv = w.s_inv[my_row]; my_fl = w.s_floss[my_row];
  }
  for (int iter = 0; iter < m->solver_iterations; iter++) {
    real improvement = 0, imp_local = 0;
    for (int r = 0; r < ns; r++) {
      const int j = w.s_dof[r];
      if (j >= nl) continue;                               // cube rows: lane-local, below
      const real sg = w.s_sign[r];
      const real Ja = sg * __shfl(a, j, G);
      const real f = w.s_f[r];
      const real mij = sub < nl ? w.Minv[sub][j] : 0.0;
      const real dlt = pgs_row(Ja, w.s_aref[r], w.s_R[r], w.s_den[r], w.s_inv[r], f, w.s_type[r], w.s_floss[r], improvement);
      w.s_f[r] = f + dlt;
      a += sg * mij * dlt;
    }
    if (my_row >= 0) {
      const real dlt = pgs_row(a, my_aref, my_R, my_den, my_inv, my_f, 0, my_fl, imp_local);
      my_f += dlt;
      a += dlt * invm;
    }
    for (int c = 0; c < nc; c++) {
      // block-load the contact record (group-uniform broadcast reads) and this lane's basis columns
      const ConRec& rr = w.c_rec[c];
      real Gs[10], mu[3], Ab[4], inv[6], f[6];
      const real Rc = rr.R;
#pragma unroll
      for (int i = 0; i < 10; i++) Gs[i] = rr.G[i];
#pragma unroll
      for (int i = 0; i < 3; i++) mu[i] = rr.mu[i];
#pragma unroll
      for (int i = 0; i < 4; i++) Ab[i] = rr.A[i];
#pragma unroll
      for (int i = 0; i < 6; i++) { inv[i] = rr.inv[i]; f[i] = rr.f[i]; }
      real jb[4], bb[4], u[4], Dk[4] = {0, 0, 0, 0};
#pragma unroll
      for (int k = 0; k < 4; k++) { jb[k] = sub < nv ? w.c_Jb[c][k][sub] : 0.0; bb[k] = sub < nv ? w.c_Bb[c][k][sub] : 0.0; }
#pragma unroll
      for (int k = 0; k < 4; k++) u[k] = gsum<G>(jb[k] * a);
#pragma unroll
      for (int e = 0; e < 6; e++) {
        const int k = e / 2 + 1;
        const real sm = (e & 1) ? -mu[k - 1] : mu[k - 1];
        real Ge[4];
#pragma unroll
        for (int l = 0; l < 4; l++) Ge[l] = Gs[gidx(l, 0)] + sm * Gs[gidx(l, k)];   // J_l . (B_0 + sm B_k)
        const real den = Ge[0] + sm * Ge[k] + Rc;
        const real res = (u[0] + sm * u[k]) - (Ab[0] + sm * Ab[k]) + Rc * f[e];
        const real fn = fmax(f[e] - res * inv[e], 0.0);       // inv == 0 for the unused edges of a condim-3 pair
        const real dlt = fn - f[e];
        improvement -= dlt * (res + 0.5 * den * dlt);
        f[e] = fn;
        Dk[0] += dlt; Dk[k] += sm * dlt;
#pragma unroll
        for (int l = 0; l < 4; l++) u[l] += Ge[l] * dlt;
      }
#pragma unroll
      for (int e = 0; e < 6; e++) w.c_rec[c].f[e] = f[e];
#pragma unroll
      for (int k = 0; k < 4; k++) a += bb[k] * Dk[k];
    }
    improvement += gsum<G>(imp_local);
    if (improvement * scale < m->solver_tolerance) break;
  }
  if (my_row >= 0) w.s_f[my_row] = my_f;
  return a;
}

// everything mj_step1 computes that mj_step2 needs, at the state held in w.qpos / w.qvel
template <int NL, int G>
__device__ __forceinline__ void step1_products(Ws<NL>& w, const KDeviceModel* dm, int sub) {
  const KModelDesc* m = &dm->d;
  if (sub == 0) { fk_serial<NL>(w, m); }
  GSYNC();
  if (sub == 0) { bias_bodies_serial<NL>(w, m); }
  if (sub == 1 % G) { collide_serial<NL>(w, m); scalar_rows_serial<NL>(w, m); }
  mass_matrix<NL, G>(w, dm, sub);
  GSYNC();
  bias_project<NL, G>(w, dm, sub);
  invert_mass<NL, G>(w, m->nlink, sub);
  build_constraints<NL, G>(w, dm, sub);
}

// mj_Euler: qvel += dt*qacc, then positions with the NEW velocity (semi-implicit); free-joint quaternion
// integrated on the group's lane 0
template <int NL, int G>
__device__ __forceinline__ void integrate(Ws<NL>& w, const KModelDesc* m, int sub, real a) {
  const int nl = m->nlink, nv = nl + 6;
  const real dt = m->timestep;
  if (sub < nv) {
    real v = w.qvel[sub] + dt * a;
    w.qvel[sub] = v;
    w.warm[sub] = a;
    if (sub < nl + 3) w.qpos[sub] += dt * v;
  }
  GSYNC();
  if (sub == 0) {
    real ax[3] = {w.qvel[nl + 3], w.qvel[nl + 4], w.qvel[nl + 5]};
    real ang = dt * normalize3(ax), qr[4], qn[4];
    real q[4] = {w.qpos[nl + 3], w.qpos[nl + 4], w.qpos[nl + 5], w.qpos[nl + 6]};
    axis_angle2quat(qr, ax, ang);
    normalize4(q);
    qmul(qn, q, qr);
    normalize4(qn);
    w.qpos[nl + 3] = qn[0]; w.qpos[nl + 4] = qn[1]; w.qpos[nl + 5] = qn[2]; w.qpos[nl + 6] = qn[3];
  }
  GSYNC();
}

__device__ __forceinline__ real clip1(real x) { return fmin(fmax(x, -1.0), 1.0); }

// get_observation, env_sim.py:110-146 (state keys; cameras are out of this kernel)
template <int NL, int G>
__device__ __forceinline__ void write_obs(const Ws<NL>& w, const KModelDesc* m, int sub, double* obs_row) {
  const int nl = m->nlink;
  for (int i = sub; i < nl; i += G) {
    obs_row[i] = clip1((w.qpos[i] - m->jnt_range[i][0]) / (m->jnt_range[i][1] - m->jnt_range[i][0]));
    obs_row[nl + i] = clip1(w.qvel[i] / m->max_q_vel);
  }
  for (int c = sub; c < 7; c += G) {
    if (c < 3) obs_row[2 * nl + c] = clip1((w.qpos[nl + c] - m->cube_spawn_lo[c]) / (m->cube_spawn_hi[c] - m->cube_spawn_lo[c]));
    else obs_row[2 * nl + c] = w.qpos[nl + c];
  }
}

// initialize_episode (env_sim.py:23-36) + mj_forward without actuation (dm_control after_reset)
template <int NL, int G>
__device__ __forceinline__ void reset_env(Ws<NL>& w, const KDeviceModel* dm, int sub, uint64_t seed, int64_t genv, int episode) {
  const KModelDesc* m = &dm->d;
  const int nl = m->nlink, nv = nl + 6;
  if (sub < nv) { w.qvel[sub] = 0; w.warm[sub] = 0; }
  if (sub < nl) { w.qpos[sub] = m->q_home[sub]; w.ctrl[sub] = m->q_home[sub]; }
  if (sub == 0) {
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t ctr[4] = {(uint32_t)genv, (uint32_t)((uint64_t)genv >> 32), (uint32_t)episode, 0}, o[4];
    philox4x32_10(ctr, key, o);
    real u0 = u53(o[0], o[1]), u1 = u53(o[2], o[3]);
    ctr[3] = 1;
    philox4x32_10(ctr, key, o);
    real u2 = u53(o[0], o[1]);
    w.qpos[nl] = m->cube_spawn_lo[0] + (m->cube_spawn_hi[0] - m->cube_spawn_lo[0]) * u0;
    w.qpos[nl + 1] = m->cube_spawn_lo[1] + (m->cube_spawn_hi[1] - m->cube_spawn_lo[1]) * u1;
    w.qpos[nl + 2] = m->cube_spawn_lo[2] + (m->cube_spawn_hi[2] - m->cube_spawn_lo[2]) * u2;
    for (int c = 0; c < 4; c++) w.qpos[nl + 3 + c] = m->cube_quat0[c];
    w.bad = 0;
  }
  GSYNC();
  step1_products<NL, G>(w, dm, sub);
  real a = solve_accel<NL, G>(w, dm, sub, 0);
  if (sub < nv) w.warm[sub] = a;
  GSYNC();
}

template <int NL, int G>
__device__ __forceinline__ void load_state(Ws<NL>& w, const KDeviceState& st, int env, int sub, int nl) {
  const int NE = st.num_envs, nv = nl + 6, nq = nl + 7;
  for (int i = sub; i < nq; i += G) w.qpos[i] = st.qpos[(size_t)i * NE + env];
  for (int i = sub; i < nv; i += G) { w.qvel[i] = st.qvel[(size_t)i * NE + env]; w.warm[i] = st.warm[(size_t)i * NE + env]; }
  for (int i = sub; i < nl; i += G) { w.ctrl[i] = st.ctrl[(size_t)i * NE + env]; w.qpos_ik[i] = st.qpos_ik[(size_t)i * NE + env]; }
  if (sub == 0) w.bad = 0;
}
template <int NL, int G>
__device__ __forceinline__ void store_state(const Ws<NL>& w, const KDeviceState& st, int env, int sub, int nl) {
  const int NE = st.num_envs, nv = nl + 6, nq = nl + 7;
  for (int i = sub; i < nq; i += G) st.qpos[(size_t)i * NE + env] = w.qpos[i];
  for (int i = sub; i < nv; i += G) { st.qvel[(size_t)i * NE + env] = w.qvel[i]; st.warm[(size_t)i * NE + env] = w.warm[i]; }
  for (int i = sub; i < nl; i += G) st.ctrl[(size_t)i * NE + env] = w.ctrl[i];
}

// copy the model into LDS with all 64 lanes (8-byte words), then a workgroup barrier (one wave: cheap)
__device__ __forceinline__ void stage_model(KDeviceModel* dst, const KDeviceModel* src) {
  static_assert(sizeof(KDeviceModel) % 8 == 0, "model must be a whole number of 8-byte words");
  const uint64_t* s8 = reinterpret_cast<const uint64_t*>(src);
  uint64_t* d8 = reinterpret_cast<uint64_t*>(dst);
  for (int i = threadIdx.x; i < (int)(sizeof(KDeviceModel) / 8); i += 64) d8[i] = s8[i];
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------
template <int NL, int G>
__global__ __launch_bounds__(64) void k_step(const KDeviceModel* __restrict__ dm_g, KDeviceState st, double* __restrict__ obs,
                                             double* __restrict__ reward, uint8_t* __restrict__ done) {
  constexpr int EPB = 64 / G;
  __shared__ Ws<NL> ws[EPB];
  __shared__ KDeviceModel smodel;     // model constants staged once per workgroup (lane-indexed reads stay on-chip)
  stage_model(&smodel, dm_g);
  const KDeviceModel* dm = &smodel;
  const KModelDesc* m = &dm->d;
  const int lane = threadIdx.x, grp = lane / G, sub = lane % G;
  const int env = blockIdx.x * EPB + grp;
  if (env >= st.num_envs) return;     // whole group exits together
  Ws<NL>& w = ws[grp];
  const int nl = m->nlink, nv = nl + 6;
  load_state<NL, G>(w, st, env, sub, nl);
  GSYNC();
  int bad = 0;
  for (int s = 0; s < m->n_sub_steps; s++) {
    step1_products<NL, G>(w, dm, sub);                 // s == 0: products of the pre-IK state (stale mj_step2)
    real a = solve_accel<NL, G>(w, dm, sub, 1);
    int lb = (sub < nv) && (!isfinite(a) || fabs(a) > 1e10);   // mjWARN_BADQACC
    bad = gor<G>(lb) | w.bad;
    if (bad) break;
    if (s == 0) {                                       // the IK teleported the arm (ik_mujoco.py:34,67)
      if (sub < nl) w.qpos[sub] = w.qpos_ik[sub];
      GSYNC();
    }
    integrate<NL, G>(w, m, sub, a);
  }
  if (!bad) {
    int lb = 0;
    for (int i = sub; i < nl + 7; i += G) lb |= !isfinite(w.qpos[i]);
    bad = gor<G>(lb);
  }
  uint8_t dn = 0;
  real rew = 0;
  double* obs_row = obs + (size_t)env * m->obs_dim;
  if (!bad) {
    // trailing mj_step1: kinematics + collision feed reward and the contact mask
    if (sub == 0) { fk_serial<NL>(w, m); }
    GSYNC();
    if (sub == 0) collide_serial<NL>(w, m);
    // get_reward, env_sim.py:148-179
    real v2 = gsum<G>(sub < nv ? w.qvel[sub] * w.qvel[sub] : 0.0);
    GSYNC();
    rew = -m->reward_vel_penalty * sqrt(v2);
    for (int arm = 1; arm >= 0; arm--) {
      if (!m->arm_present[arm] || !m->arm_has_grip[arm]) continue;
      const int l = m->arm_site_link[arm];
      real so[3] = {m->arm_site_pos[arm][0], m->arm_site_pos[arm][1], m->arm_site_pos[arm][2]}, sp[3];
      mat_vec3(sp, w.xmat[l], so);
      real df[3] = {w.qpos[nl] - (sp[0] + w.xpos[l][0]), w.qpos[nl + 1] - (sp[1] + w.xpos[l][1]), w.qpos[nl + 2] - (sp[2] + w.xpos[l][2])};
      rew += m->reward_grip_dist * (1.0 / (sqrt(dot3(df, df)) + m->epsilon));
    }
    if (m->touch_reward_enabled && w.touch_fc) {
      rew += m->reward_touch_cube;
      if (!w.touch_ct) rew += m->reward_lift_cube;
    }
    write_obs<NL, G>(w, m, sub, obs_row);
    if (sub == 0) st.contact_mask[env] = w.contact_mask;
  } else {
    dn |= KM_DONE_DIVERGED;
    for (int i = sub; i < m->obs_dim; i += G) obs_row[i] = 0;
    if (sub == 0) st.contact_mask[env] = 0;
  }
  int step_idx = st.step_idx[env] + 1;
  int episode = st.episode[env];
  if (step_idx >= m->max_episode_steps) dn |= KM_DONE_TRUNCATED;
  if (dn && (m->auto_reset || bad)) {
    episode += 1; step_idx = 0;
    GSYNC();
    reset_env<NL, G>(w, dm, sub, st.seed, st.env_id_offset + env, episode);
    write_obs<NL, G>(w, m, sub, obs_row);
  }
  if (sub == 0) { reward[env] = rew; done[env] = dn; st.step_idx[env] = step_idx; st.episode[env] = episode; }
  GSYNC();
  store_state<NL, G>(w, st, env, sub, nl);
}

// KManipEnvSim.k_reset for the envs selected by mask (NULL = all) or, with use_done_bits, by nonzero bytes of mask
template <int NL, int G>
__global__ __launch_bounds__(64) void k_reset(const KDeviceModel* __restrict__ dm_g, KDeviceState st,
                                              const uint8_t* __restrict__ mask, double* __restrict__ obs) {
  constexpr int EPB = 64 / G;
  __shared__ Ws<NL> ws[EPB];
  __shared__ KDeviceModel smodel;
  stage_model(&smodel, dm_g);
  const KDeviceModel* dm = &smodel;
  const KModelDesc* m = &dm->d;
  const int lane = threadIdx.x, grp = lane / G, sub = lane % G;
  const int env = blockIdx.x * EPB + grp;
  if (env >= st.num_envs) return;
  if (mask && !mask[env]) return;
  Ws<NL>& w = ws[grp];
  const int nl = m->nlink;
  int episode = st.episode[env] + 1;
  reset_env<NL, G>(w, dm, sub, st.seed, st.env_id_offset + env, episode);
  if (obs) write_obs<NL, G>(w, m, sub, obs + (size_t)env * m->obs_dim);
  if (sub == 0) { st.step_idx[env] = 0; st.episode[env] = episode; st.contact_mask[env] = 0; }
  GSYNC();
  store_state<NL, G>(w, st, env, sub, nl);
}

void kmanip_launch_step(const KDeviceModel* dm, const KModelDesc& hd, const KDeviceState& st, double* obs, double* reward,
                        uint8_t* done, hipStream_t stream) {
  if (hd.nlink <= 10) {
    constexpr int G = 16, EPB = 64 / G;
    hipLaunchKernelGGL((k_step<10, G>), dim3((st.num_envs + EPB - 1) / EPB), dim3(64), 0, stream, dm, st, obs, reward, done);
  } else {
    constexpr int G = 32, EPB = 64 / G;
    hipLaunchKernelGGL((k_step<20, G>), dim3((st.num_envs + EPB - 1) / EPB), dim3(64), 0, stream, dm, st, obs, reward, done);
  }
}
void kmanip_launch_reset(const KDeviceModel* dm, const KModelDesc& hd, const KDeviceState& st, const uint8_t* mask,
                         int use_done_bits, double* obs, hipStream_t stream) {
  (void)use_done_bits;
  if (hd.nlink <= 10) {
    constexpr int G = 16, EPB = 64 / G;
    hipLaunchKernelGGL((k_reset<10, G>), dim3((st.num_envs + EPB - 1) / EPB), dim3(64), 0, stream, dm, st, mask, obs);
  } else {
    constexpr int G = 32, EPB = 64 / G;
    hipLaunchKernelGGL((k_reset<20, G>), dim3((st.num_envs + EPB - 1) / EPB), dim3(64), 0, stream, dm, st, mask, obs);
  }
}
