/*
 * kmanip_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C float64 CPU restatement of gym-kmanip's hot path, used only as the parity checker
 * (tests/, __graft_entry__.smoke()) and as the timed `cpu_baseline` ("port") leg of bench.py.
 * The shipped path (gym_kmanip_amd/csrc) never links, imports or calls this file.
 *
 * PARITY STATUS: the restatements of gym-kmanip's OWN code below (before_step, ik / ik_res / ik_jac, get_observation, get_reward,
 * initialize_episode, the k_step tuple) are pinned to the reference's own Python, run in the build container over stand-ins for
 * its absent third-party packages (tests/tools/refrun.py -> tests/golden/ref_*.npz; tests/test_ref_fixtures.py: ko_step
 * reproduces 419 reference steps with ctrl bit-exact), and the IK's solver to the real scipy.optimize.least_squares.
 * The restatement of MuJoCo's mj_step (CRBA, RNE, constraint rows, solvers, Euler) and of dm_control's step order stays
 * "parity unpinned": the reference publishes no golden vectors (tests/test_env.py:8-24 only runs gymnasium's check_env) and its
 * engine cannot run here (gymnasium / mujoco / dm_control absent, robot meshes git-ignored: SURVEY.md section 8c).
 *
 * Restated reference code (file:line under /root/reference/gym_kmanip/):
 *   env_sim.py:23-36    KManipTask.initialize_episode      -> ko_reset
 *   env_sim.py:38-108   KManipTask.before_step             -> before_step
 *   env_sim.py:110-146  KManipTask.get_observation         -> pack_obs
 *   env_sim.py:148-179  KManipTask.get_reward              -> compute_reward
 *   env_sim.py:196-200  KManipEnvSim.k_step                -> ko_step
 *   ik_mujoco.py:20-53  ik_res                             -> ik_fun
 *   ik_mujoco.py:56-97  ik_jac                             -> ik_jacobian
 *   ik_mujoco.py:100-155 ik                                -> ik_solve
 * Third-party stages restated from their published algorithms (not vendored in the
 * reference; pins: pyproject.toml:14-22 mujoco>=2.3.7, dm-control>=1.0.14, scipy unpinned):
 *   scipy 1.15.3 optimize/_lsq/trf.py trf_bounds + common.py helpers -> trf_bounds (exact SVD
 *     replaced by one-sided Jacobi SVD)
 *   dm_control Physics.step legacy order: mj_step2; (n-1) x mj_step; mj_step1 -> ko_step
 *   MuJoCo mj_kinematics / mj_jacSite / mju_mat2Quat / mju_subQuat / mjd_subQuat, CRBA, RNE,
 *     position actuators, soft constraints (solref/solimp, friction loss, limits, pyramidal
 *     contacts), PGS, semi-implicit Euler -> functions below (DESIGN.md lists each deviation).
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#include "../include/kmanip.h"

#define NLMAX KM_MAX_LINKS
#define NVMAX (KM_MAX_LINKS + 6)
#define NQMAX (KM_MAX_LINKS + 7)
#define NCON_MAX (4 + KM_SPHERE_SLOTS(KM_MAX_LINKS) + KM_SPHERE_TABLE_SLOTS(KM_MAX_LINKS))
#define NEFC_MAX (NVMAX + 2 * KM_MAX_LINKS + 6 * NCON_MAX)
#define MJ_MINVAL 1e-15
#define MJ_MINIMP 0.0001
#define MJ_MAXIMP 0.9999
#define DBL_EPS 2.220446049250313e-16

typedef struct KoState {
  double qpos[NQMAX];
  double qvel[NVMAX];
  double ctrl[NLMAX];
  double qacc_warm[NVMAX];
  double time;
  int32_t step_idx;
  int32_t episode;
} KoState;

typedef struct KoDiag {
  uint32_t contact_mask;
  int32_t ik_nfev[2];
  int32_t ik_status[2];
  int32_t nefc;
  int32_t solver_iter;
  int32_t diverged;
} KoDiag;

/* ------------------------------------------------------------------ small vector helpers */
static double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void cross3(double* r, const double* a, const double* b) {
  double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
static double norm3(const double* a) { return sqrt(dot3(a, a)); }
/* mju_normalize3 */
static double normalize3(double* v) {
  double n = norm3(v);
  if (n < MJ_MINVAL) { v[0] = 1; v[1] = 0; v[2] = 0; }
  else { v[0] /= n; v[1] /= n; v[2] /= n; }
  return n;
}
static void normalize4(double* q) {
  double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < MJ_MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; }
  else { q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n; }
}
static void qmul(double* r, const double* a, const double* b) {
  double w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  double x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  double y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  double z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  r[0] = w; r[1] = x; r[2] = y; r[3] = z;
}
static void quat2mat(double* m, const double* q) {
  double w = q[0], x = q[1], y = q[2], z = q[3];
  m[0] = w * w + x * x - y * y - z * z; m[1] = 2 * (x * y - w * z); m[2] = 2 * (x * z + w * y);
  m[3] = 2 * (x * y + w * z); m[4] = w * w - x * x + y * y - z * z; m[5] = 2 * (y * z - w * x);
  m[6] = 2 * (x * z - w * y); m[7] = 2 * (y * z + w * x); m[8] = w * w - x * x - y * y + z * z;
}
static void mat_vec3(double* r, const double* m, const double* v) {
  double x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2];
  double y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2];
  double z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static void matT_vec3(double* r, const double* m, const double* v) {
  double x = m[0] * v[0] + m[3] * v[1] + m[6] * v[2];
  double y = m[1] * v[0] + m[4] * v[1] + m[7] * v[2];
  double z = m[2] * v[0] + m[5] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static void axis_angle2quat(double* q, const double* axis, double angle) {
  if (angle == 0) { q[0] = 1; q[1] = q[2] = q[3] = 0; return; }
  double s = sin(angle * 0.5);
  q[0] = cos(angle * 0.5); q[1] = axis[0] * s; q[2] = axis[1] * s; q[3] = axis[2] * s;
}
/* mju_mat2Quat */
static void mat2quat(double* q, const double* m) {
  if (m[0] + m[4] + m[8] > 0) {
    q[0] = 0.5 * sqrt(1 + m[0] + m[4] + m[8]);
    q[1] = 0.25 * (m[7] - m[5]) / q[0]; q[2] = 0.25 * (m[2] - m[6]) / q[0]; q[3] = 0.25 * (m[3] - m[1]) / q[0];
  } else if (m[0] > m[4] && m[0] > m[8]) {
    q[1] = 0.5 * sqrt(1 + m[0] - m[4] - m[8]);
    q[0] = 0.25 * (m[7] - m[5]) / q[1]; q[2] = 0.25 * (m[1] + m[3]) / q[1]; q[3] = 0.25 * (m[2] + m[6]) / q[1];
  } else if (m[4] > m[8]) {
    q[2] = 0.5 * sqrt(1 - m[0] + m[4] - m[8]);
    q[0] = 0.25 * (m[2] - m[6]) / q[2]; q[1] = 0.25 * (m[1] + m[3]) / q[2]; q[3] = 0.25 * (m[5] + m[7]) / q[2];
  } else {
    q[3] = 0.5 * sqrt(1 - m[0] - m[4] + m[8]);
    q[0] = 0.25 * (m[3] - m[1]) / q[3]; q[1] = 0.25 * (m[2] + m[6]) / q[3]; q[2] = 0.25 * (m[5] + m[7]) / q[3];
  }
  normalize4(q);
}
/* mju_subQuat: res = 3D velocity taking qb to qa (qb * quat(res) = qa) */
static void sub_quat(double* res, const double* qa, const double* qb) {
  double qneg[4] = {qb[0], -qb[1], -qb[2], -qb[3]}, qdif[4];
  qmul(qdif, qneg, qa);
  double axis[3] = {qdif[1], qdif[2], qdif[3]};
  double sin_a_2 = normalize3(axis);
  double speed = 2 * atan2(sin_a_2, qdif[0]);
  if (speed > M_PI) speed -= 2 * M_PI;
  res[0] = axis[0] * speed; res[1] = axis[1] * speed; res[2] = axis[2] * speed;
}
/* mjd_subQuat, Db only (= -Da^T); row-major 3x3 */
static void d_sub_quat_b(double* Db, const double* qa, const double* qb) {
  double axis[3];
  sub_quat(axis, qa, qb);
  double half = 0.5 * normalize3(axis);
  double K[9] = {0, -axis[2], axis[1], axis[2], 0, -axis[0], -axis[1], axis[0], 0};
  double KK[9];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
    double s = 0; for (int k = 0; k < 3; k++) s += K[3 * i + k] * K[3 * k + j];
    KK[3 * i + j] = s;
  }
  double coef = 1.0 - (half < 6e-8 ? 1.0 : half / tan(half));
  double Da[9];
  for (int i = 0; i < 9; i++) Da[i] = ((i % 4 == 0) ? 1.0 : 0.0) + half * K[i] + coef * KK[i];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Db[3 * i + j] = -Da[3 * j + i];
}

/* ------------------------------------------------------------------ kinematics */
typedef struct Kin {
  double xpos[NLMAX][3];   /* link frame origin (== joint anchor, jnt pos = 0) */
  double xquat[NLMAX][4];
  double xmat[NLMAX][9];
  double axis[NLMAX][3];   /* joint axis, world */
  double cpos[NLMAX][3];   /* link com, world */
  double cube_pos[3], cube_quat[4], cube_mat[9];
} Kin;

/* mj_kinematics restated for 1-joint-per-link trees (ref = 0). */
static void kinematics(const KModelDesc* m, const double* qpos, Kin* k) {
  for (int i = 0; i < m->nlink; i++) {
    int p = m->link_parent[i];
    double pos[3], quat[4];
    if (p < 0) {
      memcpy(pos, m->link_pos[i], sizeof pos);
      memcpy(quat, m->link_quat[i], sizeof quat);
    } else {
      mat_vec3(pos, k->xmat[p], m->link_pos[i]);
      for (int c = 0; c < 3; c++) pos[c] += k->xpos[p][c];
      qmul(quat, k->xquat[p], m->link_quat[i]);
    }
    double mat0[9];
    quat2mat(mat0, quat);
    mat_vec3(k->axis[i], mat0, m->jnt_axis[i]);
    if (m->jnt_type[i] == KM_JNT_SLIDE) {
      for (int c = 0; c < 3; c++) pos[c] += k->axis[i][c] * qpos[i];
    } else {
      double qloc[4], qn[4];
      axis_angle2quat(qloc, m->jnt_axis[i], qpos[i]);
      qmul(qn, quat, qloc);
      memcpy(quat, qn, sizeof quat);
    }
    normalize4(quat);
    memcpy(k->xpos[i], pos, sizeof pos);
    memcpy(k->xquat[i], quat, sizeof quat);
    quat2mat(k->xmat[i], quat);
    mat_vec3(k->cpos[i], k->xmat[i], m->com[i]);
    for (int c = 0; c < 3; c++) k->cpos[i][c] += pos[c];
  }
  const double* qc = qpos + m->nlink;
  memcpy(k->cube_pos, qc, 3 * sizeof(double));
  memcpy(k->cube_quat, qc + 3, 4 * sizeof(double));
  normalize4(k->cube_quat);
  quat2mat(k->cube_mat, k->cube_quat);
}

static void site_pose(const KModelDesc* m, const Kin* k, int arm, double* pos, double* mat) {
  int l = m->arm_site_link[arm];
  mat_vec3(pos, k->xmat[l], m->arm_site_pos[arm]);
  for (int c = 0; c < 3; c++) pos[c] += k->xpos[l][c];
  double q[4];
  qmul(q, k->xquat[l], m->arm_site_quat[arm]);
  normalize4(q);
  quat2mat(mat, q);
}

/* mj_jac for a world point attached to `link`: jacp/jacr are 3 x nlink row-major (robot dofs). */
static void jac_point(const KModelDesc* m, const Kin* k, int link, const double* point,
                      double* jacp, double* jacr) {
  int nl = m->nlink;
  memset(jacp, 0, sizeof(double) * 3 * nl);
  if (jacr) memset(jacr, 0, sizeof(double) * 3 * nl);
  for (int j = link; j >= 0; j = m->link_parent[j]) {
    if (m->jnt_type[j] == KM_JNT_SLIDE) {
      for (int c = 0; c < 3; c++) jacp[c * nl + j] = k->axis[j][c];
    } else {
      double r[3] = {point[0] - k->xpos[j][0], point[1] - k->xpos[j][1], point[2] - k->xpos[j][2]};
      double t[3];
      cross3(t, k->axis[j], r);
      for (int c = 0; c < 3; c++) {
        jacp[c * nl + j] = t[c];
        if (jacr) jacr[c * nl + j] = k->axis[j][c];
      }
    }
  }
}

/* ------------------------------------------------------------------ IK residual / Jacobian */
typedef struct IkProb {
  const KModelDesc* m;
  int arm, n;
  double* qpos;          /* full qpos, MUTATED by every evaluation (ik_mujoco.py:34,67) */
  double goal_pos[3], goal_quat[4];
  double q_prev[KM_MAX_IK], q_home[KM_MAX_IK], lb[KM_MAX_IK], ub[KM_MAX_IK];
} IkProb;

/* ik_mujoco.py:20-53 */
static void ik_fun(IkProb* P, const double* x, double* f) {
  const KModelDesc* m = P->m;
  int n = P->n;
  for (int i = 0; i < n; i++) P->qpos[m->arm_q_id[P->arm][i]] = x[i];
  Kin k;
  kinematics(m, P->qpos, &k);
  double pos[3], mat[9], cur[4], rq[3];
  site_pose(m, &k, P->arm, pos, mat);
  for (int c = 0; c < 3; c++) f[c] = pos[c] - P->goal_pos[c];
  mat2quat(cur, mat);
  sub_quat(rq, P->goal_quat, cur);
  for (int c = 0; c < 3; c++) f[3 + c] = rq[c] * m->ik_res_rad;
  for (int i = 0; i < n; i++) {
    f[6 + i] = m->ik_res_reg_prev * (x[i] - P->q_prev[i]);
    f[6 + n + i] = m->ik_res_reg_home * (x[i] - P->q_home[i]);
  }
}

/* ik_mujoco.py:56-97; J is (6+2n) x n row-major */
static void ik_jacobian(IkProb* P, const double* x, double* J) {
  const KModelDesc* m = P->m;
  int n = P->n, nl = m->nlink;
  for (int i = 0; i < n; i++) P->qpos[m->arm_q_id[P->arm][i]] = x[i];
  Kin k;
  kinematics(m, P->qpos, &k);
  double pos[3], mat[9], cur[4];
  site_pose(m, &k, P->arm, pos, mat);
  double jacp[3 * NLMAX], jacr[3 * NLMAX];
  jac_point(m, &k, m->arm_site_link[P->arm], pos, jacp, jacr);
  mat2quat(cur, mat);
  double Db[9];
  d_sub_quat_b(Db, P->goal_quat, cur);
  /* mat = rad * D_ee^T @ site_xmat^T */
  double T[9];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
    double s = 0;
    for (int c = 0; c < 3; c++) s += Db[3 * c + i] * mat[3 * j + c];
    T[3 * i + j] = m->ik_jac_rad * s;
  }
  memset(J, 0, sizeof(double) * (6 + 2 * n) * n);
  for (int i = 0; i < n; i++) {
    int q = m->arm_q_id[P->arm][i];
    for (int r = 0; r < 3; r++) {
      J[r * n + i] = jacp[r * nl + q];
      J[(3 + r) * n + i] = T[3 * r] * jacr[q] + T[3 * r + 1] * jacr[nl + q] + T[3 * r + 2] * jacr[2 * nl + q];
    }
    J[(6 + i) * n + i] = m->ik_jac_reg;
    J[(6 + n + i) * n + i] = m->ik_jac_reg;
  }
}

/* ------------------------------------------------------------------ scipy TRF restated */
#define TRF_MMAX (6 + 2 * KM_MAX_IK)          /* residuals */
#define TRF_RMAX (TRF_MMAX + KM_MAX_IK)       /* augmented rows */

static double vnorm(const double* a, int n) { double s = 0; for (int i = 0; i < n; i++) s += a[i] * a[i]; return sqrt(s); }
static double vdot(const double* a, const double* b, int n) { double s = 0; for (int i = 0; i < n; i++) s += a[i] * b[i]; return s; }

/* one-sided Jacobi SVD of A (R x n, row-major, overwritten by U*S); V n x n row-major.
 * Outputs singular values s (descending) and suf = (U^T f) * s = A_rot^T f. */
static void svd_jacobi(double* A, int R, int n, const double* f, double* s, double* V, double* suf) {
  for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) V[i * n + j] = (i == j);
  for (int sweep = 0; sweep < 60; sweep++) {
    int rotated = 0;
    for (int p = 0; p < n - 1; p++) for (int q = p + 1; q < n; q++) {
      double al = 0, be = 0, ga = 0;
      for (int r = 0; r < R; r++) { double ap = A[r * n + p], aq = A[r * n + q]; al += ap * ap; be += aq * aq; ga += ap * aq; }
      if (fabs(ga) <= 1e-15 * sqrt(al * be) || ga == 0) continue;
      rotated = 1;
      double zeta = (be - al) / (2 * ga);
      double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1 + zeta * zeta));
      double c = 1 / sqrt(1 + t * t), sn = c * t;
      for (int r = 0; r < R; r++) {
        double ap = A[r * n + p], aq = A[r * n + q];
        A[r * n + p] = c * ap - sn * aq; A[r * n + q] = sn * ap + c * aq;
      }
      for (int r = 0; r < n; r++) {
        double vp = V[r * n + p], vq = V[r * n + q];
        V[r * n + p] = c * vp - sn * vq; V[r * n + q] = sn * vp + c * vq;
      }
    }
    if (!rotated) break;
  }
  double sv[KM_MAX_IK], sf[KM_MAX_IK];
  int ord[KM_MAX_IK];
  for (int j = 0; j < n; j++) {
    double a = 0, b = 0;
    for (int r = 0; r < R; r++) { a += A[r * n + j] * A[r * n + j]; b += A[r * n + j] * f[r]; }
    sv[j] = sqrt(a); sf[j] = b; ord[j] = j;
  }
  for (int i = 0; i < n; i++) for (int j = i + 1; j < n; j++) if (sv[ord[j]] > sv[ord[i]]) { int t = ord[i]; ord[i] = ord[j]; ord[j] = t; }
  double Vt[KM_MAX_IK * KM_MAX_IK];
  memcpy(Vt, V, sizeof(double) * n * n);
  for (int j = 0; j < n; j++) {
    s[j] = sv[ord[j]]; suf[j] = sf[ord[j]];
    for (int r = 0; r < n; r++) V[r * n + j] = Vt[r * n + ord[j]];
  }
}

static int in_bounds(const double* x, const double* lb, const double* ub, int n) {
  for (int i = 0; i < n; i++) if (!(x[i] >= lb[i] && x[i] <= ub[i])) return 0;
  return 1;
}
/* common.py make_strictly_feasible */
static void make_strictly_feasible(double* x, const double* lb, const double* ub, int n, double rstep) {
  for (int i = 0; i < n; i++) {
    double xn = x[i];
    if (rstep == 0) {
      if (x[i] <= lb[i]) xn = nextafter(lb[i], ub[i]);
      if (x[i] >= ub[i]) xn = nextafter(ub[i], lb[i]);
    } else {
      double ld = x[i] - lb[i], ud = ub[i] - x[i];
      double lt = rstep * fmax(1.0, fabs(lb[i])), ut = rstep * fmax(1.0, fabs(ub[i]));
      if (ld <= fmin(ud, lt)) xn = lb[i] + lt;
      if (ud <= fmin(ld, ut)) xn = ub[i] - ut;
    }
    if (xn < lb[i] || xn > ub[i]) xn = 0.5 * (lb[i] + ub[i]);
    x[i] = xn;
  }
}
static void cl_scaling(const double* x, const double* g, const double* lb, const double* ub, int n, double* v, double* dv) {
  for (int i = 0; i < n; i++) {
    v[i] = 1; dv[i] = 0;
    if (g[i] < 0) { v[i] = ub[i] - x[i]; dv[i] = -1; }
    if (g[i] > 0) { v[i] = x[i] - lb[i]; dv[i] = 1; }
  }
}
static double step_size_to_bound(const double* x, const double* s, const double* lb, const double* ub, int n, int* hits) {
  double steps[KM_MAX_IK], mn = INFINITY;
  for (int i = 0; i < n; i++) {
    if (s[i] != 0) steps[i] = fmax((lb[i] - x[i]) / s[i], (ub[i] - x[i]) / s[i]);
    else steps[i] = INFINITY;
    if (steps[i] < mn) mn = steps[i];
  }
  if (hits) for (int i = 0; i < n; i++) hits[i] = (steps[i] == mn) ? (s[i] > 0 ? 1 : (s[i] < 0 ? -1 : 0)) : 0;
  return mn;
}
/* J_h is m x n row-major */
static void jh_mul(const double* Jh, int m, int n, const double* s, double* out) {
  for (int r = 0; r < m; r++) { double a = 0; for (int c = 0; c < n; c++) a += Jh[r * n + c] * s[c]; out[r] = a; }
}
static double evaluate_quadratic(const double* Jh, int m, int n, const double* g, const double* s, const double* diag) {
  double Js[TRF_MMAX];
  jh_mul(Jh, m, n, s, Js);
  double q = vdot(Js, Js, m);
  for (int i = 0; i < n; i++) q += s[i] * diag[i] * s[i];
  return 0.5 * q + vdot(s, g, n);
}
static void build_quadratic_1d(const double* Jh, int m, int n, const double* g, const double* s, const double* diag,
                               const double* s0, double* a, double* b, double* c) {
  double v[TRF_MMAX], u[TRF_MMAX];
  jh_mul(Jh, m, n, s, v);
  double aa = vdot(v, v, m);
  for (int i = 0; i < n; i++) aa += s[i] * diag[i] * s[i];
  aa *= 0.5;
  double bb = vdot(g, s, n), cc = 0;
  if (s0) {
    jh_mul(Jh, m, n, s0, u);
    bb += vdot(u, v, m);
    cc = 0.5 * vdot(u, u, m) + vdot(g, s0, n);
    for (int i = 0; i < n; i++) { bb += s0[i] * diag[i] * s[i]; cc += 0.5 * s0[i] * diag[i] * s0[i]; }
  }
  *a = aa; *b = bb; if (c) *c = cc;
}
static void minimize_quadratic_1d(double a, double b, double lb, double ub, double c, double* t_out, double* y_out) {
  double t[3] = {lb, ub, 0};
  int nt = 2;
  if (a != 0) { double ex = -0.5 * b / a; if (lb < ex && ex < ub) t[nt++] = ex; }
  int best = 0; double yb = 0;
  for (int i = 0; i < nt; i++) { double y = t[i] * (a * t[i] + b) + c; if (i == 0 || y < yb) { yb = y; best = i; } }
  *t_out = t[best]; *y_out = yb;
}
/* common.py solve_lsq_trust_region; uses suf = s*uf */
static void solve_lsq_trust_region(int n, int m, const double* suf, const double* s, const double* V, double Delta,
                                   double* alpha_io, double* p) {
  double tmp[KM_MAX_IK];
  int full_rank = 0;
  if (m >= n) full_rank = s[n - 1] > DBL_EPS * m * s[0];
  if (full_rank) {
    for (int i = 0; i < n; i++) tmp[i] = suf[i] / (s[i] * s[i]);
    for (int r = 0; r < n; r++) { double a = 0; for (int c = 0; c < n; c++) a += V[r * n + c] * tmp[c]; p[r] = -a; }
    if (vnorm(p, n) <= Delta) { *alpha_io = 0.0; return; }
  }
  double alpha_upper = vnorm(suf, n) / Delta, alpha_lower = 0.0, alpha = *alpha_io;
#define PHI(al, phi, phip) do { double pn2 = 0, sm = 0; for (int i_ = 0; i_ < n; i_++) { double de = s[i_] * s[i_] + (al); \
      pn2 += (suf[i_] / de) * (suf[i_] / de); sm += suf[i_] * suf[i_] / (de * de * de); } double pn = sqrt(pn2); \
      phi = pn - Delta; phip = -sm / pn; } while (0)
  if (full_rank) { double phi, phip; PHI(0.0, phi, phip); alpha_lower = -phi / phip; }
  if (!full_rank && alpha == 0) alpha = fmax(0.001 * alpha_upper, sqrt(alpha_lower * alpha_upper));
  for (int it = 0; it < 10; it++) {
    if (alpha < alpha_lower || alpha > alpha_upper) alpha = fmax(0.001 * alpha_upper, sqrt(alpha_lower * alpha_upper));
    double phi, phip;
    PHI(alpha, phi, phip);
    if (phi < 0) alpha_upper = alpha;
    double ratio = phi / phip;
    alpha_lower = fmax(alpha_lower, alpha - ratio);
    alpha -= (phi + Delta) * ratio / Delta;
    if (fabs(phi) < 0.01 * Delta) break;
  }
#undef PHI
  for (int i = 0; i < n; i++) tmp[i] = suf[i] / (s[i] * s[i] + alpha);
  for (int r = 0; r < n; r++) { double a = 0; for (int c = 0; c < n; c++) a += V[r * n + c] * tmp[c]; p[r] = -a; }
  double sc = Delta / vnorm(p, n);
  for (int i = 0; i < n; i++) p[i] *= sc;
  *alpha_io = alpha;
}

/* trf.py select_step */
static double select_step(const double* x, const double* Jh, int m, int n, const double* diag_h, const double* g_h,
                          double* p, double* p_h, const double* d, double Delta, const double* lb, const double* ub,
                          double theta, double* step, double* step_h) {
  double xp[KM_MAX_IK];
  for (int i = 0; i < n; i++) xp[i] = x[i] + p[i];
  if (in_bounds(xp, lb, ub, n)) {
    double pv = evaluate_quadratic(Jh, m, n, g_h, p_h, diag_h);
    memcpy(step, p, sizeof(double) * n); memcpy(step_h, p_h, sizeof(double) * n);
    return -pv;
  }
  int hits[KM_MAX_IK];
  double p_stride = step_size_to_bound(x, p, lb, ub, n, hits);
  double r_h[KM_MAX_IK], r[KM_MAX_IK], x_on_bound[KM_MAX_IK];
  for (int i = 0; i < n; i++) { r_h[i] = hits[i] ? -p_h[i] : p_h[i]; r[i] = d[i] * r_h[i]; }
  for (int i = 0; i < n; i++) { p[i] *= p_stride; p_h[i] *= p_stride; x_on_bound[i] = x[i] + p[i]; }
  /* intersect_trust_region(p_h, r_h, Delta) -> positive root */
  double to_tr;
  {
    double a = vdot(r_h, r_h, n), b = vdot(p_h, r_h, n), c = vdot(p_h, p_h, n) - Delta * Delta;
    double dd = sqrt(b * b - a * c);
    double q = -(b + copysign(dd, b));
    double t1 = q / a, t2 = c / q;
    to_tr = t1 < t2 ? t2 : t1;
  }
  double to_bound = step_size_to_bound(x_on_bound, r, lb, ub, n, NULL);
  double r_stride = fmin(to_bound, to_tr), r_stride_l, r_stride_u;
  if (r_stride > 0) {
    r_stride_l = (1 - theta) * p_stride / r_stride;
    r_stride_u = (r_stride == to_bound) ? theta * to_bound : to_tr;
  } else { r_stride_l = 0; r_stride_u = -1; }
  double r_value;
  if (r_stride_l <= r_stride_u) {
    double a, b, c;
    build_quadratic_1d(Jh, m, n, g_h, r_h, diag_h, p_h, &a, &b, &c);
    minimize_quadratic_1d(a, b, r_stride_l, r_stride_u, c, &r_stride, &r_value);
    for (int i = 0; i < n; i++) { r_h[i] = r_h[i] * r_stride + p_h[i]; r[i] = r_h[i] * d[i]; }
  } else r_value = INFINITY;
  for (int i = 0; i < n; i++) { p[i] *= theta; p_h[i] *= theta; }
  double p_value = evaluate_quadratic(Jh, m, n, g_h, p_h, diag_h);
  double ag_h[KM_MAX_IK], ag[KM_MAX_IK];
  for (int i = 0; i < n; i++) { ag_h[i] = -g_h[i]; ag[i] = d[i] * ag_h[i]; }
  double to_tr2 = Delta / vnorm(ag_h, n);
  double to_bound2 = step_size_to_bound(x, ag, lb, ub, n, NULL);
  double ag_stride = (to_bound2 < to_tr2) ? theta * to_bound2 : to_tr2;
  double a, b, ag_value;
  build_quadratic_1d(Jh, m, n, g_h, ag_h, diag_h, NULL, &a, &b, NULL);
  minimize_quadratic_1d(a, b, 0, ag_stride, 0, &ag_stride, &ag_value);
  for (int i = 0; i < n; i++) { ag_h[i] *= ag_stride; ag[i] *= ag_stride; }
  if (p_value < r_value && p_value < ag_value) {
    memcpy(step, p, sizeof(double) * n); memcpy(step_h, p_h, sizeof(double) * n); return -p_value;
  } else if (r_value < p_value && r_value < ag_value) {
    memcpy(step, r, sizeof(double) * n); memcpy(step_h, r_h, sizeof(double) * n); return -r_value;
  }
  memcpy(step, ag, sizeof(double) * n); memcpy(step_h, ag_h, sizeof(double) * n); return -ag_value;
}

/* trf.py trf_bounds with tr_solver='exact', x_scale=1, loss='linear'.  x: in = strictly feasible
 * start, out = result.x.  Returns status; *nfev_out = function evaluations. */
static int trf_bounds(IkProb* P, double* x, int* nfev_out) {
  const double ftol = 1e-8, xtol = 1e-8, gtol = 1e-8;
  int n = P->n, m = 6 + 2 * n, R = m + n;
  int max_nfev = P->m->ik_max_nfev > 0 ? P->m->ik_max_nfev : 100 * n;   /* least_squares default; KModelDesc.ik_max_nfev: opt-in cap */
  double f[TRF_MMAX], f_new[TRF_MMAX], J[TRF_MMAX * KM_MAX_IK], g[KM_MAX_IK] = {0};
  double v[KM_MAX_IK], dv[KM_MAX_IK], d[KM_MAX_IK], diag_h[KM_MAX_IK], g_h[KM_MAX_IK];
  double Jaug[TRF_RMAX * KM_MAX_IK], Jh[TRF_MMAX * KM_MAX_IK], faug[TRF_RMAX];
  double s[KM_MAX_IK], V[KM_MAX_IK * KM_MAX_IK], suf[KM_MAX_IK];
  double x_new[KM_MAX_IK], step[KM_MAX_IK], step_h[KM_MAX_IK], p[KM_MAX_IK], p_h[KM_MAX_IK];
  const double* lb = P->lb; const double* ub = P->ub;
  ik_fun(P, x, f);
  int nfev = 1;
  ik_jacobian(P, x, J);
  double cost = 0.5 * vdot(f, f, m);
#define GRAD() do { for (int c_ = 0; c_ < n; c_++) { double a_ = 0; for (int r_ = 0; r_ < m; r_++) a_ += J[r_ * n + c_] * f[r_]; g[c_] = a_; } } while (0)
  GRAD();
  cl_scaling(x, g, lb, ub, n, v, dv);
  double Delta;
  { double t[KM_MAX_IK]; for (int i = 0; i < n; i++) t[i] = x[i] / sqrt(v[i]); Delta = vnorm(t, n); if (Delta == 0) Delta = 1.0; }
  double alpha = 0.0;
  int status = -1;
  double cost_new = cost;
  for (;;) {
    cl_scaling(x, g, lb, ub, n, v, dv);
    double g_norm = 0;
    for (int i = 0; i < n; i++) g_norm = fmax(g_norm, fabs(g[i] * v[i]));
    if (g_norm < gtol) status = 1;
    if (status != -1 || nfev == max_nfev) break;
    for (int i = 0; i < n; i++) { d[i] = sqrt(v[i]); diag_h[i] = g[i] * dv[i]; g_h[i] = d[i] * g[i]; }
    for (int r = 0; r < m; r++) for (int c = 0; c < n; c++) { Jh[r * n + c] = J[r * n + c] * d[c]; Jaug[r * n + c] = Jh[r * n + c]; }
    for (int r = 0; r < n; r++) for (int c = 0; c < n; c++) Jaug[(m + r) * n + c] = (r == c) ? sqrt(diag_h[r]) : 0.0;
    memcpy(faug, f, sizeof(double) * m);
    memset(faug + m, 0, sizeof(double) * n);
    svd_jacobi(Jaug, R, n, faug, s, V, suf);
    double theta = fmax(0.995, 1 - g_norm);
    double actual_reduction = -1;
    while (actual_reduction <= 0 && nfev < max_nfev) {
      solve_lsq_trust_region(n, m, suf, s, V, Delta, &alpha, p_h);
      for (int i = 0; i < n; i++) p[i] = d[i] * p_h[i];
      double predicted = select_step(x, Jh, m, n, diag_h, g_h, p, p_h, d, Delta, lb, ub, theta, step, step_h);
      for (int i = 0; i < n; i++) x_new[i] = x[i] + step[i];
      make_strictly_feasible(x_new, lb, ub, n, 0.0);
      ik_fun(P, x_new, f_new);
      nfev++;
      double step_h_norm = vnorm(step_h, n);
      int finite = 1;
      for (int i = 0; i < m; i++) if (!isfinite(f_new[i])) finite = 0;
      if (!finite) { Delta = 0.25 * step_h_norm; continue; }
      cost_new = 0.5 * vdot(f_new, f_new, m);
      actual_reduction = cost - cost_new;
      double ratio, Delta_new = Delta;
      if (predicted > 0) ratio = actual_reduction / predicted;
      else if (predicted == 0 && actual_reduction == 0) ratio = 1;
      else ratio = 0;
      if (ratio < 0.25) Delta_new = 0.25 * step_h_norm;
      else if (ratio > 0.75 && step_h_norm > 0.95 * Delta) Delta_new = Delta * 2.0;
      double step_norm = vnorm(step, n), x_norm = vnorm(x, n);
      int ft = (actual_reduction < ftol * cost) && (ratio > 0.25);
      int xt = step_norm < xtol * (xtol + x_norm);
      if (ft && xt) status = 4; else if (ft) status = 2; else if (xt) status = 3;
      if (status != -1) break;
      alpha *= Delta / Delta_new;
      Delta = Delta_new;
    }
    if (actual_reduction > 0) {
      memcpy(x, x_new, sizeof(double) * n);
      memcpy(f, f_new, sizeof(double) * m);
      cost = cost_new;
      ik_jacobian(P, x, J);
      GRAD();
    }
  }
#undef GRAD
  if (status == -1) status = 0;
  *nfev_out = nfev;
  return status;
}

/* ik_mujoco.py:100-155.  qpos (full) is left at the last evaluated point; q_out = value the
 * reference writes into ctrl[q_mask]. */
static void ik_solve(const KModelDesc* m, int arm, double* qpos, const double* q_prev_full,
                     const double* goal_pos, const double* goal_quat, double* q_out, int* nfev, int* status) {
  IkProb P;
  memset(&P, 0, sizeof P);
  P.m = m; P.arm = arm; P.n = m->arm_nq[arm]; P.qpos = qpos;
  memcpy(P.goal_pos, goal_pos, sizeof P.goal_pos);
  memcpy(P.goal_quat, goal_quat, sizeof P.goal_quat);
  int n = P.n;
  double x[KM_MAX_IK];
  for (int i = 0; i < n; i++) {
    int q = m->arm_q_id[arm][i];
    x[i] = qpos[q];
    P.q_prev[i] = q_prev_full[q];
    P.q_home[i] = m->q_home[q];
    P.lb[i] = m->jnt_range[q][0]; P.ub[i] = m->jnt_range[q][1];
  }
  *nfev = 0; *status = -2;
  /* least_squares raises ValueError("x0 is infeasible") -> "IK failed", q stays (ik_mujoco.py:128-138) */
  if (in_bounds(x, P.lb, P.ub, n)) {
    make_strictly_feasible(x, P.lb, P.ub, n, 1e-10);
    *status = trf_bounds(&P, x, nfev);
  }
  /* velocity clip is a no-op (ik_mujoco.py:140-145); position clip :147-152 */
  for (int i = 0; i < n; i++) q_out[i] = fmin(fmax(x[i], P.lb[i]), P.ub[i]);
}

/* ------------------------------------------------------------------ smooth dynamics */
typedef struct Dyn {
  Kin kin;
  double M[NLMAX * NLMAX];      /* robot joint-space inertia (dense, symmetric) */
  double L[NLMAX * NLMAX];      /* Cholesky factor, lower */
  double bias[NVMAX];
  double act_len[NLMAX];        /* actuator_length = q_j at step1 time */
  /* constraints */
  int nefc, ncon;
  int type[NEFC_MAX];           /* 0 friction loss, 1 unilateral */
  double J[NEFC_MAX][NVMAX];
  double B[NEFC_MAX][NVMAX];    /* M^-1 J^T */
  double Adiag[NEFC_MAX], Rr[NEFC_MAX], aref[NEFC_MAX], floss[NEFC_MAX];
  uint32_t contact_mask;
  int touch_finger_cube, touch_cube_table;   /* any SPHERE on the cube (couples arm and cube dofs) | a cube corner on the table */
} Dyn;

/* CRBA restated with composite (mass, first moment, inertia about world origin). */
static void crba(const KModelDesc* m, const Kin* k, double* M) {
  int nl = m->nlink;
  double cm[NLMAX], ch[NLMAX][3], cI[NLMAX][9];
  for (int i = 0; i < nl; i++) {
    const double* c = k->cpos[i];
    double mi = m->mass[i];
    cm[i] = mi;
    for (int a = 0; a < 3; a++) ch[i][a] = mi * c[a];
    /* I_world = R diag R^T + m (|c|^2 1 - c c^T) */
    const double* Rm = k->xmat[i];
    double cc = dot3(c, c);
    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) {
      double s = 0;
      for (int e = 0; e < 3; e++) s += Rm[3 * a + e] * m->inertia[i][e] * Rm[3 * b + e];
      cI[i][3 * a + b] = s + mi * ((a == b ? cc : 0.0) - c[a] * c[b]);
    }
  }
  for (int i = nl - 1; i >= 0; i--) {
    int p = m->link_parent[i];
    if (p >= 0) {
      cm[p] += cm[i];
      for (int a = 0; a < 3; a++) ch[p][a] += ch[i][a];
      for (int a = 0; a < 9; a++) cI[p][a] += cI[i][a];
    }
  }
  memset(M, 0, sizeof(double) * nl * nl);
  for (int j = 0; j < nl; j++) {
    /* unit acceleration of joint j, subtree(j) moves rigidly: alpha, a_O (accel of world-origin point) */
    double al[3] = {0, 0, 0}, aO[3];
    if (m->jnt_type[j] == KM_JNT_SLIDE) memcpy(aO, k->axis[j], sizeof aO);
    else { memcpy(al, k->axis[j], sizeof al); cross3(aO, k->xpos[j], k->axis[j]); }
    double F[3], N[3], t[3];
    cross3(t, al, ch[j]);
    for (int a = 0; a < 3; a++) F[a] = cm[j] * aO[a] + t[a];
    mat_vec3(N, cI[j], al);
    cross3(t, ch[j], aO);
    for (int a = 0; a < 3; a++) N[a] += t[a];
    for (int i = j; i >= 0; i = m->link_parent[i]) {
      double val;
      if (m->jnt_type[i] == KM_JNT_SLIDE) val = dot3(k->axis[i], F);
      else { cross3(t, k->xpos[i], F); double mo[3] = {N[0] - t[0], N[1] - t[1], N[2] - t[2]}; val = dot3(k->axis[i], mo); }
      M[i * nl + j] = val; M[j * nl + i] = val;
    }
  }
}

static int cholesky(const double* A, double* L, int n) {
  memset(L, 0, sizeof(double) * n * n);
  for (int j = 0; j < n; j++) {
    double s = A[j * n + j];
    for (int k = 0; k < j; k++) s -= L[j * n + k] * L[j * n + k];
    if (!(s > 0)) return -1;
    double d = sqrt(s);
    L[j * n + j] = d;
    for (int i = j + 1; i < n; i++) {
      double t = A[i * n + j];
      for (int k = 0; k < j; k++) t -= L[i * n + k] * L[j * n + k];
      L[i * n + j] = t / d;
    }
  }
  return 0;
}
static void chol_solve(const double* L, int n, double* x) {
  for (int i = 0; i < n; i++) { double s = x[i]; for (int k = 0; k < i; k++) s -= L[i * n + k] * x[k]; x[i] = s / L[i * n + i]; }
  for (int i = n - 1; i >= 0; i--) { double s = x[i]; for (int k = i + 1; k < n; k++) s -= L[k * n + i] * x[k]; x[i] = s / L[i * n + i]; }
}

/* RNE with zero joint acceleration (qfrc_bias), gravity as base acceleration -g. */
static void rne_bias(const KModelDesc* m, const Kin* k, const double* qvel, double* bias) {
  int nl = m->nlink;
  double w[NLMAX][3], al[NLMAX][3], a[NLMAX][3], F[NLMAX][3], N[NLMAX][3];
  for (int i = 0; i < nl; i++) {
    int p = m->link_parent[i];
    double wp[3] = {0, 0, 0}, alp[3] = {0, 0, 0}, ap[3] = {-m->gravity[0], -m->gravity[1], -m->gravity[2]}, op[3] = {0, 0, 0};
    if (p >= 0) { memcpy(wp, w[p], sizeof wp); memcpy(alp, al[p], sizeof alp); memcpy(ap, a[p], sizeof ap); memcpy(op, k->xpos[p], sizeof op); }
    double r[3] = {k->xpos[i][0] - op[0], k->xpos[i][1] - op[1], k->xpos[i][2] - op[2]};
    double t1[3], t2[3];
    cross3(t1, alp, r);
    cross3(t2, wp, r); cross3(t2, wp, t2);
    for (int c = 0; c < 3; c++) a[i][c] = ap[c] + t1[c] + t2[c];
    memcpy(w[i], wp, sizeof wp); memcpy(al[i], alp, sizeof alp);
    double axv[3] = {k->axis[i][0] * qvel[i], k->axis[i][1] * qvel[i], k->axis[i][2] * qvel[i]};
    if (m->jnt_type[i] == KM_JNT_SLIDE) {
      cross3(t1, wp, axv);
      for (int c = 0; c < 3; c++) a[i][c] += 2 * t1[c];
    } else {
      cross3(t1, wp, axv);
      for (int c = 0; c < 3; c++) { w[i][c] += axv[c]; al[i][c] += t1[c]; }
    }
    /* com acceleration, force and moment (about com) */
    double cr[3] = {k->cpos[i][0] - k->xpos[i][0], k->cpos[i][1] - k->xpos[i][1], k->cpos[i][2] - k->xpos[i][2]};
    cross3(t1, al[i], cr);
    cross3(t2, w[i], cr); cross3(t2, w[i], t2);
    for (int c = 0; c < 3; c++) F[i][c] = m->mass[i] * (a[i][c] + t1[c] + t2[c]);
    /* N = I al + w x (I w), I = R diag R^T */
    double wl[3], all[3], Iw[3], Ial[3];
    matT_vec3(wl, k->xmat[i], w[i]); matT_vec3(all, k->xmat[i], al[i]);
    for (int c = 0; c < 3; c++) { Iw[c] = m->inertia[i][c] * wl[c]; Ial[c] = m->inertia[i][c] * all[c]; }
    cross3(t1, wl, Iw);
    for (int c = 0; c < 3; c++) t1[c] += Ial[c];
    mat_vec3(N[i], k->xmat[i], t1);
    /* shift moment to world origin */
    cross3(t2, k->cpos[i], F[i]);
    for (int c = 0; c < 3; c++) N[i][c] += t2[c];
  }
  for (int i = nl - 1; i >= 0; i--) {
    double t[3];
    if (m->jnt_type[i] == KM_JNT_SLIDE) bias[i] = dot3(k->axis[i], F[i]);
    else { cross3(t, k->xpos[i], F[i]); double mo[3] = {N[i][0] - t[0], N[i][1] - t[1], N[i][2] - t[2]}; bias[i] = dot3(k->axis[i], mo); }
    int p = m->link_parent[i];
    if (p >= 0) for (int c = 0; c < 3; c++) { F[p][c] += F[i][c]; N[p][c] += N[i][c]; }
  }
  /* cube: qvel = [v_world, w_body]; bias = [-m g, w x (I w)] */
  const double* wv = qvel + nl + 3;
  double Iw[3] = {m->cube_inertia[0] * wv[0], m->cube_inertia[1] * wv[1], m->cube_inertia[2] * wv[2]}, t[3];
  cross3(t, wv, Iw);
  for (int c = 0; c < 3; c++) { bias[nl + c] = -m->cube_mass * m->gravity[c]; bias[nl + 3 + c] = t[c]; }
}

/* x <- M^-1 x over all nv dofs (robot block via Cholesky, cube block diagonal) */
static void minv_mul(const KModelDesc* m, const Dyn* D, double* x) {
  int nl = m->nlink;
  chol_solve(D->L, nl, x);
  for (int c = 0; c < 3; c++) { x[nl + c] /= m->cube_mass; x[nl + 3 + c] /= m->cube_inertia[c]; }
}

/* ------------------------------------------------------------------ constraints */
static double get_impedance(const double* solimp, double pos) {
  double d0 = fmin(fmax(solimp[0], MJ_MINIMP), MJ_MAXIMP), dw = fmin(fmax(solimp[1], MJ_MINIMP), MJ_MAXIMP);
  double width = fmax(MJ_MINVAL, solimp[2]), mid = fmin(fmax(solimp[3], MJ_MINIMP), MJ_MAXIMP), power = fmax(1.0, solimp[4]);
  if (d0 == dw || width <= MJ_MINVAL) return 0.5 * (d0 + dw);
  double x = fabs(pos) / width, y;
  if (x >= 1) return dw;
  if (x <= 0) return d0;
  if (power == 1) y = x;
  else if (x <= mid) y = pow(x, power) / pow(mid, power - 1);
  else y = 1 - pow(1 - x, power) / pow(1 - mid, power - 1);
  return d0 + y * (dw - d0);
}
static void get_kb(const KModelDesc* m, const double* solref, const double* solimp, double* kk, double* bb) {
  double tc = fmax(solref[0], 2 * m->timestep), dr = solref[1];
  double dmax = fmin(fmax(solimp[1], MJ_MINIMP), MJ_MAXIMP);
  *bb = 2 / (dmax * tc);
  *kk = 1 / (dmax * dmax * tc * tc * dr * dr);
}
/* mju_makeFrame: frame[0:3] = normal given */
static void make_frame(double* fr) {
  normalize3(fr);
  double y[3] = {0, 0, 0};
  if (fr[1] < 0.5 && fr[1] > -0.5) y[1] = 1; else y[2] = 1;
  double t = dot3(fr, y);
  for (int c = 0; c < 3; c++) y[c] -= t * fr[c];
  normalize3(y);
  memcpy(fr + 3, y, sizeof y);
  cross3(fr + 6, fr, fr + 3);
}

/* body ids: -1 world, 0..nl-1 link, nl = cube.  Jacobian of world point on body: jp/jr 3 x nv */
static void body_jac(const KModelDesc* m, const Kin* k, int body, const double* pt, double* jp, double* jr) {
  int nl = m->nlink, nv = nl + 6;
  memset(jp, 0, sizeof(double) * 3 * nv); memset(jr, 0, sizeof(double) * 3 * nv);
  if (body < 0) return;
  if (body < nl) {
    double tp[3 * NLMAX], tr[3 * NLMAX];
    jac_point(m, k, body, pt, tp, tr);
    for (int c = 0; c < 3; c++) for (int j = 0; j < nl; j++) { jp[c * nv + j] = tp[c * nl + j]; jr[c * nv + j] = tr[c * nl + j]; }
    return;
  }
  double r[3] = {pt[0] - k->cube_pos[0], pt[1] - k->cube_pos[1], pt[2] - k->cube_pos[2]};
  for (int c = 0; c < 3; c++) jp[c * nv + nl + c] = 1;
  for (int e = 0; e < 3; e++) {
    double col[3] = {k->cube_mat[e], k->cube_mat[3 + e], k->cube_mat[6 + e]}, t[3];
    cross3(t, col, r);
    for (int c = 0; c < 3; c++) { jp[c * nv + nl + 3 + e] = t[c]; jr[c * nv + nl + 3 + e] = col[c]; }
  }
}

static double dof_invweight(const KModelDesc* m, int j) {
  return j < m->nlink ? m->dof_invweight0[j] : m->cube_invweight0[(j - m->nlink) / 3];
}
/* body ids as in body_jac: -1 world, 0..nl-1 link, nl = cube; k = 0 translational, 1 rotational */
static double body_invweight(const KModelDesc* m, int body, int k) {
  if (body < 0) return 0.0;
  return body < m->nlink ? m->body_invweight0[body][k] : m->cube_invweight0[k];
}

typedef struct Contact { double pos[3], frame[9], dist; int b1, b2, dim; int cube_pair; } Contact;

/* diag_approx = MuJoCo's efc_diagApprox for the row (mj_diagApprox: from dof_invweight0 / body_invweight0, i.e. qpos0-time
 * constants) -- it, not the exact A_ii, sets the regulariser R (mj_makeImpedance); A_ii itself is kept for the PGS update. */
static void add_row(const KModelDesc* m, Dyn* D, int type, const double* Jrow, double pos, const double* qvel,
                    const double* solref, const double* solimp, double floss, int nv, double diag_approx) {
  int i = D->nefc++;
  D->type[i] = type;
  memcpy(D->J[i], Jrow, sizeof(double) * nv);
  memcpy(D->B[i], Jrow, sizeof(double) * nv);
  minv_mul(m, D, D->B[i]);
  D->Adiag[i] = vdot(D->J[i], D->B[i], nv);
  double imp = get_impedance(solimp, pos), kk, bb;
  get_kb(m, solref, solimp, &kk, &bb);
  D->Rr[i] = fmax(MJ_MINVAL, (1 - imp) / imp * diag_approx);
  double vel = vdot(Jrow, qvel, nv);
  D->aref[i] = -bb * vel - kk * imp * pos;
  D->floss[i] = floss;
}

/* the table top is a rectangle (kmanip.h table_rect; examples/4_teleop.py:82-84): a point is over it while its x, y lie inside */
static int over_table(const KModelDesc* m, const double* p) {
  return p[0] >= m->table_rect[0] && p[0] <= m->table_rect[1] && p[1] >= m->table_rect[2] && p[1] <= m->table_rect[3];
}

static void collide(const KModelDesc* m, const Kin* k, Contact* con, int* ncon, uint32_t* mask, int* tfc, int* tct) {
  int n = 0, nl = m->nlink;
  *mask = 0; *tfc = 0; *tct = 0;
  /* plane(table, geom1) - box(cube, geom2): corners below the plane, first 4 */
  int cnt = 0;
  for (int i = 0; i < 8 && cnt < 4; i++) {
    double loc[3] = {(i & 1 ? 1 : -1) * m->cube_half[0], (i & 2 ? 1 : -1) * m->cube_half[1], (i & 4 ? 1 : -1) * m->cube_half[2]}, c[3];
    mat_vec3(c, k->cube_mat, loc);
    for (int a = 0; a < 3; a++) c[a] += k->cube_pos[a];
    double dist = c[2] - m->table_z;
    if (dist < 0 && over_table(m, c)) {
      Contact* ct = &con[n++];
      ct->frame[0] = 0; ct->frame[1] = 0; ct->frame[2] = 1;
      make_frame(ct->frame);
      ct->dist = dist;
      ct->pos[0] = c[0]; ct->pos[1] = c[1]; ct->pos[2] = c[2] - 0.5 * dist;
      ct->b1 = -1; ct->b2 = nl; ct->dim = 4; ct->cube_pair = 1;
      *mask |= KM_CON_CUBE_TABLE(i); *tct = 1; cnt++;
    }
  }
  /* sphere(finger / link, geom1) - box(cube, geom2): the first KM_SPHERE_SLOTS penetrating spheres */
  const int nslot = KM_SPHERE_SLOTS(nl);
  cnt = 0;
  for (int s = 0; s < m->nsphere && cnt < nslot; s++) {
    int l = m->sphere_link[s];
    double ctr[3], rel[3], loc[3], clamped[3], seg[3];
    mat_vec3(ctr, k->xmat[l], m->sphere_pos[s]);
    for (int a = 0; a < 3; a++) ctr[a] += k->xpos[l][a];
    /* capsule section (kmanip.h sphere_seg): the collider is the point of the link's segment closest to the cube centre */
    mat_vec3(seg, k->xmat[l], m->sphere_seg[s]);
    {
      double ss = dot3(seg, seg);
      if (ss > 0) {
        double t = ((k->cube_pos[0] - ctr[0]) * seg[0] + (k->cube_pos[1] - ctr[1]) * seg[1] + (k->cube_pos[2] - ctr[2]) * seg[2]) / ss;
        t = fmin(fmax(t, 0.0), 1.0);
        for (int a = 0; a < 3; a++) ctr[a] += t * seg[a];
      }
    }
    for (int a = 0; a < 3; a++) rel[a] = ctr[a] - k->cube_pos[a];
    matT_vec3(loc, k->cube_mat, rel);
    int inside = 1;
    for (int a = 0; a < 3; a++) {
      clamped[a] = fmin(fmax(loc[a], -m->cube_half[a]), m->cube_half[a]);
      if (clamped[a] != loc[a]) inside = 0;
    }
    double nloc[3], dist;
    if (!inside) {
      for (int a = 0; a < 3; a++) nloc[a] = clamped[a] - loc[a];
      double dn = normalize3(nloc);
      dist = dn - m->sphere_radius[s];
    } else {
      int best = 0; double bd = INFINITY;
      for (int a = 0; a < 3; a++) { double dd = m->cube_half[a] - fabs(loc[a]); if (dd < bd) { bd = dd; best = a; } }
      nloc[0] = nloc[1] = nloc[2] = 0;
      nloc[best] = loc[best] >= 0 ? -1 : 1;
      dist = -bd - m->sphere_radius[s];
    }
    if (dist < 0) {
      Contact* ct = &con[n++];
      mat_vec3(ct->frame, k->cube_mat, nloc);
      make_frame(ct->frame);
      ct->dist = dist;
      for (int a = 0; a < 3; a++) ct->pos[a] = ctr[a] + ct->frame[a] * (m->sphere_radius[s] + 0.5 * dist);
      ct->b1 = l; ct->b2 = nl; ct->dim = 4; ct->cube_pair = 1;
      *mask |= KM_CON_SPHERE_CUBE(s); *tfc = 1; cnt++;
    }
  }
  /* plane(table, geom1) - sphere(finger / link, geom2): the first KM_SPHERE_SLOTS penetrating spheres */
  cnt = 0;
  const int ntab = KM_SPHERE_TABLE_SLOTS(nl);
  for (int s = 0; s < m->nsphere && cnt < ntab; s++) {
    int l = m->sphere_link[s];
    double ctr[3];
    mat_vec3(ctr, k->xmat[l], m->sphere_pos[s]);
    for (int a = 0; a < 3; a++) ctr[a] += k->xpos[l][a];
    double dist = ctr[2] - m->table_z - m->sphere_radius[s];
    if (dist < 0 && over_table(m, ctr)) {
      Contact* ct = &con[n++];
      ct->frame[0] = 0; ct->frame[1] = 0; ct->frame[2] = 1;
      make_frame(ct->frame);
      ct->dist = dist;
      ct->pos[0] = ctr[0]; ct->pos[1] = ctr[1]; ct->pos[2] = ctr[2] - (m->sphere_radius[s] + 0.5 * dist);
      ct->b1 = -1; ct->b2 = l; ct->dim = 3; ct->cube_pair = 0;
      *mask |= KM_CON_SPHERE_TABLE(s); cnt++;
    }
  }
  *ncon = n;
}

/* mj_makeConstraint order: friction loss, limits, contacts */
static void make_constraints(const KModelDesc* m, Dyn* D, const double* qpos, const double* qvel) {
  int nl = m->nlink, nv = nl + 6;
  const Kin* k = &D->kin;
  D->nefc = 0;
  double row[NVMAX];
  for (int j = 0; j < nv; j++) {
    double fl = j < nl ? m->frictionloss[j] : m->cube_frictionloss;
    if (fl > 0) {
      memset(row, 0, sizeof row); row[j] = 1;
      add_row(m, D, 0, row, 0.0, qvel, m->con_def_solref, m->con_def_solimp, fl, nv, dof_invweight(m, j));
    }
  }
  for (int j = 0; j < nl; j++) {
    double dl = qpos[j] - m->jnt_range[j][0], du = m->jnt_range[j][1] - qpos[j];
    if (dl < 0) { memset(row, 0, sizeof row); row[j] = 1; add_row(m, D, 1, row, dl, qvel, m->con_def_solref, m->con_def_solimp, 0, nv, dof_invweight(m, j)); }
    if (du < 0) { memset(row, 0, sizeof row); row[j] = -1; add_row(m, D, 1, row, du, qvel, m->con_def_solref, m->con_def_solimp, 0, nv, dof_invweight(m, j)); }
  }
  Contact con[NCON_MAX];
  collide(m, k, con, &D->ncon, &D->contact_mask, &D->touch_finger_cube, &D->touch_cube_table);
  for (int c = 0; c < D->ncon; c++) {
    Contact* ct = &con[c];
    double jp1[3 * NVMAX], jr1[3 * NVMAX], jp2[3 * NVMAX], jr2[3 * NVMAX];
    body_jac(m, k, ct->b1, ct->pos, jp1, jr1);
    body_jac(m, k, ct->b2, ct->pos, jp2, jr2);
    double basis[4][NVMAX];
    for (int j = 0; j < nv; j++) {
      double dl[3] = {jp2[j] - jp1[j], jp2[nv + j] - jp1[nv + j], jp2[2 * nv + j] - jp1[2 * nv + j]};
      double dr[3] = {jr2[j] - jr1[j], jr2[nv + j] - jr1[nv + j], jr2[2 * nv + j] - jr1[2 * nv + j]};
      basis[0][j] = dot3(ct->frame, dl); basis[1][j] = dot3(ct->frame + 3, dl); basis[2][j] = dot3(ct->frame + 6, dl);
      basis[3][j] = dot3(ct->frame, dr);
    }
    const double* fr = ct->cube_pair ? m->con_cube_friction : m->con_def_friction;
    const double* sr = ct->cube_pair ? m->con_cube_solref : m->con_def_solref;
    const double* si = ct->cube_pair ? m->con_cube_solimp : m->con_def_solimp;
    double mu[3] = {fr[0], fr[0], fr[1]};
    int first = D->nefc;
    /* mj_diagApprox, pyramidal contact: edge j gets tran + mu_{j/2}^2 * (j < 4 ? tran : rot), with tran / rot the summed
     * body_invweight0 of the two bodies (world = 0) */
    double tran = body_invweight(m, ct->b1, 0) + body_invweight(m, ct->b2, 0);
    double rot = body_invweight(m, ct->b1, 1) + body_invweight(m, ct->b2, 1);
    for (int e = 0; e < ct->dim - 1; e++) for (int sgn = 0; sgn < 2; sgn++) {
      for (int j = 0; j < nv; j++) row[j] = basis[0][j] + (sgn ? -1 : 1) * mu[e] * basis[1 + e][j];
      add_row(m, D, 1, row, ct->dist, qvel, sr, si, 0, nv, tran + mu[e] * mu[e] * (e < 2 ? tran : rot));
    }
    /* pyramidal regularisation: every edge gets Rpy = 2 mu^2 R(first edge) */
    double Rpy = 2 * fr[0] * fr[0] * D->Rr[first];
    for (int r = first; r < D->nefc; r++) D->Rr[r] = Rpy;
  }
}

/* mj_step1 products at (qpos, qvel) */
static int step1(const KModelDesc* m, const double* qpos, const double* qvel, Dyn* D, int full) {
  kinematics(m, qpos, &D->kin);
  if (!full) {
    Contact con[NCON_MAX];
    collide(m, &D->kin, con, &D->ncon, &D->contact_mask, &D->touch_finger_cube, &D->touch_cube_table);
    return 0;
  }
  crba(m, &D->kin, D->M);
  if (cholesky(D->M, D->L, m->nlink) != 0) return -1;
  rne_bias(m, &D->kin, qvel, D->bias);
  for (int i = 0; i < m->nlink; i++) D->act_len[i] = qpos[i];
  make_constraints(m, D, qpos, qvel);
  return 0;
}

static int solve_newton(const KModelDesc* m, const Dyn* D, const double* a_s, const double* qacc_warm, double* a);
/* mj_step2 minus integration: actuation, smooth acceleration, constraint solve.  Returns solver iterations. */
static int step2_accel(const KModelDesc* m, const Dyn* D, const double* ctrl, int actuation, double* qacc_warm,
                       double* qacc) {
  int nl = m->nlink, nv = nl + 6, ne = D->nefc;
  double a_s[NVMAX];
  for (int j = 0; j < nv; j++) a_s[j] = -D->bias[j];
  if (actuation) for (int i = 0; i < nl; i++) {
    double c = fmin(fmax(ctrl[i], m->ctrlrange[i][0]), m->ctrlrange[i][1]);
    double force = m->kp[i] * c - m->kp[i] * D->act_len[i];
    if (m->forcelimited[i]) force = fmin(fmax(force, m->forcerange[i][0]), m->forcerange[i][1]);
    a_s[i] += force;
  }
  minv_mul(m, D, a_s);
  if (ne == 0) { memcpy(qacc, a_s, sizeof(double) * nv); memcpy(qacc_warm, a_s, sizeof(double) * nv); return 0; }
  if (m->solver == KM_SOLVER_NEWTON) {
    double an[NVMAX];
    int it = solve_newton(m, D, a_s, qacc_warm, an);
    memcpy(qacc, an, sizeof(double) * nv); memcpy(qacc_warm, an, sizeof(double) * nv);
    return it;
  }
  double f[NEFC_MAX], b[NEFC_MAX];
  /* warmstart: forces from constraintUpdate at qacc_warmstart, kept only if the dual cost is negative */
  double y[NVMAX];
  memset(y, 0, sizeof y);
  for (int i = 0; i < ne; i++) {
    b[i] = vdot(D->J[i], a_s, nv) - D->aref[i];
    double jar = vdot(D->J[i], qacc_warm, nv) - D->aref[i], Dn = 1 / D->Rr[i];
    if (D->type[i] == 0) {
      double fl = D->floss[i];
      if (jar <= -D->Rr[i] * fl) f[i] = fl; else if (jar >= D->Rr[i] * fl) f[i] = -fl; else f[i] = -Dn * jar;
    } else f[i] = jar < 0 ? -Dn * jar : 0;
    for (int j = 0; j < nv; j++) y[j] += D->J[i][j] * f[i];
  }
  double z[NVMAX];
  memcpy(z, y, sizeof(double) * nv);
  minv_mul(m, D, z);
  double cost = 0.5 * vdot(y, z, nv);
  for (int i = 0; i < ne; i++) cost += 0.5 * D->Rr[i] * f[i] * f[i] + f[i] * b[i];
  double a[NVMAX];
  if (cost > 0) { memset(f, 0, sizeof(double) * ne); memcpy(a, a_s, sizeof(double) * nv); }
  else for (int j = 0; j < nv; j++) a[j] = a_s[j] + z[j];
  /* PGS in acceleration space: a = a_s + M^-1 J^T f maintained incrementally */
  double scale = 1.0 / (m->meaninertia * nv);          /* MuJoCo: 1 / (stat.meaninertia * max(1, nv)), a qpos0 constant */
  int iter = 0;
  while (iter < m->solver_iterations) {
    double improvement = 0;
    for (int i = 0; i < ne; i++) {
      double den = D->Adiag[i] + D->Rr[i];
      double res = vdot(D->J[i], a, nv) - D->aref[i] + D->Rr[i] * f[i];
      double fn = f[i] - res / den;
      if (D->type[i] == 0) fn = fmin(fmax(fn, -D->floss[i]), D->floss[i]);
      else fn = fmax(fn, 0.0);
      double dlt = fn - f[i];
      if (dlt != 0) {
        f[i] = fn;
        for (int j = 0; j < nv; j++) a[j] += D->B[i][j] * dlt;
        improvement -= dlt * (res + 0.5 * den * dlt);
      }
    }
    iter++;
    if (improvement * scale < m->solver_tolerance) break;
  }
  memcpy(qacc, a, sizeof(double) * nv);
  memcpy(qacc_warm, a, sizeof(double) * nv);
  return iter;
}


/* ------------------------------------------------------------------ Newton solver (MuJoCo's default)
 * Primal problem (MuJoCo "Computation" chapter): minimise over qacc
 *     1/2 (a - a_s)^T M (a - a_s) + sum_i s_i(J_i a - aref_i)
 * s_i: unilateral rows (limits, pyramid edges) 1/2 D x^2 for x < 0 else 0; friction-loss rows quadratic for
 * |x| < R*fl, linear outside.  Newton direction with the exact Hessian of the active set + exact line search
 * (safeguarded Newton on the piecewise-linear phi'); terminates like mj_solNewton on scaled improvement or
 * gradient < tolerance.  The minimiser is unique (M is SPD), so any converged solver returns the same qacc;
 * the iteration counts are this build's, not MuJoCo's. */
static double row_cost(const Dyn* D, int i, double x, double* f, int* quad) {
  double R = D->Rr[i], Dn = 1 / R;
  if (D->type[i] == 0) {
    double fl = D->floss[i];
    if (x <= -R * fl) { *f = fl; *quad = 0; return fl * (-0.5 * R * fl - x); }
    if (x >= R * fl) { *f = -fl; *quad = 0; return fl * (-0.5 * R * fl + x); }
    *f = -Dn * x; *quad = 1; return 0.5 * Dn * x * x;
  }
  if (x < 0) { *f = -Dn * x; *quad = 1; return 0.5 * Dn * x * x; }
  *f = 0; *quad = 0; return 0;
}
static void mfull_mul(const KModelDesc* m, const Dyn* D, const double* v, double* out) {
  int nl = m->nlink;
  for (int i = 0; i < nl; i++) { double s = 0; for (int j = 0; j < nl; j++) s += D->M[i * nl + j] * v[j]; out[i] = s; }
  for (int c = 0; c < 3; c++) { out[nl + c] = m->cube_mass * v[nl + c]; out[nl + 3 + c] = m->cube_inertia[c] * v[nl + 3 + c]; }
}
/* Dof / row subsets.  Arm and cube dofs meet only in finger-cube contact rows; while no such contact is active the primal
 * cost is the SUM of an arm part and a cube part, two independent strictly convex problems with the same joint minimiser.
 * The solver then minimises them one after the other (same Newton + exact line search, each on its own Hessian block); with a
 * finger-cube contact it runs on everything at once.  dsel[j] / rsel[i] flag the dofs / rows of the current subset. */
static void select_subset(const KModelDesc* m, const Dyn* D, int which, int* dsel, int* rsel) {   /* 0 all, 1 arm, 2 cube */
  int nl = m->nlink, nv = nl + 6;
  for (int j = 0; j < nv; j++) dsel[j] = which == 0 || (which == 1 ? j < nl : j >= nl);
  for (int i = 0; i < D->nefc; i++) {
    int arm = 0, cube = 0;
    for (int j = 0; j < nl; j++) if (D->J[i][j] != 0) arm = 1;
    for (int j = nl; j < nv; j++) if (D->J[i][j] != 0) cube = 1;
    rsel[i] = which == 0 || (which == 1 ? (arm && !cube) : (cube && !arm));
  }
}
/* cost, forces, active set, gradient at a -- over the subset (gradient entries of unselected dofs are left untouched) */
static double newton_eval(const KModelDesc* m, const Dyn* D, const double* a_s, const double* a, double* x, double* f,
                          int* quad, double* Mr, double* grad, const int* dsel, const int* rsel) {
  int nv = m->nlink + 6, ne = D->nefc;
  double r[NVMAX] = {0}, Mfull[NVMAX];      /* (zeroed: gcc cannot see that nv entries get written) */
  for (int j = 0; j < nv; j++) r[j] = a[j] - a_s[j];
  mfull_mul(m, D, r, Mfull);
  double cost = 0;
  for (int j = 0; j < nv; j++) if (dsel[j]) { Mr[j] = Mfull[j]; cost += 0.5 * r[j] * Mr[j]; grad[j] = Mr[j]; }
  for (int i = 0; i < ne; i++) if (rsel[i]) {
    x[i] = vdot(D->J[i], a, nv) - D->aref[i];
    cost += row_cost(D, i, x[i], &f[i], &quad[i]);
    for (int j = 0; j < nv; j++) if (dsel[j]) grad[j] -= D->J[i][j] * f[i];
  }
  return cost;
}
static void ls_deriv(const Dyn* D, int ne, const int* rsel, const double* x, const double* y, double alpha, double gp, double pMp,
                     double* d1, double* d2) {
  double a1 = gp + alpha * pMp, a2 = pMp;
  for (int i = 0; i < ne; i++) if (rsel[i]) {
    double xi = x[i] + alpha * y[i], R = D->Rr[i], Dn = 1 / R;
    if (D->type[i] == 0) {
      double fl = D->floss[i];
      if (xi <= -R * fl) a1 += -fl * y[i];
      else if (xi >= R * fl) a1 += fl * y[i];
      else { a1 += Dn * xi * y[i]; a2 += Dn * y[i] * y[i]; }
    } else if (xi < 0) { a1 += Dn * xi * y[i]; a2 += Dn * y[i] * y[i]; }
  }
  *d1 = a1; *d2 = a2;
}
int newton_trace = 0;   /* debugging aid: tools may set it through ko_set_trace */
void ko_set_trace(int on) { newton_trace = on; }
/* design statistics (tests/tools/newton_stats.py): coupled solves, their iterations, and how many of those iterations (after a
 * solve's first) found the active set of every row that touches an ARM dof unchanged -- the share a partial refactorisation
 * (cube block only) could serve.  [0] coupled solves, [1] their iterations, [2] iterations after the first, [3] of those with the
 * arm-side set unchanged, [4] of those with the whole set unchanged */
long long ko_newton_stats[8];
void ko_get_newton_stats(long long* out, int reset) { memcpy(out, ko_newton_stats, sizeof ko_newton_stats); if (reset) memset(ko_newton_stats, 0, sizeof ko_newton_stats); }
/* Newton iterations on one subset, from a (state arrays x, f, quad, Mr, grad hold the evaluation at a) */
static int newton_run(const KModelDesc* m, const Dyn* D, const double* a_s, double* a, const int* dsel, const int* rsel,
                      double* x, double* f, int* quad, double* Mr, double* grad, double cost) {
  int nl = m->nlink, nv = nl + 6, ne = D->nefc;
  double y[NEFC_MAX], p[NVMAX], Mp[NVMAX], gs[NVMAX];
  double H[NVMAX * NVMAX], Lh[NVMAX * NVMAX];
  double scale = 1.0 / (m->meaninertia * nv);          /* MuJoCo: 1 / (stat.meaninertia * max(1, nv)), a qpos0 constant */
  int iter = 0;
  for (int j = 0; j < nv; j++) gs[j] = dsel[j] ? grad[j] : 0.0;
  if (vnorm(gs, nv) * scale < m->solver_tolerance) return 0;
  int all_dofs = 1, prev_quad[NEFC_MAX];
  for (int j = 0; j < nv; j++) all_dofs &= dsel[j];
  if (all_dofs) ko_newton_stats[0]++;
  while (iter < m->solver_iterations) {
    if (all_dofs) {
      ko_newton_stats[1]++;
      if (iter > 0) {
        int arm_same = 1, all_same = 1;
        for (int i = 0; i < ne; i++) if (rsel[i] && quad[i] != prev_quad[i]) {
          all_same = 0;
          for (int j = 0; j < nl; j++) if (D->J[i][j] != 0) arm_same = 0;
        }
        ko_newton_stats[2]++; ko_newton_stats[3] += arm_same; ko_newton_stats[4] += all_same;
      }
      memcpy(prev_quad, quad, sizeof(int) * ne);
    }
    /* H = M + sum_active D J^T J on the subset's block; unselected dofs get identity rows (p = 0 there) */
    memset(H, 0, sizeof(double) * nv * nv);
    for (int i = 0; i < nl; i++) for (int j = 0; j < nl; j++) H[i * nv + j] = D->M[i * nl + j];
    for (int c = 0; c < 3; c++) { H[(nl + c) * nv + nl + c] = m->cube_mass; H[(nl + 3 + c) * nv + nl + 3 + c] = m->cube_inertia[c]; }
    for (int i = 0; i < ne; i++) if (rsel[i] && quad[i]) {
      double Dn = 1 / D->Rr[i];
      for (int r = 0; r < nv; r++) if (D->J[i][r] != 0) for (int c = 0; c < nv; c++) H[r * nv + c] += Dn * D->J[i][r] * D->J[i][c];
    }
    for (int r = 0; r < nv; r++) if (!dsel[r]) for (int c = 0; c < nv; c++) { H[r * nv + c] = (r == c); H[c * nv + r] = (r == c); }
    if (cholesky(H, Lh, nv) != 0) break;
    for (int j = 0; j < nv; j++) p[j] = dsel[j] ? -grad[j] : 0.0;
    chol_solve(Lh, nv, p);
    /* exact line search on phi(alpha) = cost(a + alpha p) */
    mfull_mul(m, D, p, Mp);
    double gp = 0, pMp = vdot(p, Mp, nv);
    for (int j = 0; j < nv; j++) if (dsel[j]) gp += p[j] * Mr[j];
    for (int i = 0; i < ne; i++) if (rsel[i]) y[i] = vdot(D->J[i], p, nv);
    /* phi'(0) = grad . p; p is the exact Newton direction of the current active set, so the first trial is the full step
     * alpha = 1 (phi''(0) = p^T H p = -grad . p): no evaluation of the rows at alpha = 0 */
    double d1, d2, d10 = 0, alpha = 0, lo = 0, hi = INFINITY;
    for (int j = 0; j < nv; j++) if (dsel[j]) d10 += p[j] * grad[j];
    if (d10 < 0) {
      alpha = 1;
      for (int it = 0; it < 50; it++) {
        ls_deriv(D, ne, rsel, x, y, alpha, gp, pMp, &d1, &d2);
        if (fabs(d1) <= 1e-8 * fabs(d10)) break;      /* MuJoCo's ls_tolerance is 1e-2; the outer Newton absorbs the rest */
        if (d1 < 0) lo = alpha; else hi = alpha;
        if (hi - lo <= 1e-14 * hi) break;               /* bracket collapsed to roundoff */
        if (it == 49) break;
        double an = alpha - d1 / d2;
        if (!(an > lo && an < hi)) an = isfinite(hi) ? 0.5 * (lo + hi) : 2 * alpha + 1;
        alpha = an;
      }
    }
    for (int j = 0; j < nv; j++) a[j] += alpha * p[j];
    double cost_new = newton_eval(m, D, a_s, a, x, f, quad, Mr, grad, dsel, rsel);
    for (int j = 0; j < nv; j++) gs[j] = dsel[j] ? grad[j] : 0.0;
    double improvement = scale * (cost - cost_new), gradient = scale * vnorm(gs, nv);
    if (newton_trace) {
      int nq = 0, nr = 0; for (int i = 0; i < ne; i++) if (rsel[i]) { nq += quad[i]; nr++; }
      fprintf(stderr, "  newton it %d cost %.12e impr %.3e grad %.3e alpha %.6f nquad %d/%d\n", iter, cost_new, improvement, gradient, alpha, nq, nr);
    }
    cost = cost_new;
    iter++;
    if (improvement < m->solver_tolerance || gradient < m->solver_tolerance) break;
  }
  return iter;
}
static int solve_newton(const KModelDesc* m, const Dyn* D, const double* a_s, const double* qacc_warm, double* a) {
  int nv = m->nlink + 6;
  double x[NEFC_MAX], f[NEFC_MAX], Mr[NVMAX], grad[NVMAX];
  int quad[NEFC_MAX], dsel[NVMAX], rsel[NEFC_MAX];
  /* warm start: the better of qacc_warmstart and qacc_smooth (whole-problem costs, as MuJoCo compares them) */
  select_subset(m, D, 0, dsel, rsel);
  double cw = newton_eval(m, D, a_s, qacc_warm, x, f, quad, Mr, grad, dsel, rsel);
  double cs = newton_eval(m, D, a_s, a_s, x, f, quad, Mr, grad, dsel, rsel);
  memcpy(a, (cw < cs) ? qacc_warm : a_s, sizeof(double) * nv);
  const int coupled = D->touch_finger_cube;      /* the contact list decides, like the device's slot mask */
  int it = 0;
  if (coupled) {
    double cost = newton_eval(m, D, a_s, a, x, f, quad, Mr, grad, dsel, rsel);
    it = newton_run(m, D, a_s, a, dsel, rsel, x, f, quad, Mr, grad, cost);
  } else {
    for (int which = 1; which <= 2; which++) {
      select_subset(m, D, which, dsel, rsel);
      double cost = newton_eval(m, D, a_s, a, x, f, quad, Mr, grad, dsel, rsel);
      it += newton_run(m, D, a_s, a, dsel, rsel, x, f, quad, Mr, grad, cost);
    }
  }
  return it;
}

/* mj_Euler: qvel += dt qacc; integrate positions with the new velocity */
static void euler(const KModelDesc* m, double* qpos, double* qvel, const double* qacc) {
  int nl = m->nlink, nv = nl + 6;
  double dt = m->timestep;
  for (int j = 0; j < nv; j++) qvel[j] += dt * qacc[j];
  for (int j = 0; j < nl; j++) qpos[j] += dt * qvel[j];
  double* qc = qpos + nl;
  for (int c = 0; c < 3; c++) qc[c] += dt * qvel[nl + c];
  double ax[3] = {qvel[nl + 3], qvel[nl + 4], qvel[nl + 5]};
  double ang = dt * normalize3(ax), qr[4], qn[4];
  axis_angle2quat(qr, ax, ang);
  normalize4(qc + 3);
  qmul(qn, qc + 3, qr);
  memcpy(qc + 3, qn, sizeof qn);
  normalize4(qc + 3);
}

/* ------------------------------------------------------------------ Philox4x32-10 */
static void philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; r++) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
static double u53(uint32_t hi, uint32_t lo) { return ((double)(hi >> 5) * 67108864.0 + (double)(lo >> 6)) / 9007199254740992.0; }
void ko_philox(const uint32_t* ctr, const uint32_t* key, uint32_t* out) { philox4x32_10(ctr, key, out); }

/* cube spawn: key = seed, counter = (global env id lo/hi, episode, 0) */
static void spawn_uniforms(uint64_t seed, int64_t env_id, int32_t episode, double* u) {
  uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  uint32_t ctr[4] = {(uint32_t)env_id, (uint32_t)((uint64_t)env_id >> 32), (uint32_t)episode, 0}, o[4];
  philox4x32_10(ctr, key, o);
  u[0] = u53(o[0], o[1]); u[1] = u53(o[2], o[3]);
  ctr[3] = 1;
  philox4x32_10(ctr, key, o);
  u[2] = u53(o[0], o[1]);
}

/* action_space.sample() (examples/2_log_with_h5py.py:22-26; every key a Box(-1, 1, float32), env_base.py:151-188) from the
 * counter-based stream SURVEY 8d specifies: key = seed, counter = (global env id lo/hi, episode, 0x10000 + 4 * step + block),
 * four float32 per block; a component = ((r >> 8) - 2^23) * 2^-23, exact in float32.  The device draws identical bits
 * (kmanip_sample_action).  `ahead`: the action the env needs that many control steps from now (TimeLimit-only episodes). */
void ko_sample_action(const KModelDesc* m, uint64_t seed, int64_t env_id0, int n, const KoState* s, int ahead, float* act) {
  uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  int T = m->max_episode_steps;
  for (int e = 0; e < n; e++) {
    int64_t genv = env_id0 + e;
    int s0 = s[e].step_idx + ahead, step = s0 % T, episode = s[e].episode + s0 / T;
    for (int blk = 0; 4 * blk < m->act_dim; blk++) {
      uint32_t ctr[4] = {(uint32_t)genv, (uint32_t)((uint64_t)genv >> 32), (uint32_t)episode, 0x10000u + 4u * (uint32_t)step + (uint32_t)blk}, o[4];
      philox4x32_10(ctr, key, o);
      for (int c = 0; c < 4 && 4 * blk + c < m->act_dim; c++)
        act[(size_t)e * m->act_dim + 4 * blk + c] = (float)((int32_t)(o[c] >> 8) - 8388608) * (1.0f / 8388608.0f);
    }
  }
}

/* ------------------------------------------------------------------ reward / obs */
/* env_sim.py:148-179 (xpos of the eel/eer_site BODIES == site position: site local pos is 0) */
static double compute_reward(const KModelDesc* m, const Dyn* D, const double* qvel) {
  int nv = m->nlink + 6;
  double reward = 0;
  reward -= m->reward_vel_penalty * vnorm(qvel, nv);
  for (int arm = 1; arm >= 0; arm--) {      /* grip_l first, then grip_r */
    if (!m->arm_present[arm] || !m->arm_has_grip[arm]) continue;
    double pos[3], mat[9];
    site_pose(m, &D->kin, arm, pos, mat);
    double df[3] = {D->kin.cube_pos[0] - pos[0], D->kin.cube_pos[1] - pos[1], D->kin.cube_pos[2] - pos[2]};
    reward += m->reward_grip_dist * (1 / (norm3(df) + m->epsilon));
  }
  if (m->touch_reward_enabled && (D->contact_mask & KM_CON_FINGERS_CUBE(m->nlink))) {   /* a FINGER on the cube; palm / link spheres do not count */
    reward += m->reward_touch_cube;
    if (!D->touch_cube_table) reward += m->reward_lift_cube;
  }
  return reward;
}
static double clip1(double x) { return fmin(fmax(x, -1.0), 1.0); }
/* env_sim.py:110-146 */
static void pack_obs(const KModelDesc* m, const double* qpos, const double* qvel, double* obs) {
  int nl = m->nlink;
  for (int i = 0; i < nl; i++) {
    obs[i] = clip1((qpos[i] - m->jnt_range[i][0]) / (m->jnt_range[i][1] - m->jnt_range[i][0]));
    obs[nl + i] = clip1(qvel[i] / m->max_q_vel);
  }
  for (int c = 0; c < 3; c++)
    obs[2 * nl + c] = clip1((qpos[nl + c] - m->cube_spawn_lo[c]) / (m->cube_spawn_hi[c] - m->cube_spawn_lo[c]));
  for (int c = 0; c < 4; c++) obs[2 * nl + 3 + c] = qpos[nl + 3 + c];
}

/* scipy Rotation.from_matrix(M).as_euler("xyz") (extrinsic) followed by from_euler("xyz", e).as_quat()
 * reordered to wxyz (env_sim.py:66-69) */
static void euler_add_to_quat(const double* mat, const double* delta, double* quat_wxyz) {
  double e[3];
  e[0] = atan2(mat[7], mat[8]);
  e[1] = atan2(-mat[6], sqrt(mat[7] * mat[7] + mat[8] * mat[8]));
  e[2] = atan2(mat[3], mat[0]);
  for (int c = 0; c < 3; c++) e[c] += delta[c];
  double qx[4] = {cos(e[0] / 2), sin(e[0] / 2), 0, 0}, qy[4] = {cos(e[1] / 2), 0, sin(e[1] / 2), 0}, qz[4] = {cos(e[2] / 2), 0, 0, sin(e[2] / 2)};
  double t[4];
  qmul(t, qy, qx);
  qmul(quat_wxyz, qz, t);
}
static double f32r(double x) { return (double)(float)x; }

/* ------------------------------------------------------------------ public: reset / step */
int ko_state_size(void) { return (int)sizeof(KoState); }
int ko_diag_size(void) { return (int)sizeof(KoDiag); }
int ko_model_desc_size(void) { return (int)sizeof(KModelDesc); }

/* dm_control Physics.after_reset(): mj_forward with actuation disabled at the state in s (qacc_warmstart <- its qacc) */
int ko_after_reset(const KModelDesc* m, KoState* s) {
  Dyn* D = (Dyn*)malloc(sizeof(Dyn));
  int rc = step1(m, s->qpos, s->qvel, D, 1);
  if (rc == 0) {
    double qacc[NVMAX];
    step2_accel(m, D, s->ctrl, 0, s->qacc_warm, qacc);
  }
  free(D);
  return rc;
}
/* env_sim.py:23-36 + dm_control reset_context/after_reset (mj_forward with actuation disabled) */
int ko_reset(const KModelDesc* m, uint64_t seed, int64_t env_id, int32_t episode, KoState* s, double* obs) {
  int nl = m->nlink, nv = nl + 6;
  memset(s, 0, sizeof *s);
  s->episode = episode;
  for (int i = 0; i < nl; i++) { s->qpos[i] = m->q_home[i]; s->ctrl[i] = m->q_home[i]; }
  double u[3];
  spawn_uniforms(seed, env_id, episode, u);
  for (int c = 0; c < 3; c++) s->qpos[nl + c] = m->cube_spawn_lo[c] + (m->cube_spawn_hi[c] - m->cube_spawn_lo[c]) * u[c];
  memcpy(s->qpos + nl + 3, m->cube_quat0, 4 * sizeof(double));
  int rc = ko_after_reset(m, s);
  (void)nv;
  if (obs) pack_obs(m, s->qpos, s->qvel, obs);
  return rc;
}

/* env_sim.py:38-108; P0 = step1 products at the pre-IK state */
static void before_step(const KModelDesc* m, KoState* s, const Dyn* P0, const float* act, KoDiag* dg) {
  int nl = m->nlink;
  double q_pos[NQMAX], ctrl[NLMAX];
  memcpy(q_pos, s->qpos, sizeof(double) * (nl + 7));
  for (int i = 0; i < nl; i++) ctrl[i] = f32r(s->ctrl[i]);        /* .astype(float32), env_sim.py:40 */
  static const int grip_key[2] = {KM_ACT_GRIP_R, KM_ACT_GRIP_L};
  static const int pos_key[2] = {KM_ACT_EER_POS, KM_ACT_EEL_POS};
  static const int orn_key[2] = {KM_ACT_EER_ORN, KM_ACT_EEL_ORN};
  static const int qp_key[2] = {KM_ACT_QPOS_R, KM_ACT_QPOS_L};
  for (int arm = 0; arm < 2; arm++) {                               /* grip_r, then grip_l */
    int c = m->act_col[grip_key[arm]];
    if (c < 0) continue;
    /* float32 array * python float -> float32; "+=" with a float64 scalar computes in float64 and
     * stores float32; np.clip with python-float bounds stays float32 (env_sim.py:43-51) */
    float g = (float)act[c] * (float)m->ee_s_delta;
    g = (float)((double)g + s->qpos[m->arm_grip_id[arm][0]]);
    g = fminf(fmaxf(g, (float)m->ee_s_min), (float)m->ee_s_max);
    ctrl[m->arm_grip_id[arm][0]] = (double)g;
    ctrl[m->arm_grip_id[arm][1]] = (double)g;
  }
  for (int arm = 0; arm < 2; arm++) {                               /* eer (right) IK, then eel */
    int cp = m->act_col[pos_key[arm]], co = m->act_col[orn_key[arm]];
    if (cp < 0) continue;
    double spos[3], smat[9], goal_pos[3], dorn[3], goal_quat[4];
    site_pose(m, &P0->kin, arm, spos, smat);
    for (int c = 0; c < 3; c++) {
      goal_pos[c] = (double)act[cp + c] * m->ee_pos_delta[c] + spos[c];   /* float32 * float64 array -> float64 */
      dorn[c] = (double)act[co + c] * m->ee_orn_delta[c];
    }
    euler_add_to_quat(smat, dorn, goal_quat);
    double qo[KM_MAX_IK];
    int nfev, st;
    ik_solve(m, arm, s->qpos, q_pos, goal_pos, goal_quat, qo, &nfev, &st);
    for (int i = 0; i < m->arm_nq[arm]; i++) ctrl[m->arm_q_id[arm][i]] = f32r(qo[i]);
    if (dg) { dg->ik_nfev[arm] = nfev; dg->ik_status[arm] = st; }
  }
  for (int arm = 0; arm < 2; arm++) {                               /* q_pos_r, q_pos_l joint-delta modes */
    int c = m->act_col[qp_key[arm]];
    if (c < 0) continue;
    for (int i = 0; i < m->arm_nq[arm]; i++) {
      int q = m->arm_q_id[arm][i];
      ctrl[q] = f32r(q_pos[q] + (double)(float)((float)act[c + i] * (float)m->q_pos_delta));
    }
  }
  /* CTRL_ALPHA = 1.0: identity filter (env_sim.py:106); set_control */
  memcpy(s->ctrl, ctrl, sizeof(double) * nl);
}

/* dm_control Physics.step(n), legacy order (mj_step2; (n-1) x mj_step; mj_step1): D holds the mj_step1 products the first
 * mj_step2 consumes (those of the state BEFORE the IK moved qpos); the trailing mj_step1 is left to the caller.  Returns 1 on
 * mjWARN_BADQACC / a non-finite qpos. */
static int physics_step(const KModelDesc* m, KoState* s, Dyn* D, int nsub, KoDiag* dg) {
  int nl = m->nlink, nv = nl + 6, bad = 0;
  double qacc[NVMAX];
  for (int sub = 0; sub < nsub && !bad; sub++) {
    if (sub > 0 && step1(m, s->qpos, s->qvel, D, 1) != 0) { bad = 1; break; }
    int it = step2_accel(m, D, s->ctrl, 1, s->qacc_warm, qacc);
    if (dg) { dg->solver_iter += it; dg->nefc = D->nefc; }
    for (int j = 0; j < nv; j++) if (!isfinite(qacc[j]) || fabs(qacc[j]) > 1e10) bad = 1;   /* mjWARN_BADQACC */
    if (bad) break;
    euler(m, s->qpos, s->qvel, qacc);
    s->time += m->timestep;
  }
  if (!bad) {
    for (int j = 0; j < nl + 7; j++) if (!isfinite(s->qpos[j])) bad = 1;
  }
  return bad;
}
/* Physics.step(nsub) alone, for tests/tools/refrun.py (the reference's own Python running over this physics): qpos_step1 = the
 * qpos the previous mj_step1 saw (the IK has since written data.qpos: ik_mujoco.py:34,67).  contact_mask / touch flags are the
 * trailing mj_step1's.  Returns 1 if the state diverged. */
int ko_physics_step(const KModelDesc* m, KoState* s, const double* qpos_step1, int nsub, uint32_t* contact_mask,
                    int32_t* touch_finger_cube, int32_t* touch_cube_table) {
  Dyn* D = (Dyn*)malloc(sizeof(Dyn));
  int bad = step1(m, qpos_step1, s->qvel, D, 1) != 0;
  if (!bad) bad = physics_step(m, s, D, nsub, NULL);
  if (!bad) {
    step1(m, s->qpos, s->qvel, D, 0);
    if (contact_mask) *contact_mask = D->contact_mask;
    if (touch_finger_cube) *touch_finger_cube = D->touch_finger_cube;
    if (touch_cube_table) *touch_cube_table = D->touch_cube_table;
  }
  free(D);
  return bad;
}
/* collision at a state (what a trailing mj_step1 / mj_forward leaves in data.contact) */
void ko_contact_mask(const KModelDesc* m, const double* qpos, uint32_t* contact_mask, int32_t* touch_finger_cube,
                     int32_t* touch_cube_table) {
  Dyn* D = (Dyn*)malloc(sizeof(Dyn));
  double qv[NVMAX] = {0};
  step1(m, qpos, qv, D, 0);
  *contact_mask = D->contact_mask;
  if (touch_finger_cube) *touch_finger_cube = D->touch_finger_cube;
  if (touch_cube_table) *touch_cube_table = D->touch_cube_table;
  free(D);
}
/* get_reward / get_observation at a state (env_sim.py:110-179), for the fixtures made from the reference's own Python */
void ko_observe(const KModelDesc* m, const double* qpos, const double* qvel, double* obs, double* reward) {
  Dyn* D = (Dyn*)malloc(sizeof(Dyn));
  step1(m, qpos, qvel, D, 0);
  if (reward) *reward = compute_reward(m, D, qvel);
  if (obs) pack_obs(m, qpos, qvel, obs);
  free(D);
}

/* KManipEnvSim.k_step: one control step.  done bits: KM_DONE_*.  If auto_reset is set in the
 * desc and done != 0 the state is reset (episode + 1) and obs is the new episode's first obs. */
int ko_step(const KModelDesc* m, uint64_t seed, int64_t env_id, KoState* s, const float* act, double* obs,
            double* reward, uint8_t* done, KoDiag* dg) {
  Dyn* D = (Dyn*)malloc(sizeof(Dyn));
  KoDiag local;
  if (!dg) dg = &local;
  memset(dg, 0, sizeof *dg);
  dg->ik_status[0] = dg->ik_status[1] = -3;
  int bad = 0;
  /* trailing mj_step1 of the previous step, recomputed from the stored (pre-IK) state */
  if (step1(m, s->qpos, s->qvel, D, 1) != 0) bad = 1;
  if (!bad) {
    before_step(m, s, D, act, dg);
    bad = physics_step(m, s, D, m->n_sub_steps, dg);
  }
  uint8_t dn = 0;
  if (!bad) {
    step1(m, s->qpos, s->qvel, D, 0);
    dg->contact_mask = D->contact_mask;
    *reward = compute_reward(m, D, s->qvel);
    pack_obs(m, s->qpos, s->qvel, obs);
  } else {
    *reward = 0; dn |= KM_DONE_DIVERGED; dg->diverged = 1;
    memset(obs, 0, sizeof(double) * m->obs_dim);
  }
  s->step_idx += 1;
  if (s->step_idx >= m->max_episode_steps) dn |= KM_DONE_TRUNCATED;
  *done = dn;
  free(D);
  if (dn && (m->auto_reset || bad)) {
    int ep = s->episode + 1;
    ko_reset(m, seed, env_id, ep, s, obs);
  }
  return 0;
}

/* batched step over n envs, optionally multi-threaded (OpenMP) -- the cpu_baseline leg */
int ko_step_batch(const KModelDesc* m, uint64_t seed, int64_t env_id0, int n, KoState* s, const float* act,
                  double* obs, double* reward, uint8_t* done, KoDiag* dg, int nthreads) {
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads) schedule(static)
#endif
  for (int e = 0; e < n; e++)
    ko_step(m, seed, env_id0 + e, &s[e], act + (size_t)e * m->act_dim, obs + (size_t)e * m->obs_dim,
            &reward[e], &done[e], dg ? &dg[e] : NULL);
  (void)nthreads;
  return 0;
}
int ko_reset_batch(const KModelDesc* m, uint64_t seed, int64_t env_id0, int n, const int32_t* episode, KoState* s, double* obs) {
  for (int e = 0; e < n; e++) ko_reset(m, seed, env_id0 + e, episode ? episode[e] : 0, &s[e], obs ? obs + (size_t)e * m->obs_dim : NULL);
  return 0;
}

/* ------------------------------------------------------------------ public: pieces for unit tests */
void ko_fk(const KModelDesc* m, const double* qpos, double* xpos, double* xquat, double* site_pos, double* site_mat) {
  Kin k;
  kinematics(m, qpos, &k);
  for (int i = 0; i < m->nlink; i++) { memcpy(xpos + 3 * i, k.xpos[i], 24); memcpy(xquat + 4 * i, k.xquat[i], 32); }
  for (int a = 0; a < 2; a++) if (m->arm_present[a]) site_pose(m, &k, a, site_pos + 3 * a, site_mat + 9 * a);
}
static void ik_prob_init(IkProb* P, const KModelDesc* m, int arm, double* qpos, const double* q_prev_full,
                         const double* goal_pos, const double* goal_quat) {
  P->m = m; P->arm = arm; P->n = m->arm_nq[arm]; P->qpos = qpos;
  memcpy(P->goal_pos, goal_pos, 24); memcpy(P->goal_quat, goal_quat, 32);
  for (int i = 0; i < P->n; i++) {
    int q = m->arm_q_id[arm][i];
    P->q_prev[i] = q_prev_full[q]; P->q_home[i] = m->q_home[q];
    P->lb[i] = m->jnt_range[q][0]; P->ub[i] = m->jnt_range[q][1];
  }
}
void ko_ik_res(const KModelDesc* m, int arm, double* qpos, const double* x, const double* q_prev_full,
               const double* goal_pos, const double* goal_quat, double* f) {
  IkProb P; ik_prob_init(&P, m, arm, qpos, q_prev_full, goal_pos, goal_quat); ik_fun(&P, x, f);
}
void ko_ik_jac(const KModelDesc* m, int arm, double* qpos, const double* x, const double* q_prev_full,
               const double* goal_pos, const double* goal_quat, double* J) {
  IkProb P; ik_prob_init(&P, m, arm, qpos, q_prev_full, goal_pos, goal_quat); ik_jacobian(&P, x, J);
}
void ko_ik(const KModelDesc* m, int arm, double* qpos, const double* goal_pos, const double* goal_quat,
           double* q_out, int32_t* nfev, int32_t* status) {
  double prev[NQMAX];
  memcpy(prev, qpos, sizeof(double) * (m->nlink + 7));
  int nf, st;
  ik_solve(m, arm, qpos, prev, goal_pos, goal_quat, q_out, &nf, &st);
  *nfev = nf; *status = st;
}
void ko_euler_goal(const double* site_mat, const double* delta, double* quat) { euler_add_to_quat(site_mat, delta, quat); }
/* M (nv x nv dense incl. cube diagonal), bias, qacc_smooth (actuation on), constraint count */
int ko_dynamics(const KModelDesc* m, const double* qpos, const double* qvel, const double* ctrl,
                double* M, double* bias, double* qacc_smooth, double* qacc, int32_t* nefc, double* efc_J,
                double* efc_aref, double* efc_R) {
  int nl = m->nlink, nv = nl + 6;
  Dyn* D = (Dyn*)malloc(sizeof(Dyn));
  int rc = step1(m, qpos, qvel, D, 1);
  if (rc == 0) {
    memset(M, 0, sizeof(double) * nv * nv);
    for (int i = 0; i < nl; i++) for (int j = 0; j < nl; j++) M[i * nv + j] = D->M[i * nl + j];
    for (int c = 0; c < 3; c++) { M[(nl + c) * nv + nl + c] = m->cube_mass; M[(nl + 3 + c) * nv + nl + 3 + c] = m->cube_inertia[c]; }
    memcpy(bias, D->bias, sizeof(double) * nv);
    Dyn* D0 = (Dyn*)malloc(sizeof(Dyn));
    memcpy(D0, D, sizeof(Dyn));
    D0->nefc = 0;
    double w0[NVMAX]; memset(w0, 0, sizeof w0);
    step2_accel(m, D0, ctrl, 1, w0, qacc_smooth);
    free(D0);
    memset(w0, 0, sizeof w0);
    step2_accel(m, D, ctrl, 1, w0, qacc);
    *nefc = D->nefc;
    if (efc_J) for (int i = 0; i < D->nefc; i++) { memcpy(efc_J + (size_t)i * nv, D->J[i], sizeof(double) * nv); efc_aref[i] = D->aref[i]; efc_R[i] = D->Rr[i]; }
  }
  free(D);
  return rc;
}
int ko_nefc_max(void) { return NEFC_MAX; }

/* examples/2_synthetic_data.py:31-37: (cube_pos - eer_site xpos) / norm, for one env */
void ko_scripted_eer_pos(const KModelDesc* m, const double* qpos, double* out3) {
  Kin k;
  kinematics(m, qpos, &k);
  double sp[3], smat[9];
  site_pose(m, &k, 0, sp, smat);
  double d[3] = {qpos[m->nlink] - sp[0], qpos[m->nlink + 1] - sp[1], qpos[m->nlink + 2] - sp[2]};
  double n = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  out3[0] = d[0] / n; out3[1] = d[1] / n; out3[2] = d[2] / n;
}

/* row kinds of the constraint problem ko_dynamics reports: type 0 = friction loss (two-sided, saturating at floss),
 * type 1 = one-sided (joint limits, pyramidal contact edges) */
int ko_constraint_rows(const KModelDesc* m, const double* qpos, const double* qvel, int32_t* type, double* floss) {
  Dyn* D = (Dyn*)malloc(sizeof(Dyn));
  int rc = step1(m, qpos, qvel, D, 1);
  int n = rc == 0 ? D->nefc : -1;
  for (int i = 0; i < n; i++) { type[i] = D->type[i]; floss[i] = D->floss[i]; }
  free(D);
  return n;
}

/* ------------------------------------------------------------------ gripper-camera depth (BASELINE config 5)
 * Camera branch of get_observation (env_sim.py:140-145) restated for a float depth image of the surrogate scene
 * (cube box, table plane, finger spheres).  Camera frame = MuJoCo mj_camlight, mode targetbody: z = (cam - target)
 * normalised, x = (0,0,1) x z, y = z x x; pixel (r, c) looks along x*dx + y*dy - z with the pinhole focal length
 * (H/2)/tan(fovy/2); depth = distance along the optical axis, clipped to [znear, zfar] (no hit = zfar). */
/* shared by the depth and the RGB render: rgb != NULL selects Lambert shading (cube rgba 1 0 0, table .2 .2 .2: scene.xml:15,20;
 * headlight ambient 0.4 + MuJoCo's default headlight diffuse 0.4 at the camera, three directional lights of diffuse 0.3:
 * scene.xml:8-13; no specular / shadows / fog; background black) */
static void render_any(const KModelDesc* m, const double* qpos, int cam, int H, int W, float* out, uint8_t* rgb) {
  Kin k;
  kinematics(m, qpos, &k);
  int cl = m->cam_link[cam], tl = m->cam_target_link[cam];
  double co[3], to[3], z[3], x[3], y[3], up[3] = {0, 0, 1};
  /* a link of -1 = world frame (the fixed top / head cameras and their target, the table body) */
  if (cl < 0) memcpy(co, m->cam_pos[cam], sizeof co);
  else { mat_vec3(co, k.xmat[cl], m->cam_pos[cam]); for (int c = 0; c < 3; c++) co[c] += k.xpos[cl][c]; }
  if (tl < 0) memcpy(to, m->cam_target_pos[cam], sizeof to);
  else { mat_vec3(to, k.xmat[tl], m->cam_target_pos[cam]); for (int c = 0; c < 3; c++) to[c] += k.xpos[tl][c]; }
  for (int c = 0; c < 3; c++) z[c] = co[c] - to[c];
  normalize3(z);
  cross3(x, up, z); normalize3(x);
  cross3(y, z, x); normalize3(y);
  double f = (0.5 * H) / tan(0.5 * m->cam_fovy[cam] * (M_PI / 180.0));
  double sph[KM_MAX_SPHERES][3];
  for (int s = 0; s < m->nsphere; s++) {
    int l = m->sphere_link[s];
    mat_vec3(sph[s], k.xmat[l], m->sphere_pos[s]);
    for (int c = 0; c < 3; c++) sph[s][c] += k.xpos[l][c];
  }
  const double r3 = 0.57735026918962576451, r2 = 0.70710678118654752440;
  const double L[3][3] = {{-r3, -r3, r3}, {r3, -r3, r3}, {0, r2, r2}};
  const double col[4][3] = {{0, 0, 0}, {0.2, 0.2, 0.2}, {1, 0, 0}, {0.647059, 0.647059, 0.647059}};
  for (int r = 0; r < H; r++) for (int c = 0; c < W; c++) {
    double dx = (c + 0.5 - 0.5 * W) / f, dy = -(r + 0.5 - 0.5 * H) / f;
    double d[3] = {x[0] * dx + y[0] * dy - z[0], x[1] * dx + y[1] * dy - z[1], x[2] * dx + y[2] * dy - z[2]};
    double best = m->cam_zfar, nrm[3] = {0, 0, 1};
    int mat = 0;
    if (d[2] != 0) {
      double t = (m->table_z - co[2]) / d[2];
      double hit[3] = {co[0] + t * d[0], co[1] + t * d[1], m->table_z};
      if (t > 0 && t < best && over_table(m, hit)) { best = t; mat = 1; }
    }
    double rel[3] = {co[0] - k.cube_pos[0], co[1] - k.cube_pos[1], co[2] - k.cube_pos[2]}, ol[3], dl[3];
    matT_vec3(ol, k.cube_mat, rel); matT_vec3(dl, k.cube_mat, d);
    double t0 = -INFINITY, t1 = INFINITY, s0 = 0, s1 = 0; int ok = 1, a0 = 0, a1 = 0;
    for (int a = 0; a < 3; a++) {
      double h = m->cube_half[a];
      if (dl[a] != 0) {
        double ta = (-h - ol[a]) / dl[a], tb = (h - ol[a]) / dl[a], sa = -1, sb = 1;
        if (ta > tb) { double s_ = ta; ta = tb; tb = s_; sa = 1; sb = -1; }
        if (ta > t0) { t0 = ta; a0 = a; s0 = sa; }
        if (tb < t1) { t1 = tb; a1 = a; s1 = sb; }
      }
      else if (ol[a] < -h || ol[a] > h) ok = 0;
    }
    if (ok && t0 <= t1 && t1 > 0) {
      int front = t0 > 0; double t = front ? t0 : t1;
      if (t < best) {
        best = t; mat = 2;
        int ax = front ? a0 : a1; double sg = front ? s0 : s1;
        nrm[0] = sg * k.cube_mat[ax]; nrm[1] = sg * k.cube_mat[3 + ax]; nrm[2] = sg * k.cube_mat[6 + ax];
      }
    }
    for (int s = 0; s < m->nsphere; s++) {
      if (!m->sphere_visible[s]) continue;            /* collision-only link spheres are not drawn */
      double oc[3] = {co[0] - sph[s][0], co[1] - sph[s][1], co[2] - sph[s][2]};
      double a = dot3(d, d), b = dot3(d, oc), cc = dot3(oc, oc) - m->sphere_radius[s] * m->sphere_radius[s];
      double disc = b * b - a * cc;
      if (disc >= 0) {
        double t = (-b - sqrt(disc)) / a;
        if (t > 0 && t < best) { best = t; mat = 3; for (int q = 0; q < 3; q++) nrm[q] = (oc[q] + t * d[q]) / m->sphere_radius[s]; }
      }
    }
    if (out) out[r * W + c] = (float)fmin(fmax(best, m->cam_znear), m->cam_zfar);
    if (rgb) {
      double I = 0;
      if (mat) {
        double head = fmax(0.0, -dot3(nrm, d) / sqrt(dot3(d, d)));
        I = 0.4 + 0.4 * head;
        for (int l = 0; l < 3; l++) I += 0.3 * fmax(0.0, dot3(nrm, L[l]));
        I = fmin(I, 1.0);
      }
      for (int q = 0; q < 3; q++) rgb[(r * W + c) * 3 + q] = (uint8_t)(255.0 * col[mat][q] * I + 0.5);
    }
  }
}
void ko_render_depth(const KModelDesc* m, const double* qpos, int cam, int H, int W, float* out) { render_any(m, qpos, cam, H, W, out, NULL); }
void ko_render_rgb(const KModelDesc* m, const double* qpos, int cam, int H, int W, uint8_t* rgb) { render_any(m, qpos, cam, H, W, NULL, rgb); }
