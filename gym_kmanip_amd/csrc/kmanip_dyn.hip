// kmanip_dyn.hip -- one control step (or a chunk of them) of every env in ONE launch (gfx950, wave64).
//
// Replaces, for every env, KManipEnvSim.k_step (reference gym_kmanip/env_sim.py:196-200): KManipTask.before_step
// (env_sim.py:38-108: action decode + IK, device code in kmanip_ik_coop.hpp, fused in here) and then
// `physics.step(n_sub_steps)` as dm_control runs it (control_timestep :210), legacy order
//   mj_step2 (on products of the PRE-IK state) ; (n-1) x mj_step ; mj_step1
// followed by KManipTask.get_reward (env_sim.py:148-179) and get_observation (env_sim.py:110-146), the
// TimeLimit done flag (__init__.py:28,247) and, when enabled, the auto-reset
// (KManipTask.initialize_episode env_sim.py:23-36 + dm_control's mj_forward without actuation).
//
// Execution model ("many envs per wavefront"): a workgroup is ONE wave of 64 lanes holding 64/G envs;
// each env is owned by a group of G lanes (G = 16 for nv = 16, G = 32 for nv = 26), lane d of the group
// owning dof d: its component of qacc, its COLUMN of every contact-basis Jacobian, its ROW of the joint-space
// inertia, of the Newton Hessian and of its Cholesky factor, and its own single-dof constraint rows all live in
// that lane's registers.  The cooperative linear algebra exchanges values with DPP row broadcasts folded into the
// FMA (v_fmac_f64_dpp row_newbcast; v_permlane16_swap copies for the two-row groups) -- no LDS, no synchronisation.
// Per-env data that needs lane-indexed random access (body frames, M^-1, contact geometry and records) sits in LDS:
// 7 KB per env for Solo, 15 KB for Dual/Torso in the Newton variant.  The kernel needs 400-500 registers, i.e. one
// wave per SIMD: 16 Solo envs per CU, all 4096 envs of the headline config resident in a single round.  HBM is
// touched once on entry and once on exit with struct-of-arrays coalesced columns.  Per-env reductions are DPP row
// reductions (row_mirror / row_half_mirror / quad_perm, v_permlane16_swap across the rows of a 32-lane group) whose
// result is bitwise identical on every lane (they opt out of FMA contraction for that reason).  A lane group never
// needs s_barrier: all its lanes sit in one wave.
//
// Formulations deliberately differ from the oracle's (so parity is a cross-check, not a re-run):
//   mass matrix      : composite inertias about the world origin, one column per lane (oracle: link-frame CRBA)
//   bias forces      : per-body bias wrenches projected with J^T   (oracle: RNE backward recursion)
//   M^-1             : explicit inverse, Gauss-Jordan on register rows (oracle: Cholesky factor + solves)
//   Newton           : incremental state, Hessian rows / Cholesky / solves in registers (oracle: dense, recomputed per iteration)
//   constraint rows  : single-dof rows + 4-vector contact bases, pyramid edges expanded on the fly
//   PGS              : per-contact block form on the 4x4 Gram matrix (algebraically the same row order)
#include "kmanip_ik_coop.hpp"
#include <stdlib.h>
// tree loops over link candidates (static addresses + a mask bit each): fully unrolled for the 10-link model, where all the
// loads can be in flight together; the 20-link models sit at the 512-register limit and keep them rolled
#define KM_TREE_UNROLL(NL) NL <= 10 ? NL : 1

// Work units one Newton iteration of each kind adds to Ws::work (roughly kilo-clocks on the two-arm kernels; only their ORDER
// matters: k_sort_envs ranks the envs by them).  Two-arm kernels only: the single-arm headline launch is one residency round.
// (measured on the single-arm kernel, tests/tools/wave_times.py with -DKM_WORK_COUNTERS_ALL: wave cycles = 375 k + 308 x the
// slowest env's work units + 13.8 k x the largest IK evaluation count, R^2 0.66; the counters cost it 1.5 %, so it ships without)
#ifdef KM_WORK_COUNTERS_ALL
#define KM_WORK_COUNTERS(NL) true
#else
#define KM_WORK_COUNTERS(NL) ((NL) > 10)
#endif
#define KM_WORK_ALL 40
#define KM_WORK_ARM 14
#define KM_WORK_PLAIN 3
#define KM_WORK_CUBE 5
template <int NL> struct Dim {
  static constexpr int NV = NL + 6;
  static constexpr int NQ = NL + 7;
  static constexpr int NS = 2 * NL;               // arm single-dof rows: friction loss (<= nl) + limits (<= nl)
  static constexpr int NSPH = 6 * (NL / 10);      // collision-sphere CANDIDATES, one lane each: per arm two fingers, palm, three joint housings
  static constexpr int NSS = KM_SPHERE_SLOTS(NL); // sphere contacts KEPT per kind and sub-step (the first penetrating ones in sphere order)
  static constexpr int NST = KM_SPHERE_TABLE_SLOTS(NL);   // ... of the sphere-table kind
  static constexpr int NC = 4 + NSS + NST;        // contact SLOTS: 4 cube-table corners, NSS sphere-cube, NST sphere-table
  static constexpr int NCF = NSS + NST;           // slots that involve arm dofs
};
// compile-time kind of contact slot c: 0 = table(plane) - cube corner, 1 = sphere - cube, 2 = table - sphere.  WHICH sphere sits
// in a sphere slot is decided per sub-step by collide_parallel (Ws::slot_sph).
template <int NL> __device__ __forceinline__ constexpr int slot_kind(int c) { return c < 4 ? 0 : (c < 4 + Dim<NL>::NSS ? 1 : 2); }

// Per-link model constants staged in LDS once per workgroup (lane-indexed reads stay on-chip); scalars
// and small fixed arrays are read straight from the global KModelDesc with wave-uniform (scalar) loads.
template <int NL>
struct alignas(16) LModel {
  int parent[NL], jtype[NL], forcelimited[NL];
  uint32_t anc[NL], desc[NL];
  int jump[4][NL], fk_rounds, split;
  real pos[NL][3], quat[NL][4], jaxis[NL][3], range[NL][2], floss[NL], kp[NL], ctrlrange[NL][2], forcerange[NL][2];
  real mass[NL], com[NL][3], inertia[NL][3], q_home[NL];
  real R[NL][9];        // constant rotation of each link in its parent (from link_quat)
  // soft-constraint constants of the two parameter sets (0 = default pairs / joint rows, 1 = pairs with the cube):
  // stiffness k, damping b (mj_makeImpedance / solref), impedance at zero distance
  real kb[2][2], imp0[2];
  // MuJoCo's qpos0-time constants (mj_setConst): efc_diagApprox of this link's single-dof rows (dof_invweight0), of the
  // FIRST pyramid edge of every contact pair (tran + mu^2 tran, tran = summed body_invweight0 of the pair: cornerA for a
  // cube corner on the table, sphA[0][s] for sphere s on the cube, sphA[1][s] for sphere s on the table), of the cube's
  // friction-loss rows (linear, angular), and the solvers' termination scale 1 / (meaninertia * nv)
  real dofw[NL], sphA[2][Dim<NL>::NSPH], cornerA, cubew[2], scale;
  // solimp of the two parameter sets, clamped like mj_makeImpedance clamps it, with the reciprocals the spline divides by
  // (mode 0: constant (d0 + dw) / 2; 1: linear; 2: MuJoCo's default quadratic spline)
  struct Imp { real d0, dw, iw, mid, imid, i1mid; int mode; } imp[2];
  // collision candidates (round 6): link, centre, radius and capsule segment of sphere s were per-lane GLOBAL loads in every
  // sub-step's narrow phase (and the link again, behind an LDS load, for every active sphere slot of the constraint assembly)
  real fric[2][2];      // (tangential, torsional) friction of pairs without / with the cube (con_def_friction, con_cube_friction)
  int sph_link[Dim<NL>::NSPH > 0 ? Dim<NL>::NSPH : 1], nsph;
  real sph_pos[Dim<NL>::NSPH > 0 ? Dim<NL>::NSPH : 1][3], sph_rad[Dim<NL>::NSPH > 0 ? Dim<NL>::NSPH : 1], sph_seg[Dim<NL>::NSPH > 0 ? Dim<NL>::NSPH : 1][3];
};

// friction coefficient k (0, 1: tangential, 2: torsional) of contact slot kind `kind`: pairs with the cube use the mixed cube
// parameters, finger-table pairs MuJoCo's defaults -- wave-uniform model scalars, not worth a slot in the LDS records
__device__ __forceinline__ real slot_mu(const KModelDesc* m, int kind, int k) {
  const real* fr = kind != 2 ? m->con_cube_friction : m->con_def_friction;
  return k < 2 ? fr[0] : fr[1];
}
// Solver view of one pyramidal contact (group-uniform scalars).  Basis index 0 = normal, 1..2 = tangents,
// 3 = torsion.  Edge e = 2*(k-1) + s uses J_0 + sm J_k with sm = (s ? -mu[k-1] : mu[k-1]).
struct ConRec {
  real mu[3];
  real R;          // regulariser shared by all edges (MuJoCo pyramidal rule)
  real D;          // 1 / R
  real inv[6];     // 1 / (A_ee + R); 0 for the unused edges of a condim-3 pair
  real den[6];     // A_ee + R
  real aref[6];    // reference acceleration of the edge
  real f[6];       // edge forces
};

#define KM_WS_PAD(NL) ((NL) <= 10 ? 9 : 1)      // doubles of padding at the end of Ws (see the note on row strides in it)
template <int NL>
struct Ws {
  static constexpr int NV = Dim<NL>::NV, NQ = Dim<NL>::NQ, NS = Dim<NL>::NS, NC = Dim<NL>::NC;
  real qpos[NQ], qvel[NV], ctrl[NL], warm[NV], qpos_ik[NL];
  union {
    // kinematics: live from fk() to the end of the contact-Jacobian build ...
    struct { real xpos[NL][3], xmat[NL][9], axis[NL][3], cpos[NL][3], cube_mat[9]; } k;
#if KM_VAR_SOLVER == KM_SOLVER_PGS
    // ... then the same bytes hold the per-edge Gram rows Ge[c][e][l] = J_l . M^-1 (J_0 + sm J_k)^T for PGS
    struct { real Ge[NC][6][4]; } p;
#endif
  };
  // Row strides of everything a lane reads or writes at [its index][k] are ODD numbers of doubles (round 5): ds_write_b64 banks
  // are (a / 4) mod 32 inside each 16-lane group and ds_read_b64 banks (a / 4) mod 64 inside each 32-lane half, so a stride of 10
  // (or 6) doubles puts lanes i and i + 8 of an env on one bank; and sizeof(Ws<10>) is 128 mod 256 bytes, which puts the two envs
  // of a 32-lane half on opposite halves of the bank row for every odd-stride and unit-stride access (KM_WS_PAD below).
  real Minv[NL][NL | 1];   // joint-space inertia, overwritten by its inverse
  union {
    struct { union { real bsc[NL][9]; real comp[NL][11]; }; real FN[NL][7]; } f;   // bias-pass scratch | composite inertias (10 used); bias wrenches (6 used)
#if KM_VAR_SOLVER == KM_SOLVER_PGS
    real stage[4][NV];                               // staging of basis rows for B = M^-1 J^T
    ConRec rec[NC];                                  // solver records (built last; Newton keeps a slot's constants in its lane)
#endif
  };
  real bias[NV], tmp[NV];
#if KM_VAR_SOLVER == KM_SOLVER_PGS
  real as[NV], tmp2[NV], tmp3[NV];
#endif
  int ns, bad, touch_ct;
  int work;                // Newton iterations of this control step, weighted by kind (KM_WORK_*): the cost predictor of k_sort_envs
  uint32_t contact_mask;   // KM_CON_* bits (which candidate pairs touch)
  uint32_t cact;           // active contact slots
  // single-dof constraint rows on ARM dofs (friction loss, then limits); the cube's friction-loss rows are
  // lane-local registers
#if KM_VAR_SOLVER == KM_SOLVER_PGS
  int s_dof[NS], s_type[NS], s_quad[NS];
  real s_sign[NS], s_pos[NS], s_f[NS], s_R[NS], s_aref[NS], s_den[NS], s_inv[NS], s_floss[NS];
#endif
#if KM_VAR_SOLVER == 1      // (Newton; the enum constants are not visible to the preprocessor)
  // the Cholesky factor of a one-row Newton system on its way from row-per-lane to column-per-lane (rows padded to an odd
  // number of doubles: the lanes' row writes then fall into different banks)
  real LT[NL <= 10 ? NL + 6 : 1][(NL <= 10 ? NL + 6 : 1) + 1];     // (one-row groups only)
#endif
  // contact geometry per slot
  real c_pos[NC][3], c_frame[NC][9], c_dist[NC];
  int slot_sph[NC];        // sphere index held by each active sphere slot (4..NC-1)
  uint32_t slot_anc[NC];   // ... and the ancestor mask of that sphere's link (round 6: the constraint assembly read it through two more dependent loads)
  real pad_[KM_WS_PAD(NL)];
};
static_assert(KM_VAR_NL != 10 || KM_VAR_SOLVER != 1 || sizeof(Ws<KM_VAR_NL>) % 256 == 128, "Ws<10>: consecutive envs 128 bytes apart modulo the 256-byte bank row");

// One-row groups (round 3): lane c < NC OWNS contact slot c for the Newton solve -- its regulariser, friction coefficients and
// reference offsets live in that lane's registers, the slot's pyramid edges are evaluated there (all slots at once, one per
// lane), and what the other lanes need (four force components, seven Hessian weights) reaches them as row broadcasts folded
// into their FMAs.  Round 2 spread the EDGES over the lanes and exchanged projections and forces through LDS records /
// per-slot broadcasts, every lane redoing each slot's scalar arithmetic.
// x_e = (u_0 - A_0) +- mu_k (u_k - A_k), u = J_c a: A_0 = -b v_0 - k imp dist, A_k = -b v_k (v = J_c qvel).
struct SlotC {
  real D, D3;        // 1 / R of the slot's edges and of its torsion pair (0: condim-3 pair / inactive slot / lane owns no slot)
  real mu, mu3;      // tangential / torsional friction coefficient
  real A[4];
};

// this lane's column of every contact basis: J (jb) and M^-1 J^T (bb); compile-time indexed only
// (bb only for the slots that involve arm dofs: for table-cube slots M^-1 is diagonal, bb = jb * invm)
// Newton path only: mrow = this lane's row of the arm inertia M; the lane's OWN single-dof constraint rows
// (dof `sub`: friction loss and, when violated, its joint limit) -- no row tables in LDS.
template <int NL> struct CReg {
  real jb[Dim<NL>::NC][4];
  real bb[Dim<NL>::NCF][4];
  real mrow[NL];
  real fl, Rf, Df, areff;    // friction-loss row x = a - areff          (fl = 0: no row); Df = 1 / Rf
  real sg, Rl, Dl, arefl;    // limit row         x = sg * a - arefl     (sg = 0: no row); Dl = 1 / Rl
  SlotC sc;                  // one-row groups: the contact slot this lane owns
  // Newton path (round 6): the cube's rotation and this lane's corner contact point relative to the cube centre, fetched once per
  // sub-step by the constraint assembly -- every projection J_c v of the table-cube slots (two per start evaluation, one per Newton
  // iteration) re-read all twelve from LDS
  real cm[9], pr[3];
};

#ifdef KM_PROFILE
#if KM_VAR_NL == 10 && KM_VAR_SOLVER == 1
extern "C" int kmanip_dbg_prof(unsigned long long* out, int reset) {
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), sizeof(unsigned long long) * KM_NPH) != hipSuccess) return -1;
  if (reset) { unsigned long long z[KM_NPH] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof z) != hipSuccess) return -1; }
  return 0;
}
extern "C" int kmanip_dbg_prof_blocks(unsigned long long* out, int nblocks) {
  if (nblocks > KM_PROF_BLOCKS) return -1;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof_blk), sizeof(unsigned long long) * KM_NPH * 4 * nblocks) == hipSuccess ? 0 : -1;
}
#endif
#if KM_VAR_NL == 20 && KM_VAR_SOLVER == 1
extern "C" int kmanip_dbg_prof20(unsigned long long* out, int reset) {          // the DualArm / Torso Newton object's accumulators
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), sizeof(unsigned long long) * KM_NPH) != hipSuccess) return -1;
  if (reset) { unsigned long long z[KM_NPH] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof z) != hipSuccess) return -1; }
  return 0;
}
#endif
#endif
#define GSYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

// sum over the G lanes of a group, result identical (bitwise) in every lane.  16-lane rows use four DPP
// steps (row_mirror, row_half_mirror, two quad_perms) instead of ds_bpermute; G = 32 adds one swizzle.
// The additions below must NOT be contracted with a multiply in the caller's argument (gsum(x * y)): lane i would add
// the exact product to its partner's ROUNDED one and the lanes of a group would no longer hold the bitwise-identical
// sum -- which group-uniform control flow (line-search breaks, termination tests) relies on.  Contraction needs the
// `contract` flag on both operations, so switching it off for this body is enough.
template <int G> __device__ __forceinline__ real gsum(real v) {
#pragma clang fp contract(off)
  static_assert(G == 16 || G == 32, "lane group must be one or two DPP rows");
  v += dpp_f64<0x140>(v);   // row_mirror:      i <-> 15 - i
  v += dpp_f64<0x141>(v);   // row_half_mirror: i <-> 7 - i within each half row
  v += dpp_f64<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_f64<0x4E>(v);    // quad_perm [2,3,0,1]
  if (G == 32) {
    const BSrc<32> r = bsrc<32>(v);                  // even-row sum and odd-row sum, each in both rows (v_permlane16_swap)
    v = r.e + r.o;                                   // same operands in the same order on every lane
  }
  return v;
}
// N group sums at once, step by step: the SAME operations per value as N gsum calls (bitwise the same results), but the chains
// interleave -- with one wave per SIMD a lone chain waits out every add's latency and the two wait states in front of each DPP read
template <int G, int N> __device__ __forceinline__ void gsum_n(real (&v)[N]) {
#pragma clang fp contract(off)
  static_assert(G == 16 || G == 32, "lane group must be one or two DPP rows");
  real t[N];
#pragma unroll
  for (int i = 0; i < N; i++) t[i] = dpp_f64<0x140>(v[i]);
#pragma unroll
  for (int i = 0; i < N; i++) v[i] += t[i];
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
  if (G == 32) {
#pragma unroll
    for (int i = 0; i < N; i++) { const BSrc<32> r = bsrc<32>(v[i]); v[i] = r.e + r.o; }
  }
}
// OR over the G lanes of a group, every lane receiving it: the same four DPP steps as gsum (round 3; round 2 went through
// four or five dependent ds_bpermute round trips -- 12 of them per sub-step for the contact masks and the divergence flag)
template <int CTRL> __device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
template <int G> __device__ __forceinline__ int gor(int v) {
  static_assert(G == 16 || G == 32, "lane group must be one or two DPP rows");
  v |= dpp_i32<0x140>(v);   // row_mirror
  v |= dpp_i32<0x141>(v);   // row_half_mirror
  v |= dpp_i32<0xB1>(v);    // quad_perm [1,0,3,2]
  v |= dpp_i32<0x4E>(v);    // quad_perm [2,3,0,1]
  if constexpr (G == 32) {
    const auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
    v = (int)(r[0] | r[1]);
  }
  return v;
}

// mj_kinematics, all links at once: lane i builds link i's transform in its parent (constant rotation times the
// planar joint rotation; one sincos per lane instead of NL in a row), then ceil(log2(depth)) rounds of pointer
// jumping compose it with the transform of the 2^k-th ancestor (staged in the link's own xmat/xpos slots).
// kin (optional, 15 doubles): the world frame of THIS lane's link as it was written to LDS -- rotation R[9], origin p[3], centre of
// mass c[3] (zeros on lanes without a link) -- so that the passes that follow need not read their own link back (round 6)
template <int NL, int G>
__device__ __forceinline__ void fk_parallel(Ws<NL>& w, const LModel<NL>& lm, int sub, real* kin = nullptr) {
  real R[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, p[3] = {0, 0, 0};
  const bool on = sub < NL;
  // (round 6) everything the pass reads about this lane's link -- its coordinate, its constant frame in the parent, its joint
  // type, its jump table, its centre of mass -- in one batch: as written, each sat behind the branch that used it
  const int li = on ? sub : 0;
  real q = w.qpos[li], lp[3] = {lm.pos[li][0], lm.pos[li][1], lm.pos[li][2]}, lR[9], cl[3] = {lm.com[li][0], lm.com[li][1], lm.com[li][2]};
#pragma unroll
  for (int c = 0; c < 9; c++) lR[c] = lm.R[li][c];
  int jt = lm.jtype[li], jmp0 = lm.jump[0][li], jmp1 = lm.jump[1][li], jmp2 = lm.jump[2][li], jmp3 = lm.jump[3][li], rounds = lm.fk_rounds;
  km_pin(q); km_pin(lp, cl); km_pin(lR); km_pin_i(jt, rounds); km_pin_i(jmp0, jmp1); km_pin_i(jmp2, jmp3);
  if (on) {
    p[0] = lp[0]; p[1] = lp[1]; p[2] = lp[2];
    if (jt == KM_JNT_SLIDE) {
#pragma unroll
      for (int c = 0; c < 9; c++) R[c] = lR[c];
      p[0] += R[2] * q; p[1] += R[5] * q; p[2] += R[8] * q;
    } else {
      real sn, cs;
      km_sincos(q, &sn, &cs);
#pragma unroll
      for (int a = 0; a < 3; a++) {
        const real c0 = lR[3 * a], c1 = lR[3 * a + 1];
        R[3 * a] = cs * c0 + sn * c1;
        R[3 * a + 1] = cs * c1 - sn * c0;
        R[3 * a + 2] = lR[3 * a + 2];
      }
    }
#pragma unroll
    for (int c = 0; c < 9; c++) w.k.xmat[sub][c] = R[c];
    w.k.xpos[sub][0] = p[0]; w.k.xpos[sub][1] = p[1]; w.k.xpos[sub][2] = p[2];
  } else if (sub == NL) {
    real cq[4] = {w.qpos[NL + 3], w.qpos[NL + 4], w.qpos[NL + 5], w.qpos[NL + 6]}, cm[9];
    normalize4_fast(cq);
    quat2mat(cm, cq);
#pragma unroll
    for (int c = 0; c < 9; c++) w.k.cube_mat[c] = cm[c];
  }
  GSYNC();
  for (int k = 0; k < rounds; k++) {
    const int jk = k == 0 ? jmp0 : (k == 1 ? jmp1 : (k == 2 ? jmp2 : jmp3));
    const int a = on ? jk : -1;
    if (a >= 0) {
      real A[9], pa[3], Rn[9], t[3];
#pragma unroll
      for (int c = 0; c < 9; c++) A[c] = w.k.xmat[a][c];
      pa[0] = w.k.xpos[a][0]; pa[1] = w.k.xpos[a][1]; pa[2] = w.k.xpos[a][2];
      mat_vec3(t, A, p);
      p[0] = t[0] + pa[0]; p[1] = t[1] + pa[1]; p[2] = t[2] + pa[2];
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) Rn[3 * i + j] = A[3 * i] * R[j] + A[3 * i + 1] * R[3 + j] + A[3 * i + 2] * R[6 + j];
#pragma unroll
      for (int c = 0; c < 9; c++) R[c] = Rn[c];
    }
    GSYNC();
    if (a >= 0) {
#pragma unroll
      for (int c = 0; c < 9; c++) w.k.xmat[sub][c] = R[c];
      w.k.xpos[sub][0] = p[0]; w.k.xpos[sub][1] = p[1]; w.k.xpos[sub][2] = p[2];
    }
    GSYNC();
  }
  real cpo[3] = {0, 0, 0};
  if (on) {
    real cw[3];
    mat_vec3(cw, R, cl);
    cpo[0] = p[0] + cw[0]; cpo[1] = p[1] + cw[1]; cpo[2] = p[2] + cw[2];
    w.k.cpos[sub][0] = cpo[0]; w.k.cpos[sub][1] = cpo[1]; w.k.cpos[sub][2] = cpo[2];
    w.k.axis[sub][0] = R[2]; w.k.axis[sub][1] = R[5]; w.k.axis[sub][2] = R[8];
  }
  if (kin) {
#pragma unroll
    for (int c = 0; c < 9; c++) kin[c] = on ? R[c] : 0.0;
#pragma unroll
    for (int c = 0; c < 3; c++) { kin[9 + c] = on ? p[c] : 0.0; kin[12 + c] = cpo[c]; }
  }
  GSYNC();
}

// column j of the com Jacobian of body b (world frame): linear part jv, angular part jw
template <int NL>
__device__ __forceinline__ void com_jac_col(const Ws<NL>& w, const LModel<NL>& lm, int b, int j, real* jv, real* jw) {
  if (lm.jtype[j] == KM_JNT_SLIDE) {
    jv[0] = w.k.axis[j][0]; jv[1] = w.k.axis[j][1]; jv[2] = w.k.axis[j][2];
    jw[0] = 0; jw[1] = 0; jw[2] = 0;
  } else {
    real r[3] = {w.k.cpos[b][0] - w.k.xpos[j][0], w.k.cpos[b][1] - w.k.xpos[j][1], w.k.cpos[b][2] - w.k.xpos[j][2]};
    real ax[3] = {w.k.axis[j][0], w.k.axis[j][1], w.k.axis[j][2]};
    cross3(jv, ax, r);
    jw[0] = ax[0]; jw[1] = ax[1]; jw[2] = ax[2];
  }
}

// Composite-rigid-body mass matrix.  Lane b first writes body b's own (mass, first moment m*c, inertia about
// the world origin) -- 10 numbers; lane 0 then suffix-accumulates them up the tree (children into parents);
// lane j finally projects the unit-acceleration wrench of its composite onto every ancestor joint:
//   F = mc*a_O + alpha x h,  N_O = Io*alpha + h x a_O   (hinge: alpha = axis_j, a_O = o_j x axis_j; slide: a_O = axis_j)
//   M_ij = axis_i . (N_O - o_i x F)  (hinge i)   |   axis_i . F  (slide i)
template <int NL, int G>
__device__ __forceinline__ void composite_own(Ws<NL>& w, const LModel<NL>& lm, int sub) {
  for (int b = sub; b < NL; b += G) {
    const real mb = lm.mass[b];
    const real c[3] = {w.k.cpos[b][0], w.k.cpos[b][1], w.k.cpos[b][2]};
    const real* R = w.k.xmat[b];
    const real I0 = lm.inertia[b][0], I1 = lm.inertia[b][1], I2 = lm.inertia[b][2];
    const real cc = dot3(c, c);
    real* o = w.f.comp[b];
    o[0] = mb; o[1] = mb * c[0]; o[2] = mb * c[1]; o[3] = mb * c[2];
    // R diag(I) R^T + m (|c|^2 1 - c c^T), packed xx xy xz yy yz zz
    o[4] = R[0] * R[0] * I0 + R[1] * R[1] * I1 + R[2] * R[2] * I2 + mb * (cc - c[0] * c[0]);
    o[5] = R[0] * R[3] * I0 + R[1] * R[4] * I1 + R[2] * R[5] * I2 - mb * c[0] * c[1];
    o[6] = R[0] * R[6] * I0 + R[1] * R[7] * I1 + R[2] * R[8] * I2 - mb * c[0] * c[2];
    o[7] = R[3] * R[3] * I0 + R[4] * R[4] * I1 + R[5] * R[5] * I2 + mb * (cc - c[1] * c[1]);
    o[8] = R[3] * R[6] * I0 + R[4] * R[7] * I1 + R[5] * R[8] * I2 - mb * c[1] * c[2];
    o[9] = R[6] * R[6] * I0 + R[7] * R[7] * I1 + R[8] * R[8] * I2 + mb * (cc - c[2] * c[2]);
  }
}
// subtree sums, one link per lane: comp/FN of link i += those of its proper descendants (read-all, sync, write)
template <int NL, int G>
__device__ __forceinline__ void composite_accumulate(Ws<NL>& w, const LModel<NL>& lm, int sub) {
  real acc[16];
  const bool on = sub < NL;
  if (on) {
    if constexpr (NL <= 10) {
    // every candidate j at a compile-time address (all loads can be in flight together; no mask-driven pointer chase),
    // taken or not by its descendant bit.  Links are ordered parents-first, so descendants have larger indices.
    const uint32_t dm = lm.desc[sub];
#pragma unroll
    for (int k = 0; k < 16; k++) acc[k] = 0;
#pragma unroll
    for (int j = 0; j < NL; j++) {
      const bool take = (dm >> j) & 1u;                 // (bit `sub` itself is set: the link's own contribution)
#pragma unroll
      for (int k = 0; k < 10; k++) { const real v = w.f.comp[j][k]; acc[k] += take ? v : 0.0; }
#pragma unroll
      for (int k = 0; k < 6; k++) { const real v = w.f.FN[j][k]; acc[10 + k] += take ? v : 0.0; }
    }
    } else {                                            // (the 20-link kernels sit at the 512-register limit: rolled mask walk)
#pragma unroll
      for (int k = 0; k < 10; k++) acc[k] = w.f.comp[sub][k];
#pragma unroll
      for (int k = 0; k < 6; k++) acc[10 + k] = w.f.FN[sub][k];
      for (uint32_t mk = lm.desc[sub] & ~(1u << sub); mk; mk &= mk - 1) {
        const int j = __ffs(mk) - 1;
#pragma unroll
        for (int k = 0; k < 10; k++) acc[k] += w.f.comp[j][k];
#pragma unroll
        for (int k = 0; k < 6; k++) acc[10 + k] += w.f.FN[j][k];
      }
    }
  }
  GSYNC();
  if (on) {
#pragma unroll
    for (int k = 0; k < 10; k++) w.f.comp[sub][k] = acc[k];
#pragma unroll
    for (int k = 0; k < 6; k++) w.f.FN[sub][k] = acc[10 + k];
  }
}
template <int NL, int G>
__device__ __forceinline__ void mass_matrix(Ws<NL>& w, const LModel<NL>& lm, int sub) {
  for (int j = sub; j < NL; j += G) {
    const real* o = w.f.comp[j];
    const real ax[3] = {w.k.axis[j][0], w.k.axis[j][1], w.k.axis[j][2]};
    const real oj[3] = {w.k.xpos[j][0], w.k.xpos[j][1], w.k.xpos[j][2]};
    const real h[3] = {o[1], o[2], o[3]};
    real F[3], N[3], t[3];
    if (lm.jtype[j] == KM_JNT_SLIDE) {
      F[0] = o[0] * ax[0]; F[1] = o[0] * ax[1]; F[2] = o[0] * ax[2];
      cross3(N, h, ax);
    } else {
      real aO[3];
      cross3(aO, oj, ax);
      cross3(t, ax, h);
      F[0] = o[0] * aO[0] + t[0]; F[1] = o[0] * aO[1] + t[1]; F[2] = o[0] * aO[2] + t[2];
      N[0] = o[4] * ax[0] + o[5] * ax[1] + o[6] * ax[2];
      N[1] = o[5] * ax[0] + o[7] * ax[1] + o[8] * ax[2];
      N[2] = o[6] * ax[0] + o[8] * ax[1] + o[9] * ax[2];
      cross3(t, h, aO);
      N[0] += t[0]; N[1] += t[1]; N[2] += t[2];
    }
    // rows i = ancestors of j (incl. j), every candidate i at a compile-time address and taken by its ancestor bit -- no
    // pointer chase up the tree through LDS; non-ancestors get the zero they need (the reader mirrors the triangle)
    const uint32_t am = lm.anc[j];
#pragma unroll KM_TREE_UNROLL(NL)
    for (int i = 0; i < NL; i++) {
      const real ai[3] = {w.k.axis[i][0], w.k.axis[i][1], w.k.axis[i][2]};
      const real oi[3] = {w.k.xpos[i][0], w.k.xpos[i][1], w.k.xpos[i][2]};
      cross3(t, oi, F);
      const real mo[3] = {N[0] - t[0], N[1] - t[1], N[2] - t[2]};
      const real val = lm.jtype[i] == KM_JNT_SLIDE ? dot3(ai, F) : dot3(ai, mo);
      w.Minv[i][j] = ((am >> i) & 1u) ? val : 0.0;
    }
  }
}
// One-row groups (NL <= 10, G = 16): composite inertias and subtree wrenches WITHOUT the LDS round trips.  Lane b builds link
// b's own ten composite numbers in registers, takes its bias wrench, and every lane sums over its descendants with
// broadcast-FMAs (acc_k += bcast_j(own_k) * [j in subtree(sub)], runs of four behind one pair of wait states): 160 LDS reads,
// 32 LDS writes and two synchronisations become 40 four-instruction runs.  Then the lane projects ITS composite's unit-
// acceleration wrench onto its ancestors' joints (column `sub` of M, rows through LDS for the row-per-lane inversion) and its
// subtree wrench onto its own joint (bias).
template <int NL, int W>
__device__ __forceinline__ void composite_mass_bias_rows(Ws<NL>& w, const LModel<NL>& lm, int li, int base, const real (&FN)[6], const real* kin = nullptr) {
  static_assert(W <= 16, "one DPP row per block");
  const bool on = li >= 0;
  const int b = on ? li : 0;
  // (round 6) the link's frame from fk_parallel's registers (kin; else one batch from LDS), its constants in one batch
  real Rk[9], oj[3], c[3];
  real massb = lm.mass[b], I0 = lm.inertia[b][0], I1 = lm.inertia[b][1], I2 = lm.inertia[b][2];
  int jtb = lm.jtype[b];
  uint32_t descb = lm.desc[b], ancb = lm.anc[b];
  if (kin) {
#pragma unroll
    for (int k = 0; k < 9; k++) Rk[k] = kin[k];
#pragma unroll
    for (int k = 0; k < 3; k++) { oj[k] = kin[9 + k]; c[k] = kin[12 + k]; }
  } else {
#pragma unroll
    for (int k = 0; k < 9; k++) Rk[k] = w.k.xmat[b][k];
#pragma unroll
    for (int k = 0; k < 3; k++) { oj[k] = w.k.xpos[b][k]; c[k] = w.k.cpos[b][k]; }
    km_pin(Rk); km_pin(oj, c);
  }
  km_pin(massb, I0, I1, I2); km_pin_i(jtb); asm volatile("" : "+v"(descb), "+v"(ancb));
  real own[16];
  {
    const real mb = on ? massb : 0.0;
    const real* R = Rk;
    const real cc = dot3(c, c);
    own[0] = mb; own[1] = mb * c[0]; own[2] = mb * c[1]; own[3] = mb * c[2];
    // R diag(I) R^T + m (|c|^2 1 - c c^T), packed xx xy xz yy yz zz
    own[4] = R[0] * R[0] * I0 + R[1] * R[1] * I1 + R[2] * R[2] * I2 + mb * (cc - c[0] * c[0]);
    own[5] = R[0] * R[3] * I0 + R[1] * R[4] * I1 + R[2] * R[5] * I2 - mb * c[0] * c[1];
    own[6] = R[0] * R[6] * I0 + R[1] * R[7] * I1 + R[2] * R[8] * I2 - mb * c[0] * c[2];
    own[7] = R[3] * R[3] * I0 + R[4] * R[4] * I1 + R[5] * R[5] * I2 + mb * (cc - c[1] * c[1]);
    own[8] = R[3] * R[6] * I0 + R[4] * R[7] * I1 + R[5] * R[8] * I2 - mb * c[1] * c[2];
    own[9] = R[6] * R[6] * I0 + R[7] * R[7] * I1 + R[8] * R[8] * I2 + mb * (cc - c[2] * c[2]);
#pragma unroll
    for (int k = 0; k < 6; k++) own[10 + k] = FN[k];
    if (!on) {
#pragma unroll
      for (int k = 0; k < 16; k++) own[k] = 0;
    }
  }
  real acc[16];
#pragma unroll
  for (int k = 0; k < 16; k++) acc[k] = 0;
  const uint32_t dm = on ? descb >> base : 0u;   // row-local bits (the link's own bit is set: its own contribution)
  static_for<0, W>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    const real take = ((dm >> j) & 1u) ? 1.0 : 0.0;
    constexpr bool WT = j == 0;                  // (the sources own[] are read again for every j: only the first pass can trail their writes)
    dppfma4<false, j, j, j, j, WT>(acc[0], own[0], take, acc[1], own[1], take, acc[2], own[2], take, acc[3], own[3], take);
    dppfma4<false, j, j, j, j, WT>(acc[4], own[4], take, acc[5], own[5], take, acc[6], own[6], take, acc[7], own[7], take);
    dppfma4<false, j, j, j, j, WT>(acc[8], own[8], take, acc[9], own[9], take, acc[10], own[10], take, acc[11], own[11], take);
    dppfma4<false, j, j, j, j, WT>(acc[12], own[12], take, acc[13], own[13], take, acc[14], own[14], take, acc[15], own[15], take);
  });
  if (on) {
    const int j = li;
    const real* o = acc;
    const real ax[3] = {Rk[2], Rk[5], Rk[8]};
    const real h[3] = {o[1], o[2], o[3]};
    real F[3], N[3], t[3];
    const bool slide = jtb == KM_JNT_SLIDE;
    if (slide) {
      F[0] = o[0] * ax[0]; F[1] = o[0] * ax[1]; F[2] = o[0] * ax[2];
      cross3(N, h, ax);
    } else {
      real aO[3];
      cross3(aO, oj, ax);
      cross3(t, ax, h);
      F[0] = o[0] * aO[0] + t[0]; F[1] = o[0] * aO[1] + t[1]; F[2] = o[0] * aO[2] + t[2];
      N[0] = o[4] * ax[0] + o[5] * ax[1] + o[6] * ax[2];
      N[1] = o[5] * ax[0] + o[7] * ax[1] + o[8] * ax[2];
      N[2] = o[6] * ax[0] + o[8] * ax[1] + o[9] * ax[2];
      cross3(t, h, aO);
      N[0] += t[0]; N[1] += t[1]; N[2] += t[2];
    }
    const uint32_t am = ancb;
    // Round 6: one basic block.  With the stores inside `if (i <= j)` the compiler sank each row's six LDS loads into that row's
    // conditional block: ten load -> wait -> compute -> store round trips in a row (one wave per SIMD: nothing hides them).  Now every
    // row's entry is stored unconditionally -- rows this lane does not own go to a scratch slot of its own (w.tmp[j], not live before
    // the solve) -- so nothing is conditional, and the scheduler issues the rows' loads together.  Same operations, same bits.
    real mcol[W];
#pragma unroll
    for (int c = 0; c < W; c++) {
      const int i = base + c < NL ? base + c : NL - 1;      // rows of the block only: M has no entries between blocks (clamped: never stored)
      const real ai[3] = {w.k.axis[i][0], w.k.axis[i][1], w.k.axis[i][2]};
      const real oi[3] = {w.k.xpos[i][0], w.k.xpos[i][1], w.k.xpos[i][2]};
      cross3(t, oi, F);
      const real mo[3] = {N[0] - t[0], N[1] - t[1], N[2] - t[2]};
      const real val = lm.jtype[i] == KM_JNT_SLIDE ? dot3(ai, F) : dot3(ai, mo);
      mcol[c] = ((am >> i) & 1u) ? val : 0.0;
    }
    // (scheduling hint for the block above: all the rows' LDS reads first, then the arithmetic)
    __builtin_amdgcn_sched_group_barrier(0x100, 8 * W, 0);
    __builtin_amdgcn_sched_group_barrier(0x002, 64 * W, 0);
#pragma unroll
    for (int c = 0; c < W; c++) {
      const int i = base + c;
      if (W != NL && i >= NL) continue;
      // both triangles: the inversion then reads plain rows; entry (a, b) is written by the lane of link max(a, b) only
      real* const up = i <= j ? &w.Minv[i][j] : &w.tmp[j];
      real* const lo = i <= j ? &w.Minv[j][i] : &w.tmp[j];
      *up = mcol[c];
      *lo = mcol[c];
    }
    // bias_j = axis_j . (subtree wrench about the joint)
    const real Fb[3] = {acc[10], acc[11], acc[12]};
    if (slide) w.bias[j] = dot3(ax, Fb);
    else {
      cross3(t, oj, Fb);
      const real mo[3] = {acc[13] - t[0], acc[14] - t[1], acc[15] - t[2]};
      w.bias[j] = dot3(ax, mo);
    }
  }
}

// lower triangle from the upper one (column j only wrote rows i <= j along its ancestor path)
template <int NL, int G>
__device__ __forceinline__ void mass_symmetrize(Ws<NL>& w, int sub) {
  for (int j = sub; j < NL; j += G)
    for (int i = j + 1; i < NL; i++) w.Minv[i][j] = w.Minv[j][i];
}

// ---------------------------------------------------------------------------------------------
// Velocity-product + gravity wrench of every body (then bias_j = sum_b J_bj^T wrench_b), one link per lane.  omega, alpha and the origin acceleration of a link are sums of per-link
// increments over its ancestors, so each lane first publishes its increment (LDS), then sums along its own
// ancestor mask in root-to-leaf order (the order of the serial recursion):
//   omega_i = sum_j wv_j,          wv_j = axis_j qvel_j (hinge)
//   alpha_i = sum_j cz_j (hinge),  cz_j = omega_parent(j) x wv_j
//   a_i     = -g + sum_j d_j,      d_j  = alpha_p x r_j + omega_p x (omega_p x r_j) (+ 2 cz_j for a slide)
template <int NL, int G>
__device__ __forceinline__ void bias_bodies_parallel(Ws<NL>& w, const LModel<NL>& lm, const KModelDesc* m, int sub) {
  real* wvb = &w.f.bsc[0][0];           // [NL][3] each
  real* czb = wvb + 3 * NL;
  real* dbb = czb + 3 * NL;
  const bool on = sub < NL;
  const bool slide = on && lm.jtype[sub] == KM_JNT_SLIDE;
  const uint32_t up = on ? (lm.anc[sub] & ~(1u << sub)) : 0u;       // proper ancestors
  real ax[3] = {0, 0, 0};
  if (on) {
    const real qv = w.qvel[sub];
    ax[0] = w.k.axis[sub][0] * qv; ax[1] = w.k.axis[sub][1] * qv; ax[2] = w.k.axis[sub][2] * qv;
    wvb[3 * sub] = slide ? 0.0 : ax[0]; wvb[3 * sub + 1] = slide ? 0.0 : ax[1]; wvb[3 * sub + 2] = slide ? 0.0 : ax[2];
  }
  GSYNC();
  real wp[3] = {0, 0, 0}, cz[3] = {0, 0, 0};
  if (on) {
#pragma unroll KM_TREE_UNROLL(NL)
    for (int j = 0; j < NL; j++) {                                     // (static addresses, taken by the ancestor bit; root-to-leaf order)
      const bool take = (up >> j) & 1u;
      const real v0 = wvb[3 * j], v1 = wvb[3 * j + 1], v2 = wvb[3 * j + 2];
      wp[0] += take ? v0 : 0.0; wp[1] += take ? v1 : 0.0; wp[2] += take ? v2 : 0.0;
    }
    cross3(cz, wp, ax);
    czb[3 * sub] = slide ? 0.0 : cz[0]; czb[3 * sub + 1] = slide ? 0.0 : cz[1]; czb[3 * sub + 2] = slide ? 0.0 : cz[2];
  }
  GSYNC();
  real alp[3] = {0, 0, 0};
  if (on) {
#pragma unroll KM_TREE_UNROLL(NL)
    for (int j = 0; j < NL; j++) {
      const bool take = (up >> j) & 1u;
      const real v0 = czb[3 * j], v1 = czb[3 * j + 1], v2 = czb[3 * j + 2];
      alp[0] += take ? v0 : 0.0; alp[1] += take ? v1 : 0.0; alp[2] += take ? v2 : 0.0;
    }
    const int p = lm.parent[sub];
    real op[3] = {0, 0, 0};
    if (p >= 0) { op[0] = w.k.xpos[p][0]; op[1] = w.k.xpos[p][1]; op[2] = w.k.xpos[p][2]; }
    real r[3] = {w.k.xpos[sub][0] - op[0], w.k.xpos[sub][1] - op[1], w.k.xpos[sub][2] - op[2]}, t1[3], t2[3];
    cross3(t1, alp, r);
    cross3(t2, wp, r); cross3(t2, wp, t2);
#pragma unroll
    for (int c = 0; c < 3; c++) dbb[3 * sub + c] = t1[c] + t2[c] + (slide ? 2 * cz[c] : 0.0);
  }
  GSYNC();
  if (on) {
    real ai[3] = {-m->gravity[0], -m->gravity[1], -m->gravity[2]};
    const uint32_t am = lm.anc[sub];
#pragma unroll KM_TREE_UNROLL(NL)
    for (int j = 0; j < NL; j++) {
      const bool take = (am >> j) & 1u;
      const real v0 = dbb[3 * j], v1 = dbb[3 * j + 1], v2 = dbb[3 * j + 2];
      ai[0] += take ? v0 : 0.0; ai[1] += take ? v1 : 0.0; ai[2] += take ? v2 : 0.0;
    }
    real wi[3] = {wp[0], wp[1], wp[2]}, ali[3] = {alp[0], alp[1], alp[2]};
    if (!slide) {
#pragma unroll
      for (int c = 0; c < 3; c++) { wi[c] += ax[c]; ali[c] += cz[c]; }
    }
    const int i = sub;
    real cr[3] = {w.k.cpos[i][0] - w.k.xpos[i][0], w.k.cpos[i][1] - w.k.xpos[i][1], w.k.cpos[i][2] - w.k.xpos[i][2]}, t1[3], t2[3];
    cross3(t1, ali, cr);
    cross3(t2, wi, cr); cross3(t2, wi, t2);
    real wl[3], all[3], Iw[3], nl3[3], nw[3];
    matT_vec3(wl, w.k.xmat[i], wi);
    matT_vec3(all, w.k.xmat[i], ali);
#pragma unroll
    for (int c = 0; c < 3; c++) Iw[c] = lm.inertia[i][c] * wl[c];
    cross3(nl3, wl, Iw);
#pragma unroll
    for (int c = 0; c < 3; c++) nl3[c] += lm.inertia[i][c] * all[c];
    mat_vec3(nw, w.k.xmat[i], nl3);
    real Fi[3] = {lm.mass[i] * (ai[0] + t1[0] + t2[0]), lm.mass[i] * (ai[1] + t1[1] + t2[1]), lm.mass[i] * (ai[2] + t1[2] + t2[2])};
    real cpi[3] = {w.k.cpos[i][0], w.k.cpos[i][1], w.k.cpos[i][2]}, sh[3];
    cross3(sh, cpi, Fi);                       // shift the moment from the com to the world origin
#pragma unroll
    for (int c = 0; c < 3; c++) { w.f.FN[i][c] = Fi[c]; w.f.FN[i][3 + c] = nw[c] + sh[c]; }
  } else if (sub == NL) {
    // cube (free joint, qvel = [v_world, w_body]): bias = [-m g, w x I w]
    real wv[3] = {w.qvel[NL + 3], w.qvel[NL + 4], w.qvel[NL + 5]};
    real Iw[3] = {m->cube_inertia[0] * wv[0], m->cube_inertia[1] * wv[1], m->cube_inertia[2] * wv[2]}, t[3];
    cross3(t, wv, Iw);
#pragma unroll
    for (int c = 0; c < 3; c++) { w.bias[NL + c] = -m->cube_mass * m->gravity[c]; w.bias[NL + 3 + c] = t[c]; }
  }
}

// One-row groups: the bias-wrench pass with the three ancestor sums as broadcast-FMAs (s += bcast_j(v) * [j in mask], link
// order = root-to-leaf order) instead of LDS publish / synchronise / read rounds; the link's wrench stays in registers (FN).
template <int NL, int G>
__device__ __forceinline__ void anc_sum3(uint32_t mask, const real* v, real* s) {      // (NL here = the row's width)
  static_for<0, NL>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    const real take = ((mask >> j) & 1u) ? 1.0 : 0.0;
    dppfma3<false, j, j, j, j == 0>(s[0], v[0], take, s[1], v[1], take, s[2], v[2], take);
  });
}
// cube (free joint, qvel = [v_world, w_body]): bias = [-m g, w x I w]
template <int NL>
__device__ __forceinline__ void cube_bias(Ws<NL>& w, const KModelDesc* m) {
  real wc[3] = {w.qvel[NL + 3], w.qvel[NL + 4], w.qvel[NL + 5]};
  real Iw[3] = {m->cube_inertia[0] * wc[0], m->cube_inertia[1] * wc[1], m->cube_inertia[2] * wc[2]}, t[3];
  cross3(t, wc, Iw);
#pragma unroll
  for (int c = 0; c < 3; c++) { w.bias[NL + c] = -m->cube_mass * m->gravity[c]; w.bias[NL + 3 + c] = t[c]; }
}
// W = links per DPP row.  One-row groups: the row holds the whole robot (li = sub, base = 0, W = NL).  Two-row groups with a
// block split (two-arm models): each row holds one block of the robot -- lane c of a row works on link li = base + c of ITS
// block, masks are taken relative to the block's first link, and both blocks go through the same instructions at once.
// kin != nullptr (one-row groups, lane = link): the link's own frame from fk_parallel's registers instead of LDS.
template <int NL, int W>
__device__ __forceinline__ void bias_bodies_rows(Ws<NL>& w, const LModel<NL>& lm, const KModelDesc* m, int li, int base, bool cube_lane, real (&FN)[6],
                                                 const real* kin = nullptr) {
  static_assert(W <= 16, "one DPP row per block");
  constexpr int G = 16;
  const bool on = li >= 0;
  const int i = on ? li : 0;
  // (round 6) the link's constants and state in one batch; the parent's origin -- the one dependent read -- right behind it
  int jti = lm.jtype[i], pari = lm.parent[i];
  uint32_t anci = lm.anc[i];
  real qvi = w.qvel[i], in0 = lm.inertia[i][0], in1 = lm.inertia[i][1], in2 = lm.inertia[i][2], massi = lm.mass[i];
  real Rk[9], xo[3], cpi[3];
  if (kin) {
#pragma unroll
    for (int c = 0; c < 9; c++) Rk[c] = kin[c];
#pragma unroll
    for (int c = 0; c < 3; c++) { xo[c] = kin[9 + c]; cpi[c] = kin[12 + c]; }
  } else {
#pragma unroll
    for (int c = 0; c < 9; c++) Rk[c] = w.k.xmat[i][c];
#pragma unroll
    for (int c = 0; c < 3; c++) { xo[c] = w.k.xpos[i][c]; cpi[c] = w.k.cpos[i][c]; }
    km_pin(Rk); km_pin(xo, cpi);
  }
  km_pin_i(jti, pari); asm volatile("" : "+v"(anci)); km_pin(qvi, in0, in1, in2, massi);
  real op[3] = {0, 0, 0};
  { const int pc = pari >= 0 ? pari : 0; op[0] = w.k.xpos[pc][0]; op[1] = w.k.xpos[pc][1]; op[2] = w.k.xpos[pc][2]; }
  if (!(on && pari >= 0)) { op[0] = 0; op[1] = 0; op[2] = 0; }
  const bool slide = on && jti == KM_JNT_SLIDE;
  const uint32_t am = on ? anci >> base : 0u, up = am & ~(1u << (i - base));       // ancestors incl. self / proper ancestors (row-local bits)
  const real qv = on ? qvi : 0.0;
  const real ax[3] = {Rk[2] * qv, Rk[5] * qv, Rk[8] * qv};                           // (the joint axis = third column of the link's rotation)
  const real wv[3] = {(slide || !on) ? 0.0 : ax[0], (slide || !on) ? 0.0 : ax[1], (slide || !on) ? 0.0 : ax[2]};
  real wp[3] = {0, 0, 0}, cz[3], alp[3] = {0, 0, 0};
  anc_sum3<W, G>(up, wv, wp);
  cross3(cz, wp, ax);
  const real czv[3] = {(slide || !on) ? 0.0 : cz[0], (slide || !on) ? 0.0 : cz[1], (slide || !on) ? 0.0 : cz[2]};
  anc_sum3<W, G>(up, czv, alp);
  real r[3] = {xo[0] - op[0], xo[1] - op[1], xo[2] - op[2]}, t1[3], t2[3], db[3];
  cross3(t1, alp, r);
  cross3(t2, wp, r); cross3(t2, wp, t2);
#pragma unroll
  for (int c = 0; c < 3; c++) db[c] = on ? t1[c] + t2[c] + (slide ? 2 * cz[c] : 0.0) : 0.0;
  real ai[3] = {-m->gravity[0], -m->gravity[1], -m->gravity[2]};
  anc_sum3<W, G>(am, db, ai);
  if (on) {
    real wi[3] = {wp[0], wp[1], wp[2]}, ali[3] = {alp[0], alp[1], alp[2]};
    if (!slide) {
#pragma unroll
      for (int c = 0; c < 3; c++) { wi[c] += ax[c]; ali[c] += cz[c]; }
    }
    real cr[3] = {cpi[0] - xo[0], cpi[1] - xo[1], cpi[2] - xo[2]};
    cross3(t1, ali, cr);
    cross3(t2, wi, cr); cross3(t2, wi, t2);
    real wl[3], all[3], Iw[3], nl3[3], nw[3];
    const real inr[3] = {in0, in1, in2};
    matT_vec3(wl, Rk, wi);
    matT_vec3(all, Rk, ali);
#pragma unroll
    for (int c = 0; c < 3; c++) Iw[c] = inr[c] * wl[c];
    cross3(nl3, wl, Iw);
#pragma unroll
    for (int c = 0; c < 3; c++) nl3[c] += inr[c] * all[c];
    mat_vec3(nw, Rk, nl3);
    real Fi[3] = {massi * (ai[0] + t1[0] + t2[0]), massi * (ai[1] + t1[1] + t2[1]), massi * (ai[2] + t1[2] + t2[2])}, sh[3];
    cross3(sh, cpi, Fi);                       // shift the moment from the com to the world origin
#pragma unroll
    for (int c = 0; c < 3; c++) { FN[c] = Fi[c]; FN[3 + c] = nw[c] + sh[c]; }
  } else {
#pragma unroll
    for (int c = 0; c < 6; c++) FN[c] = 0;
    if (cube_lane) cube_bias<NL>(w, m);
  }
}

template <int NL, int G>
__device__ __forceinline__ void bias_project(Ws<NL>& w, const LModel<NL>& lm, int sub) {
  // FN[j] now holds the accumulated wrench of subtree(j) about the world origin
  for (int j = sub; j < NL; j += G) {
    const real aj[3] = {w.k.axis[j][0], w.k.axis[j][1], w.k.axis[j][2]};
    const real F[3] = {w.f.FN[j][0], w.f.FN[j][1], w.f.FN[j][2]};
    if (lm.jtype[j] == KM_JNT_SLIDE) w.bias[j] = dot3(aj, F);
    else {
      const real oj[3] = {w.k.xpos[j][0], w.k.xpos[j][1], w.k.xpos[j][2]};
      real t[3];
      cross3(t, oj, F);
      real mo[3] = {w.f.FN[j][3] - t[0], w.f.FN[j][4] - t[1], w.f.FN[j][5] - t[2]};
      w.bias[j] = dot3(aj, mo);
    }
  }
}

// upd[j] -= bcast_K(a[j]) * f for j in [J0, J1), j != K (Gauss-Jordan row update), runs of four / singles
template <int G, int K, int J0, int J1, int N>
__device__ __forceinline__ void gj_cols(real (&upd)[N], const real (&a)[N], real f) {
  if constexpr (G == 16 && J1 - J0 >= 4 && !(K >= J0 && K < J0 + 4)) {
    dppfma4<true, K, K, K, K>(upd[J0], a[J0], f, upd[J0 + 1], a[J0 + 1], f, upd[J0 + 2], a[J0 + 2], f, upd[J0 + 3], a[J0 + 3], f);
    gj_cols<G, K, J0 + 4, J1>(upd, a, f);
  } else if constexpr (J1 - J0 >= 1) {
    if constexpr (J0 != K) fnmac_b<G, K>(upd[J0], bsrc<G>(a[J0]), f);
    gj_cols<G, K, J0 + 1, J1>(upd, a, f);
  }
}

// Minv <- inverse of the SPD joint-space inertia held in Minv.  Lane i takes row i into registers and the
// group runs an in-place Gauss-Jordan sweep (no pivoting: every pivot of an SPD matrix is a positive Schur
// complement); row k reaches the other lanes through DPP row broadcasts, so there is no LDS traffic and no
// synchronisation inside the n^2 loop.
// In-place Gauss-Jordan sweep of the SPD matrix whose row `me` this lane holds in a[0..N) (one matrix per DPP row; no
// pivoting: every pivot of an SPD matrix is a positive Schur complement).  Row k reaches the other lanes through DPP row
// broadcasts, so there is no LDS traffic and no synchronisation inside the n^2 loop.
// a[j] -= bcast_K(a[j]) * f IN PLACE for j in [J0, J1), j != K: each instruction reads its own destination register through
// DPP (lane K's copy, before any lane writes it) -- no second register set, no selects.  Every run spends the two DPP wait
// states: a source may have been written by a plain select (the loads' masking before the first pivot, the pivot column's
// update) that the scheduler is free to place directly in front of the run.
template <int K, int J0, int J1, int N>
__device__ __forceinline__ void gj_cols_inplace(real (&a)[N], real f) {
#define KM_GJ1(I) "v_fmac_f64_dpp %" #I ", -%" #I ", %4 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\t"
  if constexpr (J1 - J0 >= 4 && !(K >= J0 && K < J0 + 4)) {
    asm volatile("s_nop 1\n\t" KM_GJ1(0) KM_GJ1(1) KM_GJ1(2) KM_GJ1(3) : "+v"(a[J0]), "+v"(a[J0 + 1]), "+v"(a[J0 + 2]), "+v"(a[J0 + 3]) : "v"(f), "n"(K));
    gj_cols_inplace<K, J0 + 4, J1>(a, f);
  } else if constexpr (J1 - J0 >= 1) {
    if constexpr (J0 != K) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, -%0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(a[J0]) : "v"(f), "n"(K));
    gj_cols_inplace<K, J0 + 1, J1>(a, f);
  }
#undef KM_GJ1
}
template <int G, int N>
__device__ __forceinline__ void gj_invert_rows(real (&a)[N], int me_idx, int& bad) {
  static_for<0, N>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    if constexpr (G == 16) {
      // one-row form (round 3): the pivot row scales itself through the same update as the others -- with f = 1 - d on the
      // pivot's own lane, a_kj - a_kj (1 - d) = a_kj d -- so a pivot costs one select for f and one for the pivot column
      // instead of two per column, and the update runs in place
      if constexpr (k > 0) dpp_settle(a[k]);           // (written by the previous pivot's runs: the broadcast below is compiler code)
      const real pk = gbcast<G, k>(a[k]);
      bad |= !(pk > 0);
      const real d = frcp(pk);
      const bool me = me_idx == k;
      const real f = me ? 1.0 - d : a[k] * d;
      gj_cols_inplace<k, 0, N>(a, f);
      a[k] = me ? d : -f;
    } else {
    real pk = gbcast<G, k>(a[k]);
    if (!(pk > 0)) { bad = 1; pk = 1; }
    const real d = frcp(pk);
    const real aik = a[k];
    const bool me = me_idx == k;
    // a_ij - (a_ik / p) a_kj for every column j != k, row k arriving by DPP: all columns of a pivot are independent, so
    // they go in runs of four behind one pair of wait states
    real upd[N];
    const real f = aik * d;
#pragma unroll
    for (int j = 0; j < N; j++) upd[j] = a[j];
    gj_cols<G, k, 0, N>(upd, a, f);
#pragma unroll
    for (int j = 0; j < N; j++) if (j != k) a[j] = me ? a[j] * d : upd[j];
    a[k] = me ? d : -aik * d;
    }
  });
}

// Two-arm models: the trees [0, split) and [split, NL) share no dof, so the inertia is two diagonal blocks.  Row r of the
// group inverts block r (lane c <-> link base + c, the mapping of the tree passes; M comes from and goes back to LDS by link
// index, so nothing has to be moved between lanes): two <= KM_BLOCK_MAX-pivot sweeps side by side with the one-row
// broadcasts, instead of one NL-pivot sweep across two rows.  Lanes and columns beyond a block's size carry identity rows.
// The same operations per block in the same order as the full sweep does them (the off-block entries it carries are exact
// zeros), so the result is bitwise the same.
template <int NL, int G>
__device__ __forceinline__ void invert_mass_blocks(Ws<NL>& w, int sub, CReg<NL>& cr, int split, Prof& pf) {
  constexpr int NB = KM_BLOCK_MAX;
  const int row = (threadIdx.x >> 4) & 1, c = threadIdx.x & 15;
  const int base = row ? split : 0, nb = row ? NL - split : split;
  const bool on = c < nb;
  const int li = base + c;
  real loc[NB];
  {
    // unconditional loads at clamped addresses, then selects on the values (conditional loads would each become a branch)
    const int rs = sub < NL ? sub : NL - 1, rl = on ? li : NL - 1;
    real full[NL], blk[NB];
#pragma unroll
    for (int j = 0; j < NL; j++) full[j] = w.Minv[rs][j];
#pragma unroll
    for (int k = 0; k < NB; k++) blk[k] = w.Minv[rl][base + k < NL ? base + k : NL - 1];
#pragma unroll
    for (int j = 0; j < NL; j++) cr.mrow[j] = sub < NL ? full[j] : 0.0;
#pragma unroll
    for (int k = 0; k < NB; k++) loc[k] = (on && k < nb) ? blk[k] : ((!on && k == c) ? 1.0 : 0.0);
  }
  GSYNC();
  pf.ph(39);
  int bad = 0;
  gj_invert_rows<16, NB>(loc, c, bad);
  if (__any(bad)) { const int gb = gor<G>(bad); if (gb && sub == 0) w.bad = 1; }      // (wave-uniform branch; never taken on sane models)
  if (on) {
#pragma unroll
    for (int j = 0; j < NL; j++) w.Minv[li][j] = 0.0;
#pragma unroll
    for (int k = 0; k < NB; k++) if (k < nb) w.Minv[li][base + k] = loc[k];
  }
  GSYNC();
}

template <int NL, int G>
__device__ __forceinline__ void invert_mass(Ws<NL>& w, int sub, CReg<NL>& cr, int split, Prof& pf) {
  if constexpr (G == 32) { if (split) { invert_mass_blocks<NL, G>(w, sub, cr, split, pf); return; } }
  real a[NL];
  if constexpr (G == 16) {                      // (composite_mass_bias_rows wrote both triangles: plain rows, unconditional loads)
    const int rs = sub < NL ? sub : NL - 1;
#pragma unroll
    for (int j = 0; j < NL; j++) a[j] = w.Minv[rs][j];
#pragma unroll
    for (int j = 0; j < NL; j++) a[j] = sub < NL ? a[j] : 0.0;
  } else {
#pragma unroll
    for (int j = 0; j < NL; j++) a[j] = sub < NL ? (j >= sub ? w.Minv[sub][j] : w.Minv[j][sub]) : 0.0;   // columns hold the upper triangle
  }
  GSYNC();
#pragma unroll
  for (int j = 0; j < NL; j++) cr.mrow[j] = a[j];
  int bad = 0;
  gj_invert_rows<G, NL>(a, sub, bad);
  if (bad && sub == 0) w.bad = 1;
  if (sub < NL) {
#pragma unroll
    for (int j = 0; j < NL; j++) w.Minv[sub][j] = a[j];
  }
  GSYNC();
}

// mju_makeFrame
__device__ __forceinline__ void make_frame(real* fr) {
  normalize3_fast(fr);
  real y[3] = {0, 0, 0};
  if (fr[1] < 0.5 && fr[1] > -0.5) y[1] = 1; else y[2] = 1;
  real t = dot3(fr, y);
  y[0] -= t * fr[0]; y[1] -= t * fr[1]; y[2] -= t * fr[2];
  normalize3_fast(y);
  fr[3] = y[0]; fr[4] = y[1]; fr[5] = y[2];
  cross3(fr + 6, fr, fr + 3);
}

// contact frame of every contact whose normal is the table normal (+z): mju_makeFrame((0,0,1)) = rows n, t1, t2
#define KM_PLANE_FRAME {0, 0, 1, 0, 1, 0, -1, 0, 0}
// narrow phase for the fixed candidate set, written into fixed slots: plane-box (first 4 corners below the
// table -> slots 0..3 in corner order), sphere-box (slots 4.., the first NSS penetrating spheres), plane-sphere (slots 4 + NSS..).
// One candidate per lane: lanes 0..7 test the cube corners (slot = rank among the penetrating corners, from the
// group's ballot bits), lanes 8..8+NSPH-1 their collision sphere against cube and table.
// the table top is a rectangle (kmanip.h table_rect): a point is over it while its x, y lie inside.  tr = the four bounds, fetched
// ONCE by the caller (one scalar load); `&` not `&&`: four compares, no branch per bound
__device__ __forceinline__ bool over_table(const real (&tr)[4], const real* p) {
  return (p[0] >= tr[0]) & (p[0] <= tr[1]) & (p[1] >= tr[2]) & (p[1] <= tr[3]);
}

// NEAR (the trailing mj_step1 of the two-arm kernels only): also report whether some collider is within KM_NEAR_MARGIN of the cube
// without touching it -- the onset of the coupled Newton loop is what k_sort_envs' last-step counters cannot see coming
#define KM_NEAR_MARGIN 0.015      // (the default of callers that pass none; the handle's value is KDeviceState::near_margin: kmanip_api.hip)
template <int NL, int G, bool NEAR = false>
__device__ __forceinline__ int collide_parallel(Ws<NL>& w, const LModel<NL>& lm, const KModelDesc* m, int sub, real near_margin = KM_NEAR_MARGIN) {
  constexpr int NSPH = Dim<NL>::NSPH, NSS = Dim<NL>::NSS, NST = Dim<NL>::NST;
  static_assert(8 + NSPH <= G, "one lane per collision candidate");
  uint32_t mask = 0, act = 0;
  // (round 6) the cube's pose and this lane's candidate (sphere s = sub - 8: link, centre, radius, capsule segment) in one batch;
  // the candidate's link frame -- the one dependent read -- in a second
  const int sidx = sub >= 8 && sub - 8 < NSPH ? sub - 8 : 0;
  real cp[3] = {w.qpos[NL], w.qpos[NL + 1], w.qpos[NL + 2]}, cmat[9];
#pragma unroll
  for (int k = 0; k < 9; k++) cmat[k] = w.k.cube_mat[k];
  int lnk = lm.sph_link[sidx], nsph_ = lm.nsph;
  real slp[3] = {lm.sph_pos[sidx][0], lm.sph_pos[sidx][1], lm.sph_pos[sidx][2]}, sgp[3] = {lm.sph_seg[sidx][0], lm.sph_seg[sidx][1], lm.sph_seg[sidx][2]}, radp = lm.sph_rad[sidx];
  km_pin(cp); km_pin(cmat); km_pin(slp, sgp); km_pin(radp); km_pin_i(lnk, nsph_);
  real lmat[9], lpos[3];
#pragma unroll
  for (int k = 0; k < 9; k++) lmat[k] = w.k.xmat[lnk][k];
  lpos[0] = w.k.xpos[lnk][0]; lpos[1] = w.k.xpos[lnk][1]; lpos[2] = w.k.xpos[lnk][2];
  uint32_t lanc = lm.anc[lnk];
  const real tr[4] = {m->table_rect[0], m->table_rect[1], m->table_rect[2], m->table_rect[3]};
  bool below = false;
  real c[3] = {0, 0, 0}, dist = 0;
  if (sub < 8) {
    const real loc[3] = {(sub & 1 ? 1 : -1) * m->cube_half[0], (sub & 2 ? 1 : -1) * m->cube_half[1], (sub & 4 ? 1 : -1) * m->cube_half[2]};
    mat_vec3(c, cmat, loc);
    c[0] += cp[0]; c[1] += cp[1]; c[2] += cp[2];
    dist = c[2] - m->table_z;
    below = (dist < 0) & over_table(tr, c);
  }
  const unsigned long long bal = __ballot(below);
  const uint32_t m8 = (uint32_t)(bal >> ((threadIdx.x & 63) - sub)) & 0xFFu;
  if (below) {
    const int n = __popc(m8 & ((1u << sub) - 1u));
    if (n < 4) {
      const real fr[9] = KM_PLANE_FRAME;
#pragma unroll
      for (int k = 0; k < 9; k++) w.c_frame[n][k] = fr[k];
      w.c_dist[n] = dist;
      w.c_pos[n][0] = c[0]; w.c_pos[n][1] = c[1]; w.c_pos[n][2] = c[2] - 0.5 * dist;
      mask |= KM_CON_CUBE_TABLE(sub); act |= 1u << n;
    }
  }
  const int nsph = nsph_;
  const int s = sub - 8;
  bool hitc = false, hitt = false;
  real ctr[3] = {0, 0, 0}, ctrt[3] = {0, 0, 0}, nloc[3] = {0, 0, 0}, d1 = 0, d2 = 0, rad = 0;
  km_pin(lmat); km_pin(lpos); asm volatile("" : "+v"(lanc));
  if (sub >= 8 && sub < 8 + nsph) {
    real sl[3] = {slp[0], slp[1], slp[2]}, rel[3], loc[3], cl[3];
    mat_vec3(ctr, lmat, sl);
#pragma unroll
    for (int a = 0; a < 3; a++) ctr[a] += lpos[a];
    rad = radp;
    // table plane (geom1) - sphere (geom2): the end sphere itself (a capsule meets a plane in its end spheres)
    d2 = ctr[2] - m->table_z - rad;
    hitt = (d2 < 0) & over_table(tr, ctr);
    ctrt[0] = ctr[0]; ctrt[1] = ctr[1]; ctrt[2] = ctr[2];
    // capsule section (kmanip.h sphere_seg): against the cube the collider is the point of the link's segment closest to the
    // cube centre -- a sphere sliding along the link
    {
      const real sg[3] = {sgp[0], sgp[1], sgp[2]};
      real sw[3];
      mat_vec3(sw, lmat, sg);
      const real ss = dot3(sw, sw);
      if (ss > 0) {
        real t = ((cp[0] - ctr[0]) * sw[0] + (cp[1] - ctr[1]) * sw[1] + (cp[2] - ctr[2]) * sw[2]) / ss;
        t = fmin(fmax(t, 0.0), 1.0);
#pragma unroll
        for (int a = 0; a < 3; a++) ctr[a] += t * sw[a];
      }
    }
#pragma unroll
    for (int a = 0; a < 3; a++) rel[a] = ctr[a] - cp[a];
    // sphere (geom1) - cube box (geom2)
    matT_vec3(loc, cmat, rel);
    bool inside = true;
#pragma unroll
    for (int a = 0; a < 3; a++) { cl[a] = fmin(fmax(loc[a], -m->cube_half[a]), m->cube_half[a]); if (cl[a] != loc[a]) inside = false; }
    if (!inside) {
      nloc[0] = cl[0] - loc[0]; nloc[1] = cl[1] - loc[1]; nloc[2] = cl[2] - loc[2];
      real dn = normalize3_fast(nloc);
      d1 = dn - rad;
    } else {
      int best = 0; real bd = INFINITY;
#pragma unroll
      for (int a = 0; a < 3; a++) { real dd = m->cube_half[a] - fabs(loc[a]); if (dd < bd) { bd = dd; best = a; } }
      real sg = (best == 0 ? loc[0] : (best == 1 ? loc[1] : loc[2])) >= 0 ? -1.0 : 1.0;
      if (best == 0) nloc[0] = sg; else if (best == 1) nloc[1] = sg; else nloc[2] = sg;
      d1 = -bd - rad;
    }
    hitc = d1 < 0;
  }
  // the first NSS penetrating spheres of each kind (sphere order) get the slots: rank = penetrating spheres on lower lanes
  const uint32_t below_me = (1u << sub) - 1u;
  const uint32_t mc = (uint32_t)(__ballot(hitc) >> ((threadIdx.x & 63) - sub)) & below_me;
  const uint32_t mt = (uint32_t)(__ballot(hitt) >> ((threadIdx.x & 63) - sub)) & below_me;
  if (hitc && __popc(mc) < NSS) {
    const int n = 4 + __popc(mc);
    real fr[9];
    mat_vec3(fr, cmat, nloc);
    make_frame(fr);
#pragma unroll
    for (int k = 0; k < 9; k++) w.c_frame[n][k] = fr[k];
    w.c_dist[n] = d1;
#pragma unroll
    for (int a = 0; a < 3; a++) w.c_pos[n][a] = ctr[a] + fr[a] * (rad + 0.5 * d1);
    w.slot_sph[n] = s; w.slot_anc[n] = lanc;
    mask |= KM_CON_SPHERE_CUBE(s); act |= 1u << n;
  }
  if (hitt && __popc(mt) < NST) {
    const int n = 4 + NSS + __popc(mt);
    const real fr[9] = KM_PLANE_FRAME;
#pragma unroll
    for (int k = 0; k < 9; k++) w.c_frame[n][k] = fr[k];
    w.c_dist[n] = d2;
    w.c_pos[n][0] = ctrt[0]; w.c_pos[n][1] = ctrt[1]; w.c_pos[n][2] = ctrt[2] - (rad + 0.5 * d2);
    w.slot_sph[n] = s; w.slot_anc[n] = lanc;
    mask |= KM_CON_SPHERE_TABLE(s); act |= 1u << n;
  }
  mask = (uint32_t)gor<G>((int)mask);
  act = (uint32_t)gor<G>((int)act);
  if (sub == 0) {
    w.cact = act; w.contact_mask = mask;
    w.touch_ct = (mask & KM_CON_ANY_CUBE_TABLE) != 0;
  }
  if constexpr (NEAR) return gor<G>((int)(sub >= 8 && sub < 8 + nsph && d1 < near_margin));
  return 0;
}

// MuJoCo impedance d(r) from the staged, pre-clamped solimp constants: no divide, no pow (power is 1 or 2: kmanip_create
// refuses any other value; every reference model uses the default 2)
template <class IMP> __device__ __forceinline__ real impedance_c(const IMP& p, real pos) {
  // (round 6: the seven staged constants fetched together and the cases as selects -- as early returns each case read its own
  // constants from LDS behind its own branch, eight round trips in a row; same expressions, same value)
  real d0 = p.d0, dw = p.dw, iw = p.iw, mid = p.mid, imid = p.imid, i1mid = p.i1mid;
  int mode = p.mode;
  km_pin(d0, dw, iw, mid, imid, i1mid); km_pin_i(mode);
  const real x = fabs(pos) * iw;
  const real y = mode == 1 ? x : ((x <= mid) ? x * x * imid : 1 - (1 - x) * (1 - x) * i1mid);
  real r = d0 + y * (dw - d0);
  r = x <= 0 ? d0 : r;
  r = x >= 1 ? dw : r;
  return mode == 0 ? 0.5 * (d0 + dw) : r;
}
// the same from constants the caller fetched (together with its other inputs)
__device__ __forceinline__ real impedance_v(real d0, real dw, real iw, real mid, real imid, real i1mid, int mode, real pos) {
  const real x = fabs(pos) * iw;
  const real y = mode == 1 ? x : ((x <= mid) ? x * x * imid : 1 - (1 - x) * (1 - x) * i1mid);
  real r = d0 + y * (dw - d0);
  r = x <= 0 ? d0 : r;
  r = x >= 1 ? dw : r;
  return mode == 0 ? 0.5 * (d0 + dw) : r;
}
template <class IMP> __device__ __forceinline__ void stage_imp(IMP& p, const real* si) {
  p.d0 = fmin(fmax(si[0], MJ_MINIMP), MJ_MAXIMP); p.dw = fmin(fmax(si[1], MJ_MINIMP), MJ_MAXIMP);
  const real width = fmax(MJ_MINVAL, si[2]);
  p.mid = fmin(fmax(si[3], MJ_MINIMP), MJ_MAXIMP);
  p.iw = 1.0 / width; p.imid = 1.0 / p.mid; p.i1mid = 1.0 / (1 - p.mid);
  p.mode = (p.d0 == p.dw || width <= MJ_MINVAL) ? 0 : (fmax(1.0, si[4]) == 1 ? 1 : 2);
}
// MuJoCo impedance / reference acceleration parameters (general form; staging only)
__device__ __forceinline__ real impedance(const real* si, real pos) {
  real d0 = fmin(fmax(si[0], MJ_MINIMP), MJ_MAXIMP), dw = fmin(fmax(si[1], MJ_MINIMP), MJ_MAXIMP);
  real width = fmax(MJ_MINVAL, si[2]), mid = fmin(fmax(si[3], MJ_MINIMP), MJ_MAXIMP), power = fmax(1.0, si[4]);
  if (d0 == dw || width <= MJ_MINVAL) return 0.5 * (d0 + dw);
  real x = fabs(pos) / width, y;
  if (x >= 1) return dw;
  if (x <= 0) return d0;
  if (power == 1) y = x;
  else y = (x <= mid) ? x * x / mid : 1 - (1 - x) * (1 - x) / (1 - mid);   // power 2, MuJoCo's default (others refused at create)
  (void)power;
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
__device__ __forceinline__ void point_jac_col(const Ws<NL>& w, const LModel<NL>& lm, int body, int j, const real* pt,
                                              real* jp, real* jr) {
  jp[0] = 0; jp[1] = 0; jp[2] = 0; jr[0] = 0; jr[1] = 0; jr[2] = 0;
  if (body < 0) return;
  if (body < NL) {
    if (j >= NL || !((lm.anc[body] >> j) & 1u)) return;
    if (lm.jtype[j] == KM_JNT_SLIDE) { jp[0] = w.k.axis[j][0]; jp[1] = w.k.axis[j][1]; jp[2] = w.k.axis[j][2]; }
    else {
      real ax[3] = {w.k.axis[j][0], w.k.axis[j][1], w.k.axis[j][2]};
      real r[3] = {pt[0] - w.k.xpos[j][0], pt[1] - w.k.xpos[j][1], pt[2] - w.k.xpos[j][2]};
      cross3(jp, ax, r);
      jr[0] = ax[0]; jr[1] = ax[1]; jr[2] = ax[2];
    }
    return;
  }
  if (j < NL) return;
  const int e = j - NL;
  if (e < 3) { jp[0] = e == 0; jp[1] = e == 1; jp[2] = e == 2; return; }
  const int k = e - 3;
  real col[3] = {w.k.cube_mat[k], w.k.cube_mat[3 + k], w.k.cube_mat[6 + k]};
  real r[3] = {pt[0] - w.qpos[NL], pt[1] - w.qpos[NL + 1], pt[2] - w.qpos[NL + 2]};
  cross3(jp, col, r);
  jr[0] = col[0]; jr[1] = col[1]; jr[2] = col[2];
}

// arm single-dof rows in mj_makeConstraint order (friction loss, then limits), enumerated by one lane
template <int NL>
__device__ __forceinline__ void scalar_rows_serial(Ws<NL>& w, const LModel<NL>& lm) {
  int n = 0;
  for (int j = 0; j < NL; j++) {
    real fl = lm.floss[j];
    if (fl > 0) { w.s_dof[n] = j; w.s_type[n] = 0; w.s_sign[n] = 1; w.s_pos[n] = 0; w.s_floss[n] = fl; n++; }
  }
  for (int j = 0; j < NL; j++) {
    real dl = w.qpos[j] - lm.range[j][0], du = lm.range[j][1] - w.qpos[j];
    if (dl < 0) { w.s_dof[n] = j; w.s_type[n] = 1; w.s_sign[n] = 1; w.s_pos[n] = dl; w.s_floss[n] = 0; n++; }
    if (du < 0) { w.s_dof[n] = j; w.s_type[n] = 1; w.s_sign[n] = -1; w.s_pos[n] = du; w.s_floss[n] = 0; n++; }
  }
  w.ns = n;
}

// Constraint assembly.  Arm single-dof rows: parameters in LDS (parallel over rows).  Contacts: this lane's
// Jacobian column of each basis row in registers (cr.jb), B = M^-1 J^T columns (cr.bb), then per-contact
// Gram / edge tables (group-uniform) in LDS.
template <int NL, int G>
__device__ __forceinline__ void build_constraints(Ws<NL>& w, const LModel<NL>& lm, const KModelDesc* m, int sub,
                                                  CReg<NL>& cr, real invm) {
  constexpr int NV = Dim<NL>::NV, NC = Dim<NL>::NC;
  for (int r = sub; r < w.ns; r += G) {
    const int j = w.s_dof[r];
    real Ad = w.Minv[j][j];
    real pos = w.s_pos[r];
    real imp = impedance_c(lm.imp[0], pos), kk = lm.kb[0][0], bb = lm.kb[0][1];
    const real R = fmax(MJ_MINVAL, (1 - imp) * frcp(imp) * lm.dofw[j]);      // efc_diagApprox = dof_invweight0, not the exact A_ii
    w.s_R[r] = R;
    w.s_den[r] = Ad + R;
    w.s_inv[r] = 1.0 / (Ad + R);
    w.s_aref[r] = -bb * (w.s_sign[r] * w.qvel[j]) - kk * imp * pos;
  }
  const uint32_t act = w.cact;
  // ---- J columns (needs kinematics, which the Gram tables will overwrite: finish all slots first)
#pragma unroll
  for (int c = 0; c < NC; c++) {
    cr.jb[c][0] = 0; cr.jb[c][1] = 0; cr.jb[c][2] = 0; cr.jb[c][3] = 0;
    if (((act >> c) & 1u) && sub < NV) {
      const int kind = slot_kind<NL>(c);
      const int link = kind == 0 ? -1 : lm.sph_link[w.slot_sph[c]];
      const int b1 = kind == 1 ? link : -1, b2 = kind == 2 ? link : NL;   // geom1 / geom2 bodies
      real pt[3] = {w.c_pos[c][0], w.c_pos[c][1], w.c_pos[c][2]};
      real p1[3], r1[3], p2[3], r2[3];
      point_jac_col<NL>(w, lm, b1, sub, pt, p1, r1);
      point_jac_col<NL>(w, lm, b2, sub, pt, p2, r2);
      real dl[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]}, dr[3] = {r2[0] - r1[0], r2[1] - r1[1], r2[2] - r1[2]};
      cr.jb[c][0] = dot3(w.c_frame[c], dl);
      cr.jb[c][1] = dot3(w.c_frame[c] + 3, dl);
      cr.jb[c][2] = dot3(w.c_frame[c] + 6, dl);
      cr.jb[c][3] = dot3(w.c_frame[c], dr);
    }
  }
  GSYNC();
  // ---- B = M^-1 J^T for the slots with arm dofs: arm lanes need the whole row -> stage through LDS;
  // cube lanes (and every lane of a table-cube slot) just scale by the diagonal
#pragma unroll
  for (int c = 4; c < NC; c++) {
    cr.bb[c - 4][0] = 0; cr.bb[c - 4][1] = 0; cr.bb[c - 4][2] = 0; cr.bb[c - 4][3] = 0;
    __builtin_amdgcn_sched_barrier(0);
    if ((act >> c) & 1u) {
      if (sub < NV) {
#pragma unroll
        for (int k = 0; k < 4; k++) w.stage[k][sub] = cr.jb[c][k];
      }
      GSYNC();
      if (sub < NL) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
          real s = 0;
          for (int j = 0; j < NL; j++) s += w.Minv[sub][j] * w.stage[k][j];
          cr.bb[c - 4][k] = s;
        }
      } else if (sub < NV) {
#pragma unroll
        for (int k = 0; k < 4; k++) cr.bb[c - 4][k] = cr.jb[c][k] * invm;
      }
      GSYNC();
    }
  }
  // ---- Gram matrix + edge tables per slot (all lanes get identical sums; lane 0 stores)
  const real qv = sub < NV ? w.qvel[sub] : 0.0;
#pragma unroll
  for (int c = 0; c < NC; c++) {
    __builtin_amdgcn_sched_barrier(0);
    if ((act >> c) & 1u) {
      const int kind = slot_kind<NL>(c);
      real Gm[4][4], vb[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        vb[k] = gsum<G>(cr.jb[c][k] * qv);
#pragma unroll
        for (int l = k; l < 4; l++) {
          const real bl = c < 4 ? cr.jb[c][l] * invm : cr.bb[c < 4 ? 0 : c - 4][l];
          Gm[k][l] = gsum<G>(cr.jb[c][k] * bl); Gm[l][k] = Gm[k][l];
        }
      }
      const bool cube = kind != 2;
      const real* fr = cube ? m->con_cube_friction : m->con_def_friction;
      const real* sr = cube ? m->con_cube_solref : m->con_def_solref;
      const real* si = cube ? m->con_cube_solimp : m->con_def_solimp;
      real mu[3] = {fr[0], fr[0], fr[1]};
      const real dist = w.c_dist[c];
      real imp = impedance_c(lm.imp[cube ? 1 : 0], dist), kk = lm.kb[cube ? 1 : 0][0], bb = lm.kb[cube ? 1 : 0][1];
      (void)sr; (void)si;
      const int ne = kind == 2 ? 4 : 6;
      real R = 0;
      ConRec& rc = w.rec[c];
#pragma unroll
      for (int e = 0; e < 6; e++) {
        const int k = e / 2 + 1;
        const real sm = (e & 1) ? -mu[k - 1] : mu[k - 1];
        real Ge[4];
#pragma unroll
        for (int l = 0; l < 4; l++) Ge[l] = Gm[l][0] + sm * Gm[l][k];        // J_l . M^-1 (J_0 + sm J_k)^T
        const real Ad = Ge[0] + sm * Ge[k];
        if (e == 0) R = 2 * fr[0] * fr[0] * fmax(MJ_MINVAL, (1 - imp) * frcp(imp) * (kind == 0 ? lm.cornerA : lm.sphA[kind == 2][w.slot_sph[c]]));
        const real vel = vb[0] + sm * vb[k];
        if (sub == 0) {
          rc.den[e] = Ad + R;
          rc.inv[e] = e < ne ? 1.0 / (Ad + R) : 0.0;
          rc.aref[e] = -bb * vel - kk * imp * dist;
          rc.f[e] = 0;
#pragma unroll
          for (int l = 0; l < 4; l++) w.p.Ge[c][e][l] = Ge[l];
        }
      }
      if (sub == 0) { rc.R = R; rc.mu[0] = mu[0]; rc.mu[1] = mu[1]; rc.mu[2] = mu[2]; }
    }
  }
  GSYNC();
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
__device__ __forceinline__ real solve_accel(Ws<NL>& w, const LModel<NL>& lm, const KModelDesc* m, int sub, int actuation,
                                            CReg<NL>& cr, real invm) {
  constexpr int NV = Dim<NL>::NV, NC = Dim<NL>::NC;
  // ---- actuation (position servos on actuator_length = q at mj_step1 time) and smooth acceleration
  if (sub < NV) {
    real rhs = -w.bias[sub];
    if (actuation && sub < NL) {
      real c = fmin(fmax(w.ctrl[sub], lm.ctrlrange[sub][0]), lm.ctrlrange[sub][1]);
      real force = lm.kp[sub] * c - lm.kp[sub] * w.qpos[sub];
      if (lm.forcelimited[sub]) force = fmin(fmax(force, lm.forcerange[sub][0]), lm.forcerange[sub][1]);
      rhs += force;
    }
    w.tmp[sub] = rhs;
  }
  GSYNC();
  real a_s = 0;
  if (sub < NL) { for (int j = 0; j < NL; j++) a_s += w.Minv[sub][j] * w.tmp[j]; }
  else if (sub < NV) a_s = w.tmp[sub] * invm;
  if (sub < NV) w.as[sub] = a_s;
  GSYNC();
  const int ns = w.ns;
  const uint32_t act = w.cact;
  const real warm = sub < NV ? w.warm[sub] : 0.0;
  // ---- the cube's friction-loss row owned by this lane (registers only)
  const bool my_row = sub >= NL && sub < NV && m->cube_frictionloss > 0;
  real my_f = 0, my_aref = 0, my_R = 1, my_den = 1, my_inv = 0;
  const real my_fl = m->cube_frictionloss;
  if (my_row) {
    real imp = lm.imp0[0], bb = lm.kb[0][1];
    my_R = fmax(MJ_MINVAL, (1 - imp) * frcp(imp) * lm.cubew[sub < NL + 3 ? 0 : 1]);
    my_den = invm + my_R;
    my_inv = 1.0 / my_den;
    my_aref = -bb * w.qvel[sub];
  }
  // ---- warm start: forces implied by qacc_warmstart, kept only if the dual cost is negative
  real cost_rows = 0, y = 0;
  for (int r = sub; r < ns; r += G) {
    const int j = w.s_dof[r];
    const real sg = w.s_sign[r], R = w.s_R[r], aref = w.s_aref[r];
    real jar = sg * w.warm[j] - aref, f;
    if (w.s_type[r] == 0) { const real fl = w.s_floss[r]; f = (jar <= -R * fl) ? fl : ((jar >= R * fl) ? -fl : -jar / R); }
    else f = jar < 0 ? -jar / R : 0.0;
    w.s_f[r] = f;
    cost_rows += 0.5 * R * f * f + f * (sg * w.as[j] - aref);
  }
  if (my_row) {
    real jar = warm - my_aref;
    my_f = (jar <= -my_R * my_fl) ? my_fl : ((jar >= my_R * my_fl) ? -my_fl : -jar / my_R);
    cost_rows += 0.5 * my_R * my_f * my_f + my_f * (a_s - my_aref);
    y += my_f;
  }
#pragma unroll
  for (int c = 0; c < NC; c++) {
    __builtin_amdgcn_sched_barrier(0);
    if ((act >> c) & 1u) {
      real wk[4], ak[4], F[4] = {0, 0, 0, 0};
      {
#pragma clang fp contract(off)
#pragma unroll
        for (int k = 0; k < 4; k++) { wk[k] = cr.jb[c][k] * warm; ak[k] = cr.jb[c][k] * a_s; }
        gsum_n<G, 4>(wk); gsum_n<G, 4>(ak);
      }
      ConRec& rc = w.rec[c];
      const real R = rc.R;
#pragma unroll
      for (int e = 0; e < 6; e++) {
        const int k = e / 2 + 1;
        const real sm = (e & 1) ? -rc.mu[k - 1] : rc.mu[k - 1];
        const real aref = rc.aref[e];
        real jar = wk[0] + sm * wk[k] - aref;
        real f = (rc.inv[e] != 0 && jar < 0) ? -jar / R : 0.0;
        if (sub == 0) { rc.f[e] = f; cost_rows += 0.5 * R * f * f + f * (ak[0] + sm * ak[k] - aref); }
        F[0] += f; F[k] += sm * f;
      }
#pragma unroll
      for (int k = 0; k < 4; k++) y += cr.jb[c][k] * F[k];
    }
  }
  GSYNC();
  // y = J^T f (lane j), z = M^-1 y
  if (sub < NL) { for (int r = 0; r < ns; r++) if (w.s_dof[r] == sub) y += w.s_sign[r] * w.s_f[r]; }
  if (sub < NV) w.tmp[sub] = y;
  GSYNC();
  real z = 0;
  if (sub < NL) { for (int j = 0; j < NL; j++) z += w.Minv[sub][j] * w.tmp[j]; }
  else if (sub < NV) z = y * invm;
  const real cost = gsum<G>(0.5 * y * z + cost_rows);
  real a = a_s;
  if (cost > 0) {
    for (int r = sub; r < ns; r += G) w.s_f[r] = 0;
    for (int c = sub; c < NC; c += G) for (int e = 0; e < 6; e++) w.rec[c].f[e] = 0;
    my_f = 0;
  } else a += z;
  GSYNC();
  // ---- projected Gauss-Seidel in acceleration space: a = a_s + M^-1 J^T f kept distributed (lane = dof).
  // Row order = mj_makeConstraint order.  The cube's friction-loss rows touch only the diagonal block of
  // M^-1, so the owning lanes update them locally and simultaneously -- identical to one after another,
  // and no cross-lane traffic.  A row on an arm dof needs one broadcast; a contact needs four DPP row
  // reductions (its basis projections u = J a), then its pyramid edges run on precomputed Gram rows.
  const real scale = lm.scale;
  const int maxit = m->solver_iterations;
  const real tol = m->solver_tolerance;
  for (int iter = 0; iter < maxit; iter++) {
    real improvement = 0, imp_local = 0;
    for (int r = 0; r < ns; r++) {
      const int j = w.s_dof[r];
      const real sg = w.s_sign[r];
      const real f = w.s_f[r];
      const real mij = sub < NL ? w.Minv[sub][j] : 0.0;
      const real Ja = sg * __shfl(a, j, G);
      const real dlt = pgs_row(Ja, w.s_aref[r], w.s_R[r], w.s_den[r], w.s_inv[r], f, w.s_type[r], w.s_floss[r], improvement);
      w.s_f[r] = f + dlt;
      a += sg * mij * dlt;
    }
    if (my_row) {
      const real dlt = pgs_row(a, my_aref, my_R, my_den, my_inv, my_f, 0, my_fl, imp_local);
      my_f += dlt;
      a += dlt * invm;
    }
#pragma unroll
    for (int c = 0; c < NC; c++) {
      __builtin_amdgcn_sched_barrier(0);
    if ((act >> c) & 1u) {
        // group-uniform tables come from LDS as broadcast reads (no stores in between: freely scheduled)
        const ConRec& rr = w.rec[c];
        const real Rc = rr.R;
        real u[4], Dk[4] = {0, 0, 0, 0}, f[6];
        {
          // the four basis projections u = J a as INTERLEAVED group sums (round 4; bitwise the same sums as four gsum calls:
          // the products are rounded before the first addition either way)
#pragma clang fp contract(off)
#pragma unroll
          for (int k = 0; k < 4; k++) u[k] = cr.jb[c][k] * a;
          gsum_n<G, 4>(u);
        }
#pragma unroll
        for (int e = 0; e < 6; e++) {
          const int k = e / 2 + 1;
          if (slot_kind<NL>(c) == 2 && e >= 4) { f[e] = 0; continue; }        // condim-3 pair: 4 edges
          const real sm = (e & 1) ? -rr.mu[k - 1] : rr.mu[k - 1];
          const real f0 = rr.f[e];
          const real res = (u[0] + sm * u[k]) + (Rc * f0 - rr.aref[e]);
          const real fn = fmax(f0 - res * rr.inv[e], 0.0);
          const real dlt = fn - f0;
          improvement -= dlt * (res + 0.5 * rr.den[e] * dlt);
          f[e] = fn;
          Dk[0] += dlt; Dk[k] += sm * dlt;
#pragma unroll
          for (int l = 0; l < 4; l++) u[l] += w.p.Ge[c][e][l] * dlt;
        }
        if (sub == 0) {
#pragma unroll
          for (int e = 0; e < 6; e++) w.rec[c].f[e] = f[e];
        }
#pragma unroll
        for (int k = 0; k < 4; k++) a += (c < 4 ? cr.jb[c][k] * invm : cr.bb[c < 4 ? 0 : c - 4][k]) * Dk[k];
      }
    }
    improvement += gsum<G>(imp_local);
    if (improvement * scale < tol) break;
  }
  return a;
}


// =============================================================================================
// Newton solver (MuJoCo's default solver, i.e. what the reference actually runs: no <option> element in
// any of its XML files).  Primal problem over qacc:  1/2 (a-a_s)^T M (a-a_s) + sum_i s_i(J_i a - aref_i).
// Lane d owns a_d, grad_d, the search component p_d and column d of the Hessian; the nv x nv Hessian lives
// in LDS (aliasing the dead kinematics) and is factored by a cooperative Cholesky; projections of the
// contact bases are DPP row reductions; the exact line search evaluates phi', phi'' with the rows strided
// over the lanes.  The minimiser is unique, so parity with the oracle does not depend on iteration counts.

// Dof subset of a Newton problem.  Arm and cube meet only in finger-cube contacts (slot kind 1); while none is active the
// primal cost is a SUM of an arm part (dofs 0..NL-1; rows: arm friction loss / limits, finger-table contacts) and a cube part
// (dofs NL..NV-1; rows: cube friction loss, table-cube contacts), i.e. two independent strictly convex minimisations with
// the same joint minimiser -- solved one after the other, each on its own block of the Hessian.  The hard solves of a batch
// are cube-table impacts (tens of active-set changes): they then cost 6 pivots per iteration instead of NV.
enum { KM_SUB_ALL = 0, KM_SUB_ARM = 1, KM_SUB_CUBE = 2 };
template <int NL, int S> struct SubSet {
  static constexpr int D0 = S == KM_SUB_CUBE ? NL : 0, D1 = S == KM_SUB_ARM ? NL : NL + 6;
  static constexpr int kind(int c) { return slot_kind<NL>(c); }
  static constexpr bool slot(int c) { return S == KM_SUB_ALL || (S == KM_SUB_ARM ? kind(c) == 2 : kind(c) == 0); }
  // columns of the Hessian a slot of this kind touches (its Jacobian is zero elsewhere), intersected with the subset
  static constexpr int c0(int c) { const int k = kind(c); const int lo = k == 0 ? NL : 0; return lo > D0 ? lo : D0; }
  static constexpr int c1(int c) { const int k = kind(c); const int hi = k == 2 ? NL : NL + 6; return hi < D1 ? hi : D1; }
};

// Cholesky of an SPD matrix held one ROW PER LANE in registers (h[j] = H[sub][j]), right-looking, in place, on the
// diagonal block [D0, D1): afterwards h[j] = L[sub][j] for D0 <= j <= sub (the j > sub entries are dead) and
// invd = 1 / L[sub][sub].  Lanes outside the block hold zeros and stay inert.
// Column k of L reaches the other rows through DPP row broadcasts: no LDS, no synchronisation.
template <int G, int N, int D0, int D1>
__device__ __forceinline__ void chol_rows(real (&h)[N], real& invd, int sub, int& bad) {
  static_for<D0, D1>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    real dk = gbcast<G, k>(h[k]);
    if (!(dk > 0)) { bad = 1; dk = 1; }
    const real inv = rsqrt_nr(dk);
    const real lik = h[k] * inv;
    h[k] = lik;
    if (sub == k) invd = inv;
    const BSrc<G> lsrc = bsrc<G>(lik);
    fnmac_cols<G, k + 1, D1>(h, lsrc, lik);
    if constexpr (k + 2 >= D1 && k + 1 < D1) dpp_settle(h[k + 1]);     // the next pivot's broadcast reads what the last run just wrote
  });
}
// x = (L L^T)^-1 b, b distributed one component per lane.  Forward substitution is column-oriented (z_k broadcast,
// rows below updated); the transposed solve uses the dot form (lane i contributes L[i][k] x_i, group sum).
template <int G, int N, int D0, int D1>
__device__ __forceinline__ real chol_solve_rows(const real (&h)[N], real invd, int sub, real b) {
  static_for<D0, D1>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    const real t = b * invd;                         // lane k's t is z_k
    real upd = b;
    fnmac_b<G, k>(upd, bsrc<G>(t), h[k]);
    b = sub > k ? upd : (sub == k ? t : b);
  });
  real x = 0;
  static_for<D0, D1>([&](auto kc) {
    constexpr int k = D1 - 1 - (decltype(kc)::value - D0);
    const real s = gsum<G>(sub > k ? h[k] * x : 0.0);
    if (sub == k) x = (b - s) * invd;
  });
  return x;
}

// ---- One-row systems (round 3): the same right-looking Cholesky, but column k of L is MASKED to its strictly-lower part as
// it is formed (lik = sub > k ? h[k] * inv : 0), so rows on and above the pivot never change again and hold exact zeros there.
// Both triangular solves are then column-oriented -- one multiply and one broadcast-FMA per pivot, no lane tests, no lane
// reductions -- given row `sub` of L^T next to row `sub` of L.  Row `sub` of L^T is column `sub` of L, which lives in the
// OTHER lanes' registers; it arrives either from the factorisation's own broadcasts (UT: ut[j] += bcast_j(l) * [sub == k],
// (D1-D0)(D1-D0-1)/2 extra broadcast-FMAs: small blocks) or through one LDS transposition (chol_transpose: larger blocks).
// Round 2's transposed solve took one 16-lane reduction per pivot (12-20 instructions each).
// `sl` = this lane's dof index relative to the DPP row's first dof (two-row groups run a block that sits in one row with the
// other row inert: sl < 0 or rows of zeros); `live` = the lane's row holds the block (only those lanes report a bad pivot).
template <int N, int D0, int D1, int BASE, bool UT>
__device__ __forceinline__ void chol_rows1(real (&h)[N], real (&ut)[N], real& invd, int sl, bool live, int& bad) {
  invd = 0;
  if constexpr (UT) {
#pragma unroll
    for (int j = 0; j < N; j++) ut[j] = 0;
  }
  static_for<D0, D1>([&](auto kc) {
    constexpr int k = decltype(kc)::value, kl = k - BASE;
    const real dk = gbcast<16, kl>(h[k]);
    bad |= live && !(dk > 0);
    const real inv = rsqrt_nr(dk);
    const bool me = sl == kl;
    const real lik = sl > kl ? h[k] * inv : 0.0;
    h[k] = lik;
    invd = me ? inv : invd;
    if constexpr (UT) {
      const real isk = me ? 1.0 : 0.0;
      static_for<k + 1, D1>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        dppfma_pn<j - BASE, j == k + 1>(ut[j], lik, isk, h[j], lik, lik);      // ut[j] += L[j][k] [sub == k];  h[j] -= L[j][k] L[sub][k]
      });
    } else {
      static_for<k + 1, D1>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        dppfma1<true, j - BASE, j == k + 1>(h[j], lik, lik);
      });
    }
    if constexpr (k + 2 >= D1 && k + 1 < D1) dpp_settle(h[k + 1]);     // the next pivot's broadcast reads what the last run just wrote
  });
}
// ut[k] = L[k][sub] through LDS: lane i writes row i of L (exact zeros on and above the diagonal), lane s reads column s
template <int N, int D0, int D1, int BASE, class LT>
__device__ __forceinline__ void chol_transpose(LT& lt, const real (&h)[N], real (&ut)[N], int sl) {
  const int r = sl < 0 ? 0 : sl;                        // (lanes of an inert row write zeros over zeros)
#pragma unroll
  for (int k = D0; k < D1; k++) lt[r][k - D0] = h[k];
  GSYNC();
  // lanes outside the block read a column of the block too (finite numbers, never stale LDS): their invd = 0 then gives the
  // zero they need without a select per entry
  const int col = r + BASE - D0 < 0 ? 0 : (r + BASE - D0 > D1 - D0 - 1 ? D1 - D0 - 1 : r + BASE - D0);
#pragma unroll
  for (int k = D0; k < D1; k++) ut[k] = lt[k - BASE][col];
  GSYNC();
}
// x = (L L^T)^-1 b, b distributed one component per lane (zero outside the block)
template <int N, int D0, int D1, int BASE>
__device__ __forceinline__ real chol_solve_rows1(const real (&h)[N], const real (&ut)[N], real invd, real b) {
  static_for<D0, D1>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    const real t = b * invd;                            // lane k's t is z_k (its b is final: h[j] = 0 for j >= sub)
    fnmac_bcast16<k - BASE>(b, t, h[k]);
  });
  real z = b * invd;
  static_for<D0, D1>([&](auto kc) {
    constexpr int k = D1 - 1 - (decltype(kc)::value - D0);
    const real t = z * invd;                            // lane k's t is x_k (ut[j] = 0 for j <= sub)
    fnmac_bcast16<k - BASE>(z, t, ut[k]);
  });
  return z * invd;
}

// s_i'(x) and s_i''(x) contributions of one row to the line-search derivatives
__device__ __forceinline__ void row_ls(int type, real x, real y, real R, real Dn, real fl, real& d1, real& d2) {
  if (type == 0) {
    if (x <= -R * fl) d1 += -fl * y;
    else if (x >= R * fl) d1 += fl * y;
    else { d1 += Dn * x * y; d2 += Dn * y * y; }
  } else if (x < 0) { d1 += Dn * x * y; d2 += Dn * y * y; }
}
// cost / force / quadratic-zone flag of one row
__device__ __forceinline__ real row_eval(int type, real x, real R, real Dn, real fl, real& f, int& quad) {
  if (type == 0) {
    if (x <= -R * fl) { f = fl; quad = 0; return fl * (-0.5 * R * fl - x); }
    if (x >= R * fl) { f = -fl; quad = 0; return fl * (-0.5 * R * fl + x); }
    f = -Dn * x; quad = 1; return 0.5 * Dn * x * x;
  }
  if (x < 0) { f = -Dn * x; quad = 1; return 0.5 * Dn * x * x; }
  f = 0; quad = 0; return 0;
}

// The six cube components of a lane-distributed vector, on every lane: linear part and the angular part turned
// into the world frame (the free joint's angular velocity is expressed in the body frame).
template <int NL, int G>
__device__ __forceinline__ void cube_part(const Ws<NL>& w, real x, real* lin, real* angw) {
  lin[0] = gbcast<G, NL>(x); lin[1] = gbcast<G, NL + 1>(x); lin[2] = gbcast<G, NL + 2>(x);
  const real ab[3] = {gbcast<G, NL + 3>(x), gbcast<G, NL + 4>(x), gbcast<G, NL + 5>(x)};
  mat_vec3(angw, w.k.cube_mat, ab);
}
// J_c x for a table-cube contact (slots 0..3: only the cube moves, plane frame): the velocity of the contact point
// read off in the frame -- no cross-lane reduction.  u = (normal, tangent 1, tangent 2, torsion).
template <int NL>
__device__ __forceinline__ void plane_proj(const Ws<NL>& w, int c, const real* lin, const real* angw, real* u) {
  const real r[3] = {w.c_pos[c][0] - w.qpos[NL], w.c_pos[c][1] - w.qpos[NL + 1], w.c_pos[c][2] - w.qpos[NL + 2]};
  real v[3];
  cross3(v, angw, r);
  v[0] += lin[0]; v[1] += lin[1]; v[2] += lin[2];
  u[0] = v[2]; u[1] = v[1]; u[2] = -v[0]; u[3] = angw[2];     // KM_PLANE_FRAME rows
}

// the same for the table-cube slot that lane `sub` owns (lanes 0..3; the others get slot 0's numbers, which they never use):
// ONE evaluation serves all four corner slots
template <int NL>
__device__ __forceinline__ void plane_proj_lane(const Ws<NL>& w, int sub, const real* lin, const real* angw, real* u) {
  plane_proj<NL>(w, sub < 4 ? sub : 0, lin, angw, u);
}

// Constraint assembly for Newton: like build_constraints but no B = M^-1 J^T / Gram tables -- only the
// first-edge diagonal (for MuJoCo's pyramidal regulariser) and the velocity projections (for aref).  The
// single-dof rows (friction loss, joint limits) of dof `sub` are built into this lane's registers: the primal
// cost is a sum over rows, so mj_makeConstraint's row order does not matter here (it does for PGS).
template <int NL, int G>
__device__ __forceinline__ void build_constraints_newton(Ws<NL>& w, const LModel<NL>& lm, const KModelDesc* m, int sub,
                                                         CReg<NL>& cr, real invm) {
  SlotC& sc = cr.sc;
  constexpr int NV = Dim<NL>::NV, NC = Dim<NL>::NC, NSPH = Dim<NL>::NSPH;
  cr.fl = 0; cr.Rf = 1; cr.Df = 1; cr.areff = 0; cr.sg = 0; cr.Rl = 1; cr.Dl = 1; cr.arefl = 0;
  const bool armlane = sub < NL, cubelane = sub >= NL && sub < NV;
  const int jl = armlane ? sub : 0, ce = cubelane ? sub - NL : 0;
  // ---- Round 6: EVERYTHING the assembly reads unconditionally is fetched here, at clamped addresses, in one go (km_pin: one wait
  // instead of one per input -- the phase was ~30 LDS round trips in a row with one wave per SIMD); the conditions select afterwards.
  const int si = sub < NL ? sub : NL - 1, sv = sub < NV ? sub : NV - 1, ck = ce >= 3 ? ce - 3 : 0;
  const int cs = sub < NC ? sub : NC - 1;                       // the contact slot this lane owns (slot lanes)
  real dofw = lm.dofw[si], cubew0 = lm.cubew[0], cubew1 = lm.cubew[1], qvs = w.qvel[sv], kk0 = lm.kb[0][0], bb0 = lm.kb[0][1];
  real floss = lm.floss[si], imp00 = lm.imp0[0], qps = w.qpos[si], rlo = lm.range[si][0], rhi = lm.range[si][1], distc = w.c_dist[cs];
  real ax[3] = {w.k.axis[jl][0], w.k.axis[jl][1], w.k.axis[jl][2]}, xo[3] = {w.k.xpos[jl][0], w.k.xpos[jl][1], w.k.xpos[jl][2]};
  real col0 = w.k.cube_mat[ck], col1 = w.k.cube_mat[3 + ck], col2 = w.k.cube_mat[6 + ck], cpos[3] = {w.qpos[NL], w.qpos[NL + 1], w.qpos[NL + 2]};
  real cm[9], cpc[4][3], cpl[3];                                // cube rotation; the four corner slots' contact points; this lane's corner
#pragma unroll
  for (int k = 0; k < 9; k++) cm[k] = w.k.cube_mat[k];
#pragma unroll
  for (int c = 0; c < 4; c++) { cpc[c][0] = w.c_pos[c][0]; cpc[c][1] = w.c_pos[c][1]; cpc[c][2] = w.c_pos[c][2]; }
  { const int cl = sub < 4 ? sub : 0; cpl[0] = w.c_pos[cl][0]; cpl[1] = w.c_pos[cl][1]; cpl[2] = w.c_pos[cl][2]; }
  int jt = lm.jtype[jl], sps = w.slot_sph[cs];
  uint32_t act = w.cact;
  km_pin(dofw, cubew0, cubew1, qvs, kk0, bb0); km_pin(floss, imp00, qps, rlo, rhi, distc);
  km_pin(ax, xo); km_pin(col0, col1, col2); km_pin(cpos); km_pin(cm); km_pin(cpc[0], cpc[1]); km_pin(cpc[2], cpc[3]); km_pin(cpl);
  km_pin_i(jt, sps); asm volatile("" : "+v"(act));
  if (sub < NV) {
    const real Ad = sub < NL ? dofw : (sub < NL + 3 ? cubew0 : cubew1);           // efc_diagApprox (qpos0 constants)
    const real qv = qvs;
    const real kk = kk0, bb = bb0;
    const real fl = sub < NL ? floss : m->cube_frictionloss;
    if (fl > 0) {
      const real imp = imp00;
      cr.fl = fl; cr.Rf = fmax(MJ_MINVAL, (1 - imp) * frcp(imp) * Ad); cr.Df = frcp(cr.Rf); cr.areff = -bb * qv;
    }
    if (sub < NL) {
      const real dl = qps - rlo, du = rhi - qps;
      const real pos = dl < 0 ? dl : du;
      if (pos < 0) {                                       // (lower and upper cannot both be violated: range lo < hi)
        const real imp = impedance_c(lm.imp[0], pos);
        cr.sg = dl < 0 ? 1.0 : -1.0;
        cr.Rl = fmax(MJ_MINVAL, (1 - imp) * frcp(imp) * Ad);
        cr.Dl = frcp(cr.Rl);
        cr.arefl = -bb * (cr.sg * qv) - kk * imp * pos;
      }
    }
  }
  // ---- this lane's column of every active contact's Jacobian basis.  What a dof does to a point depends on the dof only
  // through a direction A and, for rotations, a point O on the axis (an arm hinge: joint axis and origin; an arm slider: its
  // axis; the cube: a world axis, or a body axis through the cube centre) -- fetched ONCE, unconditionally, above;
  // per slot the column is then a cross product and selects, no branches and no loads under conditions (sphere slots, which are
  // rarely active, fetch their contact point, frame and ancestor mask together inside their branch).
  const bool rot = armlane ? jt != KM_JNT_SLIDE : ce >= 3;
  real A[3], O[3];
#pragma unroll
  for (int d = 0; d < 3; d++) {
    const real cold = d == 0 ? col0 : (d == 1 ? col1 : col2);
    A[d] = armlane ? ax[d] : (ce >= 3 ? cold : (ce == d ? 1.0 : 0.0));
    O[d] = armlane ? xo[d] : cpos[d];
  }
#pragma unroll
  for (int c = 0; c < NC; c++) {
    cr.jb[c][0] = 0; cr.jb[c][1] = 0; cr.jb[c][2] = 0; cr.jb[c][3] = 0;
    if ((act >> c) & 1u) {                                 // (group-uniform)
      const int kind = slot_kind<NL>(c);
      // geom1 / geom2: kind 0 table (world) / cube, kind 1 sphere's link / cube, kind 2 table (world) / sphere's link.  A lane
      // belongs to at most one of the two bodies; its column is +J for geom2's body, -J for geom1's.
      real sgn = 0, cp[3], fr[9];
      if (kind == 0) {
        sgn = cubelane ? 1.0 : 0.0;
        cp[0] = cpc[c < 4 ? c : 0][0]; cp[1] = cpc[c < 4 ? c : 0][1]; cp[2] = cpc[c < 4 ? c : 0][2];
      } else {
        cp[0] = w.c_pos[c][0]; cp[1] = w.c_pos[c][1]; cp[2] = w.c_pos[c][2];
#pragma unroll
        for (int k = 0; k < 9; k++) fr[k] = w.c_frame[c][k];
        uint32_t am = w.slot_anc[c];
        km_pin(cp, fr); asm volatile("" : "+v"(am));
        const bool mine = armlane && ((am >> jl) & 1u);
        sgn = kind == 1 ? (mine ? -1.0 : (cubelane ? 1.0 : 0.0)) : (mine ? 1.0 : 0.0);
      }
      const real r[3] = {cp[0] - O[0], cp[1] - O[1], cp[2] - O[2]};
      real jp[3];
      cross3(jp, A, r);
#pragma unroll
      for (int d = 0; d < 3; d++) jp[d] = sgn * (rot ? jp[d] : A[d]);
      const real jr[3] = {rot ? sgn * A[0] : 0.0, rot ? sgn * A[1] : 0.0, rot ? sgn * A[2] : 0.0};
      if (kind == 0) {                                     // constant plane frame: rows n = +z, t1 = +y, t2 = -x
        cr.jb[c][0] = jp[2]; cr.jb[c][1] = jp[1]; cr.jb[c][2] = -jp[0]; cr.jb[c][3] = jr[2];
      } else {
        cr.jb[c][0] = dot3(fr, jp);
        cr.jb[c][1] = dot3(fr + 3, jp);
        cr.jb[c][2] = dot3(fr + 6, jp);
        cr.jb[c][3] = dot3(fr, jr);
      }
    }
  }
  // the slot lanes' solver constants: fetched now, while the projections below run (slot `sub` of a slot lane; clamped elsewhere)
  const int kindl = cs < 4 ? 0 : (cs < 4 + Dim<NL>::NSS ? 1 : 2), pset = kindl != 2 ? 1 : 0;
  const int spc = sps < 0 ? 0 : (sps >= NSPH ? NSPH - 1 : sps);            // (an inactive slot's sphere index is stale: clamped, never used)
  real sA = kindl == 0 ? lm.cornerA : lm.sphA[kindl == 2][spc];            // efc_diagApprox of the first pyramid edge (qpos0 constant; no M^-1 product)
  real mu_t = lm.fric[pset][0], mu_r = lm.fric[pset][1], kks = lm.kb[pset][0], bbs = lm.kb[pset][1];
  real i_d0 = lm.imp[pset].d0, i_dw = lm.imp[pset].dw, i_iw = lm.imp[pset].iw, i_mid = lm.imp[pset].mid, i_imid = lm.imp[pset].imid, i_i1 = lm.imp[pset].i1mid;
  int i_mode = lm.imp[pset].mode;
  const real qv = sub < NV ? qvs : 0.0;
  // the six cube components of qvel on every lane, the angular part in the world frame (cube_part with the rotation fetched above)
  real qlin[3], qangw[3];
  {
    qlin[0] = gbcast<G, NL>(qv); qlin[1] = gbcast<G, NL + 1>(qv); qlin[2] = gbcast<G, NL + 2>(qv);
    const real ab[3] = {gbcast<G, NL + 3>(qv), gbcast<G, NL + 4>(qv), gbcast<G, NL + 5>(qv)};
    mat_vec3(qangw, cm, ab);
  }
  // velocity projections of every active slot; lane c keeps slot c's
  real vb[4] = {0, 0, 0, 0};
#pragma unroll
  for (int k = 0; k < 9; k++) cr.cm[k] = cm[k];
  cr.pr[0] = cpl[0] - cpos[0]; cr.pr[1] = cpl[1] - cpos[1]; cr.pr[2] = cpl[2] - cpos[2];
  {                                                             // table-cube slots: lane c < 4 evaluates ITS corner (plane_proj)
    const real r[3] = {cr.pr[0], cr.pr[1], cr.pr[2]};
    real v[3];
    cross3(v, qangw, r);
    v[0] += qlin[0]; v[1] += qlin[1]; v[2] += qlin[2];
    vb[0] = v[2]; vb[1] = v[1]; vb[2] = -v[0]; vb[3] = qangw[2];     // KM_PLANE_FRAME rows
  }
  static_for<4, NC>([&](auto cc) {
    constexpr int c = decltype(cc)::value;
    if ((act >> c) & 1u) {
      constexpr int NK = slot_kind<NL>(c) == 2 ? 3 : 4;
      real pj[NK];
#pragma unroll
      for (int k = 0; k < NK; k++) pj[k] = cr.jb[c][k] * qv;
      gsum_n<G, NK>(pj);
#pragma unroll
      for (int k = 0; k < NK; k++) vb[k] = sub == c ? pj[k] : vb[k];
      if (slot_kind<NL>(c) == 2) vb[3] = sub == c ? 0.0 : vb[3];
    }
  });
  // the solver constants of slot `sub`, one slot per lane (all slots through ONE pass of the impedance / regulariser / reference
  // acceleration arithmetic instead of one unrolled copy per slot)
  km_pin(sA, mu_t, mu_r, kks, bbs, i_d0); km_pin(i_dw, i_iw, i_mid, i_imid, i_i1); km_pin_i(i_mode);
  sc.D = 0; sc.D3 = 0; sc.mu = 0; sc.mu3 = 0; sc.A[0] = 0; sc.A[1] = 0; sc.A[2] = 0; sc.A[3] = 0;
  if (sub < NC && ((act >> sub) & 1u)) {
    const int kind = kindl;
    const real Ad = sA;
    const real dist = distc;
    const real imp = impedance_v(i_d0, i_dw, i_iw, i_mid, i_imid, i_i1, i_mode, dist), kk = kks, bb = bbs;
    const real R = 2 * mu_t * mu_t * fmax(MJ_MINVAL, (1 - imp) * frcp(imp) * Ad), Dn = frcp(R);
    sc.D = Dn; sc.D3 = kind == 2 ? 0.0 : Dn; sc.mu = mu_t; sc.mu3 = mu_r;
    sc.A[0] = -bb * vb[0] - kk * imp * dist; sc.A[1] = -bb * vb[1]; sc.A[2] = -bb * vb[2]; sc.A[3] = -bb * vb[3];
  }
  GSYNC();
}

// M x for a vector distributed one component per lane: arm block from the lane's register row of M
// (components arrive by DPP row broadcast), cube block diagonal
template <int NL, int G>
__device__ __forceinline__ real mass_mul(const CReg<NL>& cr, int sub, real mdiag, real x) {
  real s = 0;
  const BSrc<G> xs = bsrc<G>(x);
  fmac_rowvec<G, 0, NL>(s, xs, [&](int j) { return cr.mrow[j]; });
  return sub < NL ? s : mdiag * x;
}

// =============================================================================================
// Slot-lane Newton (round 3; one- and two-row groups).  Lane c < NC of the group's FIRST DPP row owns contact slot c; what the
// other lanes need from it arrives as a row broadcast inside an FMA (two-row groups: of the copy v_permlane16_swap makes of the
// first row's registers, ONE swap pair per broadcast value whatever the number of slots).  Same mathematics and iterates as
// round 2's edge-distributed layout; the oracle mirrors neither, only the algorithm.

// All six pyramid edges of the slot this lane owns at the shifted projections X (x_e = X_0 +- mu_k X_k): the slot's cost, the
// force it applies along its four basis rows (F = sum_e f_e (1, +-mu_k)), and the Hessian weights of its active edges
// W = sum_{x_e < 0} D_e (1, +-mu_k)(1, +-mu_k)^T, stored (W00, W01, W02, W03, W11, W22, W33).  Branch-free; an inactive slot
// (D = D3 = 0) yields zeros.
// a register of the group's first DPP row as seen from both rows (one-row groups: itself)
template <int G> __device__ __forceinline__ real row0(real x) {
  if constexpr (G == 32) return bsrc<32>(x).e; else return x;
}

template <bool WEIGHTS>
__device__ __forceinline__ real slot_eval(const SlotC& sc, const real (&X)[4], real (&F)[4], real (&W)[7]) {
  const real t1 = sc.mu * X[1], t2 = sc.mu * X[2], t3 = sc.mu3 * X[3];
  const real x1p = X[0] + t1, x1m = X[0] - t1, x2p = X[0] + t2, x2m = X[0] - t2, x3p = X[0] + t3, x3m = X[0] - t3;
  const real m1p = fmin(x1p, 0.0), m1m = fmin(x1m, 0.0), m2p = fmin(x2p, 0.0), m2m = fmin(x2m, 0.0), m3p = fmin(x3p, 0.0), m3m = fmin(x3m, 0.0);
  const real s12 = (m1p + m1m) + (m2p + m2m), s3 = m3p + m3m;
  F[0] = -(sc.D * s12 + sc.D3 * s3);                         // f_e = -D x_e on the active edges
  F[1] = -(sc.mu * sc.D) * (m1p - m1m); F[2] = -(sc.mu * sc.D) * (m2p - m2m); F[3] = -(sc.mu3 * sc.D3) * (m3p - m3m);
  if constexpr (WEIGHTS) {
    const real d1p = x1p < 0 ? sc.D : 0.0, d1m = x1m < 0 ? sc.D : 0.0, d2p = x2p < 0 ? sc.D : 0.0, d2m = x2m < 0 ? sc.D : 0.0;
    const real d3p = x3p < 0 ? sc.D3 : 0.0, d3m = x3m < 0 ? sc.D3 : 0.0;
    W[0] = ((d1p + d1m) + (d2p + d2m)) + (d3p + d3m);
    W[1] = sc.mu * (d1p - d1m); W[2] = sc.mu * (d2p - d2m); W[3] = sc.mu3 * (d3p - d3m);
    W[4] = (sc.mu * sc.mu) * (d1p + d1m); W[5] = (sc.mu * sc.mu) * (d2p + d2m); W[6] = (sc.mu3 * sc.mu3) * (d3p + d3m);
  }
  return 0.5 * (sc.D * ((m1p * m1p + m1m * m1m) + (m2p * m2p + m2m * m2m)) + sc.D3 * (m3p * m3p + m3m * m3m));
}
// which of the slot's six edges are active at the shifted projections X (the comparisons slot_eval's weights W come from)
__device__ __forceinline__ int slot_edge_mask(const SlotC& sc, const real (&X)[4]) {
  const real t1 = sc.mu * X[1], t2 = sc.mu * X[2], t3 = sc.mu3 * X[3];
  return (int)(X[0] + t1 < 0) | (int)(X[0] - t1 < 0) << 1 | (int)(X[0] + t2 < 0) << 2 | (int)(X[0] - t2 < 0) << 3
         | (int)(X[0] + t3 < 0) << 4 | (int)(X[0] - t3 < 0) << 5;
}
// this slot's contribution to phi'(alpha) and phi''(alpha) along y (X already holds u + alpha y)
__device__ __forceinline__ void slot_ls(const SlotC& sc, const real (&X)[4], const real (&y)[4], real& e1, real& e2) {
  const real t1 = sc.mu * X[1], t2 = sc.mu * X[2], t3 = sc.mu3 * X[3], s1 = sc.mu * y[1], s2 = sc.mu * y[2], s3 = sc.mu3 * y[3];
  const real x1p = X[0] + t1, x1m = X[0] - t1, x2p = X[0] + t2, x2m = X[0] - t2, x3p = X[0] + t3, x3m = X[0] - t3;
  const real y1p = y[0] + s1, y1m = y[0] - s1, y2p = y[0] + s2, y2m = y[0] - s2, y3p = y[0] + s3, y3m = y[0] - s3;
  const real a = (fmin(x1p, 0.0) * y1p + fmin(x1m, 0.0) * y1m) + (fmin(x2p, 0.0) * y2p + fmin(x2m, 0.0) * y2m);
  const real a3 = fmin(x3p, 0.0) * y3p + fmin(x3m, 0.0) * y3m;
  e1 += sc.D * a + sc.D3 * a3;
  const real b = ((x1p < 0 ? y1p * y1p : 0.0) + (x1m < 0 ? y1m * y1m : 0.0)) + ((x2p < 0 ? y2p * y2p : 0.0) + (x2m < 0 ? y2m * y2m : 0.0));
  const real b3 = (x3p < 0 ? y3p * y3p : 0.0) + (x3m < 0 ? y3m * y3m : 0.0);
  e2 += sc.D * b + sc.D3 * b3;
}

// J_c v of every active slot of the subset, delivered to the lane that owns the slot (u of the other lanes / slots: finite
// numbers that meet D = 0).  Table-cube slots: lane c < 4 reads the contact point's velocity off the cube twist (one
// evaluation for all four); sphere slots: a group sum per basis row, kept by lane c.
template <int NL, int G, int S>
__device__ __forceinline__ void slot_project(const Ws<NL>& w, const CReg<NL>& cr, uint32_t act, int sub, real v, real (&u)[4]) {
  constexpr int NC = Dim<NL>::NC;
  using SS = SubSet<NL, S>;
  u[0] = 0; u[1] = 0; u[2] = 0; u[3] = 0;
  if constexpr (S != KM_SUB_ARM) {
    // cube_part + plane_proj_lane on the registers the constraint assembly left in cr (same operations, same values)
    const real lin[3] = {gbcast<G, NL>(v), gbcast<G, NL + 1>(v), gbcast<G, NL + 2>(v)};
    const real ab[3] = {gbcast<G, NL + 3>(v), gbcast<G, NL + 4>(v), gbcast<G, NL + 5>(v)};
    real angw[3], vv[3];
    mat_vec3(angw, cr.cm, ab);
    cross3(vv, angw, cr.pr);
    vv[0] += lin[0]; vv[1] += lin[1]; vv[2] += lin[2];
    u[0] = vv[2]; u[1] = vv[1]; u[2] = -vv[0]; u[3] = angw[2];     // KM_PLANE_FRAME rows
  }
  static_for<4, NC>([&](auto cc) {
    constexpr int c = decltype(cc)::value;
    if constexpr (SS::slot(c)) {
      if ((act >> c) & 1u) {
        constexpr int NK = slot_kind<NL>(c) == 2 ? 3 : 4;
        real pj[NK];
#pragma unroll
        for (int k = 0; k < NK; k++) pj[k] = cr.jb[c][k] * v;
        gsum_n<G, NK>(pj);
#pragma unroll
        for (int k = 0; k < NK; k++) u[k] = sub == c ? pj[k] : u[k];
        if (slot_kind<NL>(c) == 2) u[3] = sub == c ? 0.0 : u[3];
      }
    }
  });
}
// grad -= J^T F over the subset's active slots: the four force components of slot c arrive from lane c inside the FMAs
template <int NL, int G, int S>
__device__ __forceinline__ void slot_grad(const CReg<NL>& cr, uint32_t act, const real (&F)[4], real& grad) {
  constexpr int NC = Dim<NL>::NC;
  static_assert(NC <= 16, "the slot lanes sit in the group's first DPP row");
  using SS = SubSet<NL, S>;
  const real F0 = row0<G>(F[0]), F1 = row0<G>(F[1]), F2 = row0<G>(F[2]), F3 = row0<G>(F[3]);
  static_for<0, NC>([&](auto cc) {
    constexpr int c = decltype(cc)::value;
    if constexpr (SS::slot(c)) {
      if ((act >> c) & 1u) {
        real g2 = 0;
        if constexpr (slot_kind<NL>(c) == 2) dppfma_acc3<c>(g2, F0, cr.jb[c][0], F1, cr.jb[c][1], F2, cr.jb[c][2]);
        else dppfma_acc4<c>(g2, F0, cr.jb[c][0], F1, cr.jb[c][1], F2, cr.jb[c][2], F3, cr.jb[c][3]);
        grad -= g2;
      }
    }
  });
}
// does lane `sub` own a slot of the subset?
template <int NL, int S> __device__ __forceinline__ bool slot_lane_in(int sub) {
  constexpr int NC = Dim<NL>::NC, NSS = Dim<NL>::NSS;
  return S == KM_SUB_ALL ? sub < NC : (S == KM_SUB_ARM ? (sub >= 4 + NSS && sub < NC) : sub < 4);
}

// Newton state at a start point (all slots, both cost parts): u = J a - A on the slot lanes, gradient, the lanes' own rows,
// the slots' Hessian weights.  cs != nullptr: also this lane's share of the cost at a_s (MuJoCo's warm-start comparison).
template <int NL, int G, bool CS>
__device__ __forceinline__ void newton_eval_sl(const Ws<NL>& w, int sub, const CReg<NL>& cr, real a, real a_s, real Mr, real& grad, int& qf,
                                               int& ql, real (&u)[4], real (&W)[7], real& c0, real& c1, real& cs) {
  constexpr int NV = Dim<NL>::NV;
  const uint32_t act = w.cact;
  const SlotC& sc = cr.sc;
  slot_project<NL, G, KM_SUB_ALL>(w, cr, act, sub, a, u);
#pragma unroll
  for (int k = 0; k < 4; k++) u[k] -= sc.A[k];
  real F[4];
  const real cslot = slot_eval<true>(sc, u, F, W);
  real csl = 0;
  if constexpr (CS) {
    real us[4], Fs[4], Ws_[7];
    slot_project<NL, G, KM_SUB_ALL>(w, cr, act, sub, a_s, us);
#pragma unroll
    for (int k = 0; k < 4; k++) us[k] -= sc.A[k];
    csl = slot_eval<false>(sc, us, Fs, Ws_);
  }
  grad = Mr;
  qf = 0; ql = 0;
  {
    real co = 0.5 * (a - a_s) * Mr;
    if (cr.fl > 0) { real f; co += row_eval(0, a - cr.areff, cr.Rf, cr.Df, cr.fl, f, qf); grad -= f; }
    if (cr.sg != 0) { real f; co += row_eval(1, cr.sg * a - cr.arefl, cr.Rl, cr.Dl, 0.0, f, ql); grad -= cr.sg * f; }
    // (selects, not `if (..) c0 = ..; else c1 = ..`: the latter made the compiler index {c0, c1} in scratch memory)
    c0 = sub < NL ? co : 0.0; c1 = (sub >= NL && sub < NV) ? co : 0.0;
    if constexpr (CS) {                         // (the Gauss term vanishes at a_s)
      real f; int qd;
      if (cr.fl > 0) csl += row_eval(0, a_s - cr.areff, cr.Rf, cr.Df, cr.fl, f, qd);
      if (cr.sg != 0) csl += row_eval(1, cr.sg * a_s - cr.arefl, cr.Rl, cr.Dl, 0.0, f, qd);
    }
  }
  c1 += sub < 4 ? cslot : 0.0; c0 += sub < 4 ? 0.0 : cslot;   // table-cube slots belong to the cube part (lanes without a slot: cslot = 0)
  slot_grad<NL, G, KM_SUB_ALL>(cr, act, F, grad);
  cs = csl;
}

// Hessian row `sub` (block [D0, D1) of the subset) from the slots' weights: H += J_c^T W_c J_c, the weights of slot c arriving
// from lane c inside the FMAs that build t = W_c J_c[:, sub]
// CUBECOLS: only the cube's columns NL..NV-1 of every row (the partial refactorisation of newton_loop_sl; the entries are built by
// the same operations in the same order as in the full build, so they come out bitwise the same)
template <int NL, int G, int S, bool CUBECOLS = false>
__device__ __forceinline__ void newton_hessian_sl(const Ws<NL>& w, int sub, const CReg<NL>& cr, real mdiag, int qf, int ql,
                                                  const real (&W)[7], real (&h)[Dim<NL>::NV], bool in, uint32_t act, bool joint) {
  constexpr int NV = Dim<NL>::NV, NC = Dim<NL>::NC, J0 = CUBECOLS ? NL : 0;
  using SS = SubSet<NL, S>;
  {
    real dg = sub < NL ? 0.0 : mdiag;
    if (qf) dg += cr.Df;
    if (ql) dg += cr.Dl;
    // dofs outside the problem: zero rows -- or, in the joint loop (whose pivots run over them too), identity rows
    const real idg = joint ? 1.0 : 0.0;
#pragma unroll
    for (int j = J0; j < NV; j++) h[j] = in ? (j < NL ? cr.mrow[j] : 0.0) + ((j == sub) ? dg : 0.0) : ((j == sub) ? idg : 0.0);
  }
  real Wb[7];
#pragma unroll
  for (int i = 0; i < 7; i++) Wb[i] = row0<G>(W[i]);
  static_for<0, NC>([&](auto cc) {
    constexpr int c = decltype(cc)::value;
    if constexpr (SS::slot(c)) {
      if ((act >> c) & 1u) {
        const real j0 = cr.jb[c][0], j1 = cr.jb[c][1], j2 = cr.jb[c][2], j3 = cr.jb[c][3];
        real t0 = 0, t1 = 0, t2 = 0, t3 = 0;
        if constexpr (SS::kind(c) != 2) {
          dppfma_acc4<c>(t0, Wb[0], j0, Wb[1], j1, Wb[2], j2, Wb[3], j3);
          dppfma3<false, c, c, c>(t1, Wb[1], j0, t2, Wb[2], j0, t3, Wb[3], j0);
          dppfma3<false, c, c, c, false>(t1, Wb[4], j1, t2, Wb[5], j2, t3, Wb[6], j3);
        } else {                                                                // (condim-3 pairs have no torsion row)
          dppfma_acc3<c>(t0, Wb[0], j0, Wb[1], j1, Wb[2], j2);
          dppfma2<false, c, c>(t1, Wb[1], j0, t2, Wb[2], j0);
          dppfma2<false, c, c, false>(t1, Wb[4], j1, t2, Wb[5], j2);
        }
        // H[sub][j] += sum_k t_k(sub) * J_k[j]: lane j's basis entries arrive by row broadcast (only the columns the slot's
        // Jacobian can be nonzero in); DPP sources = the Jacobian columns (or their row copies), written long before: the
        // first run of a two-row group still waits for the swap that made the copies
        const BSrc<G> j0s = bsrc<G>(j0), j1s = bsrc<G>(j1), j2s = bsrc<G>(j2), j3s = bsrc<G>(j3);
        static_for<(SS::c0(c) > J0 ? SS::c0(c) : J0), SS::c1(c)>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          constexpr bool WT = G == 32 && j == SS::c0(c);
          if constexpr (SS::kind(c) != 2) dppfma_acc4<j & 15, WT>(h[j], bsel<G, j>(j0s), t0, bsel<G, j>(j1s), t1, bsel<G, j>(j2s), t2, bsel<G, j>(j3s), t3);
          else dppfma_acc3<j & 15, WT>(h[j], bsel<G, j>(j0s), t0, bsel<G, j>(j1s), t1, bsel<G, j>(j2s), t2);
        });
      }
    }
  });
}

// Newton iterations on one dof subset, from the point (a, Mr, grad, qf, ql, u, W) with cost `cost` (all of the subset).
// JOINT (S = KM_SUB_ALL only): the wave holds at least one coupled env.  Its uncoupled wave-mates would otherwise run their arm
// loops BEFORE and their cube loops AFTER the coupled env's 16-dof loop (different code paths: SIMD divergence serialises them --
// 0.1-0.2 M and 0.13-0.34 M clocks on top of the slowest waves of a launch); here every group runs ITS problems inside one
// instruction stream: the coupled env its whole problem, an uncoupled env (`two`) first its arm problem (cost `cost`), then its
// cube problem (`cost_b`), each as a 16-dof problem whose other dofs are inert (identity rows, zero gradient, no slots), with its
// own iteration counts; the arm problem's Woodbury direction is the one part that stays a branch of its own.  The inert pivots and
// the zero entries they meet change nothing in a block's arithmetic: an env's result does not depend on what its wave-mates
// are (tests compare shards and launch shapes bit for bit).
template <int NL, int G, int S, bool JOINT = false>
__device__ __forceinline__ void newton_loop_sl(Ws<NL>& w, const LModel<NL>& lm, const KModelDesc* m, int sub, const CReg<NL>& cr,
                                               real mdiag, real a_s, real& a, real& Mr, real cost, real& grad, int& qf, int& ql,
                                               real (&u)[4], real (&W)[7], Prof& pf, bool two = false, real cost_b = 0, int iter0 = 0,
                                               int* resume = nullptr, real* rcost = nullptr, int* riter = nullptr) {
  static_assert(!JOINT || (S == KM_SUB_ALL && G == 16), "the joint loop is the whole-problem loop of the one-row groups");
  constexpr int NV = Dim<NL>::NV, NC = Dim<NL>::NC, NSS = Dim<NL>::NSS;
  using SS = SubSet<NL, S>;
  const SlotC& sc = cr.sc;
  // the problem this group is on (JOINT: run-time and per group), its slots, its dofs, the slot lanes that belong to it (their
  // cost counts, their u moves)
  int prob = (JOINT && two) ? (int)KM_SUB_ARM : S;
  uint32_t act = w.cact;
  bool in = sub >= SS::D0 && sub < SS::D1, slin = slot_lane_in<NL, S>(sub);
  // Partial refactorisation (round 4; one-row groups, whole-problem / joint loop).  Most iterations of a coupled env only move
  // edges of the cube's table contacts (the stiff ones): rows and columns of the ARM dofs -- the first NL pivots -- are then
  // exactly what the previous iteration factorised.  When no group of the wave has changed anything on its arm side (single-
  // dof rows of arm dofs, edge sets of the sphere slots) since the factor that sits in LDS (w.LT) was made, the iteration keeps
  // L's first NL columns, REPLAYS their updates on the cube block (the same FMAs on the same numbers in the same order as the
  // full factorisation, minus the pivots' reciprocal-square-root chains) and factorises only the cube's 6 x 6 Schur complement:
  // bitwise the result of the full path, so an env's bits still do not depend on its wave-mates -- whose state decides which
  // path the wave takes.  `sig0` = the arm-side signature of the cached factor, `invd_keep` its 1 / L_ii.
  int sig0 = 0;
  bool cache_ok = false;
  real invd_keep = 0;
  auto enter = [&](int pr) {
    constexpr uint32_t ARM_SLOTS = ((1u << NC) - 1u) & ~((1u << (4 + NSS)) - 1u);
    cache_ok = false;
    if constexpr (JOINT) {
      if (pr == KM_SUB_CUBE) {
        // a wave-mate's cube problem has identity rows on the arm dofs: the arm columns of ITS factor are known without a
        // factorisation (strictly-lower entries 0, 1 / L_ii = 1 -- and whatever 1 / L_ii a full pass would compute there only ever
        // multiplies the zero arm components of its right-hand side), so its first iteration need not force the wave onto the full path
#pragma unroll
        for (int k = 0; k < NL; k++) w.LT[sub][k] = 0;
        invd_keep = sub < NL ? 1.0 : 0.0; sig0 = 0; cache_ok = true;
      }
    }
    prob = pr;
    act = pr == KM_SUB_ARM ? (w.cact & ARM_SLOTS) : (w.cact & 0xFu);
    in = pr == KM_SUB_ARM ? sub < NL : (sub >= NL && sub < NV);
    slin = pr == KM_SUB_ARM ? (sub >= 4 + NSS && sub < NC) : sub < 4;
  };
  if (JOINT && two) enter(KM_SUB_ARM);
  const real scale = lm.scale;
  const real tol = m->solver_tolerance;
  const int maxit = m->solver_iterations;
  pf.ph(40);       // (what a group waited for wave-mates that ran a loop it does not -- SIMD divergence -- lands here)
  auto small = [&]() { const real g0 = in ? grad : 0.0; return km_sqrt(gsum<G>(g0 * g0)) * scale < tol; };
  if (small()) {
    if (!(JOINT && prob == KM_SUB_ARM)) return;
    enter(KM_SUB_CUBE); cost = cost_b;
    if (small()) return;
  }
  for (int iter = iter0; ; iter++) {
    if constexpr (JOINT) {
      // Round 5: the joint loop runs only while a COUPLED env of the wave is still iterating.  Its uncoupled mates ride along for
      // free until then; what is left of their problems afterwards (measured: a mate's arm + cube iterations in sequence outlast the
      // coupled env's by about one iteration per sub-step, at the whole-problem iteration's price) they finish in their own arm / cube
      // loops -- a third of the cost per iteration, and bit for bit the same iterates: an uncoupled env's arithmetic in here IS that
      // of its own loops (which is what keeps an env's bits independent of its wave-mates), so where an iteration runs changes nothing.
      // The hand-over carries the problem the group is on, its cost so far and its iteration count.
      if (!__any(!two)) { *resume = prob; *rcost = cost; *riter = iter; return; }
    }
#ifdef KM_PROFILE
    const bool lone_it = JOINT && __popcll(__ballot(1)) <= 16;       // this group iterates alone: its wave-mates have left the loop
    if constexpr (JOINT) pf.it_begin();
#endif
    real p = 0;
    // The arm problem's quadratic rows are usually just single-dof rows (the two slider friction-loss rows; now and then a
    // joint at its limit) -- no sphere on the table.  Its Hessian is then M + diag(delta) with at most two nonzero deltas, and
    // this sub-step already holds M^-1: by the Woodbury identity  p = -(y - M^-1[:,S] z),  y = M^-1 grad,
    // (diag(1/delta_S) + M^-1[S,S]) z = y_S  -- one row-times-vector product and a 2 x 2 solve instead of a 10-pivot
    // factorisation and two triangular solves.
    bool plain = false;
    uint32_t rows = 0;
    if (S == KM_SUB_ARM || (JOINT && prob == KM_SUB_ARM)) {
      const unsigned long long bq = __ballot(slin && W[0] != 0);              // a sphere-table slot with edges in their quadratic zone
      const unsigned long long bal = __ballot(in && (qf | ql));
      const int sh = (threadIdx.x & 63) - sub;
      constexpr uint32_t GM = G == 32 ? 0xFFFFFFFFu : 0xFFFFu;
      const bool cq = ((uint32_t)(bq >> sh) & GM) != 0;
      rows = (uint32_t)(bal >> sh) & GM;
      if constexpr (G == 32) {
        // two-arm models: M^-1 is block diagonal, so the identity holds per block -- up to two quadratic rows in EACH block,
        // every lane correcting with the rows of its own block
        const uint32_t lowm = lm.split ? (1u << lm.split) - 1u : 0xFFFFFFFFu;
        plain = !cq && __popc(rows & lowm) <= 2 && __popc(rows & ~lowm) <= 2;
        rows &= (sub < lm.split || !lm.split) ? lowm : ~lowm;
      } else plain = !cq && __popc(rows) <= 2;
    }
    if constexpr (KM_WORK_COUNTERS(NL)) {
      if (sub == 0) w.work += plain ? KM_WORK_PLAIN : (prob == KM_SUB_ALL ? KM_WORK_ALL : (prob == KM_SUB_ARM ? KM_WORK_ARM : KM_WORK_CUBE));
    }
    if (plain) {
      // everything the direction reads from LDS or from other lanes that does not depend on y is requested FIRST and together --
      // the row of M^-1, the 2 x 2 system's entries, the correction's two column entries, the two rows' weights -- so that the
      // path waits for one LDS round trip here and one more for y's two entries, not for eleven in a row
      const BSrc<G> gs = bsrc<G>(in ? grad : 0.0);
      const int row = sub < NL ? sub : 0;
      real mi[NL];
#pragma unroll
      for (int j = 0; j < NL; j++) mi[j] = w.Minv[row][j];
      const int i1 = rows ? __ffs(rows) - 1 : 0, i2 = (rows & (rows - 1)) ? __ffs(rows & (rows - 1)) - 1 : i1;
      const real m11 = w.Minv[i1][i1], m22 = w.Minv[i2][i2], a12 = w.Minv[i1][i2], r1 = w.Minv[row][i1], r2 = w.Minv[row][i2];
      const real dl = (qf ? cr.Df : 0.0) + (ql ? cr.Dl : 0.0);
      const real d1 = __shfl(dl, i1, G), d2 = __shfl(dl, i2, G);
      real y = 0;
      fmac_rowvec<G, 0, NL>(y, gs, [&](int j) { return mi[j]; });
      real corr = 0;
      if (rows) {
        const real y1 = __shfl(y, i1, G), y2 = __shfl(y, i2, G);
        const real a11 = frcp(d1) + m11;
        real z1, z2 = 0;
        if (i2 == i1) z1 = y1 * frcp(a11);
        else {
          const real a22 = frcp(d2) + m22;
          const real idet = frcp(a11 * a22 - a12 * a12);
          z1 = (a22 * y1 - a12 * y2) * idet;
          z2 = (a11 * y2 - a12 * y1) * idet;
        }
        corr = r1 * z1 + (i2 == i1 ? 0.0 : r2 * z2);
      }
      p = in ? -(y - corr) : 0.0;
      pf.ph(11 + 6 * S);
    } else {
      real h[NV];
      bool partial = false;
      int sig = 0;
      if constexpr (S == KM_SUB_ALL && G == 16) {
        // arm-side signature of this iteration's Hessian: quadratic-zone flags of the arm dofs' own rows, edge sets of the sphere slots
        // (an inactive slot's projections are arbitrary finite numbers: not part of the signature; a condim-3 pair has no torsion edges)
        sig = (in && sub < NL ? (qf | ql << 1) : 0)
              | ((slin && sub >= 4 && ((act >> sub) & 1u)) ? (slot_edge_mask(sc, u) & (sc.D3 != 0 ? 0x3F : 0xF)) << 2 : 0);
        const bool same = cache_ok && gor<G>((int)(sig != sig0)) == 0;
        partial = __all(same);                      // (the groups of the wave that are in this branch)
      }
      if constexpr (S == KM_SUB_ALL && G == 16) { if (prob == KM_SUB_ALL) pf.cnt(partial ? 41 : 42, 1); else pf.cnt(43, partial ? 1 : 0x10000); }
      if (partial) newton_hessian_sl<NL, G, S, true>(w, sub, cr, mdiag, qf, ql, W, h, in, act, JOINT);
      else newton_hessian_sl<NL, G, S>(w, sub, cr, mdiag, qf, ql, W, h, in, act, JOINT);
      pf.ph(9 + 6 * S);
      // ---- p = -H^-1 grad
      int hbad = 0;
      bool blocks = false;
      if constexpr (S == KM_SUB_ARM && G == 32) blocks = lm.split != 0;
      if (blocks) {
        // Two-arm models: the arm problem's Hessian has the inertia's two diagonal blocks (a finger / link sphere on the table
        // touches one arm only).  Each DPP row factorises and solves ONE block with the one-row code: lane c of row r takes over
        // row base_r + c of H (block-local columns) and that dof's gradient from the lane that built them, and hands the
        // direction back -- 13 wave shuffles around two 11-pivot solves side by side instead of one 20-pivot solve across rows.
        if constexpr (S == KM_SUB_ARM && G == 32) {
          constexpr int NB = KM_BLOCK_MAX;
          const int split = lm.split, lane0 = (threadIdx.x & 63) & ~31;
          const int row = (threadIdx.x >> 4) & 1, c = threadIdx.x & 15;
          const int base = row ? split : 0, nb = row ? NL - split : split;
          const bool on = c < nb;
          const int src = lane0 + (on ? base + c : 0);
          real mine[NB], loc[NB];                                  // my dof's row of H in ITS block's column order
#pragma unroll
          for (int k = 0; k < NB; k++) {
            const real lo = h[k], hi = split == 10 ? h[(10 + k) < NL ? 10 + k : NL - 1] : h[(11 + k) < NL ? 11 + k : NL - 1];
            mine[k] = sub < split ? lo : hi;
          }
          // (round 6) all twelve shuffles in flight together: taken one at a time, each pair of ds_bpermute was waited for before the
          // next was issued (twelve round trips per iteration of a two-arm env's arm problem)
          real sv[NB];
#pragma unroll
          for (int k = 0; k < NB; k++) sv[k] = __shfl(mine[k], src, 64);
          real gsrc = __shfl(in ? -grad : 0.0, src, 64);
          static_assert(NB == 11, "the pins below name eleven block columns");
          km_pin(sv[0], sv[1], sv[2], sv[3], sv[4], sv[5]); km_pin(sv[6], sv[7], sv[8], sv[9], sv[10], gsrc);
#pragma unroll
          for (int k = 0; k < NB; k++) loc[k] = (on && k < nb) ? sv[k] : ((!on && k == c) ? 1.0 : 0.0);
          real invl = 0, utl[NB];
          chol_rows1<NB, 0, NB, 0, true>(loc, utl, invl, c, true, hbad);
          if (__any(hbad)) { const int gb = gor<G>(hbad); if (gb && sub == 0) w.bad = 1; }
          pf.ph(10 + 6 * S);
          const real pl = chol_solve_rows1<NB, 0, NB, 0>(loc, utl, invl, on ? gsrc : 0.0);
          const int back = lane0 + (sub < split ? sub : 16 + (sub < NL ? sub - split : 0));
          const real pb = __shfl(pl, back, 64);
          p = in ? pb : 0.0;
        }
      } else {
        // blocks that sit inside one DPP row use the one-row code (single-arm models: every subset; two-arm models: the cube
        // block, dofs NL..NL+5 of the group's second row)
        constexpr bool onerow = G == 16 || (S == KM_SUB_CUBE && NL >= 16);
        real invd = 0;
        if constexpr (onerow) {
          constexpr int BASE = G == 16 ? 0 : 16, ND = SS::D1 - SS::D0;
          const int sl = sub - BASE;
          const bool live = G == 16 || sub >= 16;
          real ut[NV];
          if constexpr (ND <= 6) {
            chol_rows1<NV, SS::D0, SS::D1, BASE, true>(h, ut, invd, sl, live, hbad);
          } else if constexpr (S == KM_SUB_ALL && G == 16) {
            if (partial) {
              // L's arm columns from LDS (row `sub` of the factor: exact zeros on and above the diagonal), their updates replayed on
              // the cube columns, then the cube block's six pivots
#pragma unroll
              for (int k = 0; k < NL; k++) h[k] = w.LT[sub][k];
              static_for<0, NL>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                fnmac_cols<16, NL, NV, NV, true>(h, bsrc<16>(h[k]), h[k]);
              });
              dpp_settle(h[NL]);
              real invc = 0;
              chol_rows1<NV, NL, NV, 0, false>(h, ut, invc, sl, live, hbad);
              invd = sub < NL ? invd_keep : invc;
#pragma unroll
              for (int k = NL; k < NV; k++) w.LT[sub][k] = h[k];
              GSYNC();
#pragma unroll
              for (int k = 0; k < NV; k++) ut[k] = w.LT[k][sub];
              GSYNC();
            } else {
              chol_rows1<NV, SS::D0, SS::D1, BASE, false>(h, ut, invd, sl, live, hbad);
              chol_transpose<NV, SS::D0, SS::D1, BASE>(w.LT, h, ut, sl);
              invd_keep = invd; sig0 = sig; cache_ok = true;
            }
          } else {
            chol_rows1<NV, SS::D0, SS::D1, BASE, false>(h, ut, invd, sl, live, hbad);
            chol_transpose<NV, SS::D0, SS::D1, BASE>(w.LT, h, ut, sl);
          }
          if (hbad && sub == 0) w.bad = 1;
          pf.ph(10 + 6 * S);
          p = chol_solve_rows1<NV, SS::D0, SS::D1, BASE>(h, ut, invd, in ? -grad : 0.0);
        } else {
          chol_rows<G, NV, SS::D0, SS::D1>(h, invd, sub, hbad);
          if (hbad && sub == 0) w.bad = 1;
          pf.ph(10 + 6 * S);
          p = chol_solve_rows<G, NV, SS::D0, SS::D1>(h, invd, sub, in ? -grad : 0.0);
        }
      }
      pf.ph(11 + 6 * S);
    }
    // ---- exact line search on phi(alpha) = cost(a + alpha p)
    real Mp;
    if constexpr (S == KM_SUB_CUBE) Mp = mdiag * p; else Mp = mass_mul<NL, G>(cr, sub, mdiag, p);
    real s3[3] = {in ? p * Mr : 0.0, p * Mp, in ? p * grad : 0.0};
    gsum_n<G, 3>(s3);
    const real gp = s3[0], pMp = s3[1], d10 = s3[2];
    real y[4];
    slot_project<NL, G, S>(w, cr, act, sub, p, y);
    if (!slin) { y[0] = 0; y[1] = 0; y[2] = 0; y[3] = 0; }     // slots outside the subset do not move
    const real xf = a - cr.areff, xl = cr.sg * a - cr.arefl, yl = cr.sg * p;
    pf.ph(12 + 6 * S);
    real alpha = 0, lo = 0, hi = INFINITY;
    if (d10 < 0) {
      alpha = 1;
      for (int it = 0; it < 50; it++) {
        real e1 = 0, e2 = 0;
        if (in && cr.fl > 0) row_ls(0, xf + alpha * p, p, cr.Rf, cr.Df, cr.fl, e1, e2);
        if (in && cr.sg != 0) row_ls(1, xl + alpha * yl, yl, cr.Rl, cr.Dl, 0.0, e1, e2);
        const real X[4] = {u[0] + alpha * y[0], u[1] + alpha * y[1], u[2] + alpha * y[2], u[3] + alpha * y[3]};
        slot_ls(sc, X, y, e1, e2);
        real e12[2] = {e1, e2};
        gsum_n<G, 2>(e12);
        const real d1 = gp + alpha * pMp + e12[0];
        const real d2 = pMp + e12[1];
        if (fabs(d1) <= 1e-8 * fabs(d10)) break;        // MuJoCo's ls_tolerance is 1e-2; the outer Newton absorbs the rest
        if (d1 < 0) lo = alpha; else hi = alpha;
        if (hi - lo <= 1e-14 * hi) break;                 // bracket collapsed to roundoff
        if (it == 49) break;
        real an = alpha - d1 * frcp(d2);
        if (!(an > lo && an < hi)) an = isfinite(hi) ? 0.5 * (lo + hi) : 2 * alpha + 1;
        alpha = an;
      }
    }
    pf.ph(13 + 6 * S);
    // ---- advance the point and everything linear in it, evaluate
    a += alpha * p;
    Mr += alpha * Mp;
#pragma unroll
    for (int k = 0; k < 4; k++) u[k] += alpha * y[k];
    real F[4];
    const real cslot = slot_eval<true>(sc, u, F, W);
    real cl = slin ? cslot : 0.0;
    if (in) {
      cl += 0.5 * (a - a_s) * Mr;
      grad = Mr;
      qf = 0; ql = 0;
      if (cr.fl > 0) { real f; cl += row_eval(0, a - cr.areff, cr.Rf, cr.Df, cr.fl, f, qf); grad -= f; }
      if (cr.sg != 0) { real f; cl += row_eval(1, cr.sg * a - cr.arefl, cr.Rl, cr.Dl, 0.0, f, ql); grad -= cr.sg * f; }
    }
    real gsl = grad;
    slot_grad<NL, G, S>(cr, act, F, gsl);
    if (in) grad = gsl;
    const real g1 = in ? grad : 0.0;
    real cg[2] = {cl, g1 * g1};
    gsum_n<G, 2>(cg);
    const real cost_new = cg[0];
    const real improvement = scale * (cost - cost_new), gradient = scale * km_sqrt(cg[1]);
    cost = cost_new;
    pf.ph(14 + 6 * S);
#ifdef KM_PROFILE
    if constexpr (JOINT) { if (prob == KM_SUB_ALL) { pf.cnt(lone_it ? 44 : 45, 1); pf.it_end(lone_it ? 46 : 47); } }
#endif
    if (improvement < tol || gradient < tol || w.bad || iter + 1 >= maxit) {
      if (!(JOINT && prob == KM_SUB_ARM)) break;
      enter(KM_SUB_CUBE); cost = cost_b;        // an uncoupled env of the joint loop: on to its cube problem
      if (small()) break;
      iter = -1;
    }
  }
}

template <int NL, int G>
__device__ __forceinline__ real solve_newton_sl(Ws<NL>& w, const LModel<NL>& lm, const KModelDesc* m, int sub, CReg<NL>& cr, real a_s,
                                                real invm, Prof& pf) {
  constexpr int NV = Dim<NL>::NV;
  const uint32_t act = w.cact;
  const real warm = sub < NV ? w.warm[sub] : 0.0;
  const real mdiag = (sub >= NL && sub < NV) ? 1.0 / invm : 0.0;
  real grad; int qf, ql;
  real u[4], W[7];
  real c0, c1, csl;
  real a = warm;
  real Mr = mass_mul<NL, G>(cr, sub, mdiag, warm - a_s);
  newton_eval_sl<NL, G, true>(w, sub, cr, a, a_s, Mr, grad, qf, ql, u, W, c0, c1, csl);
  real c3[3] = {csl, c0, c1};
  gsum_n<G, 3>(c3);
  const real cs = c3[0];
  real cost0 = c3[1], cost1 = c3[2];
  pf.ph(8);
  if (!(cost0 + cost1 < cs)) {
    a = a_s; Mr = 0;
    real dummy;
    newton_eval_sl<NL, G, false>(w, sub, cr, a, a_s, Mr, grad, qf, ql, u, W, c0, c1, dummy);
    real c2[2] = {c0, c1};
    gsum_n<G, 2>(c2);
    cost0 = c2[0]; cost1 = c2[1];
    pf.ph(38);
  }
  constexpr uint32_t FC_MASK = ((1u << Dim<NL>::NSS) - 1u) << 4;           // sphere-cube slots couple arm and cube
  const bool coupled = (act & FC_MASK) != 0;                               // (group-uniform)
  if constexpr (G == 16) {
    // what this group still has to run in its own loops: its arm problem and then its cube problem (an uncoupled env in a wave
    // without a coupled one), or -- after a joint loop -- whatever the joint loop handed back (nothing for the coupled env itself)
    int resume = KM_SUB_ARM, riter = 0;
    real rcost = cost0;
    if (__any(coupled)) {
      // a coupled env in the wave: its whole-problem loop and the wave-mates' arm and cube loops share one instruction stream
      // for as long as the coupled env iterates
      resume = 0;
      newton_loop_sl<NL, G, KM_SUB_ALL, true>(w, lm, m, sub, cr, mdiag, a_s, a, Mr, coupled ? cost0 + cost1 : cost0, grad, qf, ql, u, W, pf, !coupled, cost1,
                                              0, &resume, &rcost, &riter);
    }
    if (resume == KM_SUB_ARM) newton_loop_sl<NL, G, KM_SUB_ARM>(w, lm, m, sub, cr, mdiag, a_s, a, Mr, rcost, grad, qf, ql, u, W, pf, false, 0, riter);
    if (resume != 0) newton_loop_sl<NL, G, KM_SUB_CUBE>(w, lm, m, sub, cr, mdiag, a_s, a, Mr, resume == KM_SUB_ARM ? cost1 : rcost, grad, qf, ql, u, W, pf,
                                                        false, 0, resume == KM_SUB_ARM ? 0 : riter);
  } else {
    // two-row groups keep the separate loops: their cube block runs the one-row code in the second DPP row while the whole
    // problem runs the two-row code -- different operation order, so a joint loop would make an env's bits depend on its wave-mates
    if (!coupled) newton_loop_sl<NL, G, KM_SUB_ARM>(w, lm, m, sub, cr, mdiag, a_s, a, Mr, cost0, grad, qf, ql, u, W, pf);
    if (coupled) newton_loop_sl<NL, G, KM_SUB_ALL>(w, lm, m, sub, cr, mdiag, a_s, a, Mr, cost0 + cost1, grad, qf, ql, u, W, pf);
    else newton_loop_sl<NL, G, KM_SUB_CUBE>(w, lm, m, sub, cr, mdiag, a_s, a, Mr, cost1, grad, qf, ql, u, W, pf);
  }
  return a;
}

template <int NL, int G>
__device__ __forceinline__ real solve_newton(Ws<NL>& w, const LModel<NL>& lm, const KModelDesc* m, int sub, int actuation,
                                             CReg<NL>& cr, real invm, Prof& pf) {
  constexpr int NV = Dim<NL>::NV;
  // ---- actuation and smooth acceleration (as in the PGS path)
  // (round 6) the lane's inputs and its row of M^-1 in one batch in front of the exchange, the right-hand sides in one behind it
  const int si = sub < NL ? sub : NL - 1, sv = sub < NV ? sub : NV - 1;
  real bia = w.bias[sv], ctl = w.ctrl[si], cr0 = lm.ctrlrange[si][0], cr1 = lm.ctrlrange[si][1], kpv = lm.kp[si], qps = w.qpos[si];
  real fr0 = lm.forcerange[si][0], fr1 = lm.forcerange[si][1];
  int flim = lm.forcelimited[si];
  real mrow[NL];
#pragma unroll
  for (int j = 0; j < NL; j++) mrow[j] = w.Minv[si][j];
  km_pin(bia, ctl, cr0, cr1, kpv, qps); km_pin(fr0, fr1); km_pin_i(flim);
  real rhs = -bia;
  if (sub < NV) {
    if (actuation && sub < NL) {
      real c = fmin(fmax(ctl, cr0), cr1);
      real force = kpv * c - kpv * qps;
      if (flim) force = fmin(fmax(force, fr0), fr1);
      rhs += force;
    }
    w.tmp[sub] = rhs;
  }
  GSYNC();
  real tv[NL];
#pragma unroll
  for (int j = 0; j < NL; j++) tv[j] = w.tmp[j];
  real a_s = 0;
  if (sub < NL) {
#pragma unroll
    for (int j = 0; j < NL; j++) a_s += mrow[j] * tv[j];
  } else if (sub < NV) a_s = rhs * invm;
  pf.ph(7);
  return solve_newton_sl<NL, G>(w, lm, m, sub, cr, a_s, invm, pf);
}

// everything mj_step1 computes that mj_step2 needs, at the state held in w.qpos / w.qvel
template <int NL, int G, int SOLVER>
__device__ __forceinline__ void step1_products(Ws<NL>& w, const LModel<NL>& lm, const KModelDesc* m, int sub,
                                               CReg<NL>& cr, real invm, Prof& pf) {
  real kin[15];
  fk_parallel<NL, G>(w, lm, sub, G == 16 ? kin : nullptr);
  pf.ph(0);
  real FN[6];
  // two-row groups with a block split: row r of the group = block r of the robot (lane c <-> link base + c) for the two tree
  // passes; everything else keeps lane = dof
  const int split = G == 32 ? lm.split : 0;
  int bli = -1, bbase = 0;
  if constexpr (G == 32) {
    const int row = (threadIdx.x >> 4) & 1, c = threadIdx.x & 15;
    bbase = row ? split : 0;
    bli = (split && c < (row ? NL - split : split)) ? bbase + c : -1;
    if (split) {                                 // entries between the blocks: never written below, read as part of the rows
      for (int e = sub; e < (int)(sizeof(w.Minv) / sizeof(real)); e += G) (&w.Minv[0][0])[e] = 0.0;      // (the padded rows whole)
    }
  }
  if constexpr (G == 16) bias_bodies_rows<NL, NL>(w, lm, m, sub < NL ? sub : -1, 0, sub == NL, FN, kin);
  else if (split) {
    bias_bodies_rows<NL, KM_BLOCK_MAX>(w, lm, m, bli, bbase, false, FN);
    if (sub == NL) cube_bias<NL>(w, m);          // (lane NL also works on a link of the second block above)
  }
  else bias_bodies_parallel<NL, G>(w, lm, m, sub);
  pf.ph(1);
  collide_parallel<NL, G>(w, lm, m, sub);
  if constexpr (SOLVER != KM_SOLVER_NEWTON) { if (sub == 0) scalar_rows_serial<NL>(w, lm); }
  GSYNC();
  pf.ph(2);
  if constexpr (G == 16) {
    composite_mass_bias_rows<NL, NL>(w, lm, sub < NL ? sub : -1, 0, FN, kin);
  } else if (split) {
    composite_mass_bias_rows<NL, KM_BLOCK_MAX>(w, lm, bli, bbase, FN);
  } else {
    composite_own<NL, G>(w, lm, sub);      // (comp aliases the bias scratch: its last reader is before the barrier above)
    GSYNC();
    composite_accumulate<NL, G>(w, lm, sub);
    GSYNC();
    mass_matrix<NL, G>(w, lm, sub);
    bias_project<NL, G>(w, lm, sub);
  }
  GSYNC();
  pf.ph(3);
  invert_mass<NL, G>(w, sub, cr, lm.split, pf);
  pf.ph(4);
  if constexpr (SOLVER == KM_SOLVER_NEWTON) build_constraints_newton<NL, G>(w, lm, m, sub, cr, invm);
  else build_constraints<NL, G>(w, lm, m, sub, cr, invm);
  pf.ph(5);
}
template <int NL, int G, int SOLVER>
__device__ __forceinline__ real solve(Ws<NL>& w, const LModel<NL>& lm, const KModelDesc* m, int sub, int actuation,
                                      CReg<NL>& cr, real invm, Prof& pf) {
  if constexpr (SOLVER == KM_SOLVER_NEWTON) return solve_newton<NL, G>(w, lm, m, sub, actuation, cr, invm, pf);
  else {
    real a = solve_accel<NL, G>(w, lm, m, sub, actuation, cr, invm);
    pf.ph(6);
    return a;
  }
}

// mj_Euler: qvel += dt*qacc, then positions with the NEW velocity (semi-implicit); free-joint quaternion
// integrated on the group's lane 0
template <int NL, int G>
__device__ __forceinline__ void integrate(Ws<NL>& w, const KModelDesc* m, int sub, real a) {
  constexpr int NV = Dim<NL>::NV;
  const real dt = m->timestep;
  // (round 6) the lane's velocity / position and the cube's quaternion in one batch; the new angular velocity reaches lane 0 by row
  // broadcast instead of through LDS (one synchronisation and one round trip less in front of the quaternion's serial chain)
  const int sv = sub < NV ? sub : NV - 1, sq = sub < NL + 3 ? sub : NL + 2;
  real v0 = w.qvel[sv], qp = w.qpos[sq], q[4] = {w.qpos[NL + 3], w.qpos[NL + 4], w.qpos[NL + 5], w.qpos[NL + 6]};
  km_pin(v0, qp, q[0], q[1], q[2], q[3]);
  real v = 0;
  if (sub < NV) {
    v = v0 + dt * a;
    w.qvel[sub] = v;
    w.warm[sub] = a;
    if (sub < NL + 3) w.qpos[sub] = qp + dt * v;
  }
  real ax[3] = {gbcast<G, NL + 3>(v), gbcast<G, NL + 4>(v), gbcast<G, NL + 5>(v)};
  if (sub == 0) {
    real ang = dt * normalize3_fast(ax), qr[4], qn[4];
    axis_angle2quat(qr, ax, ang);
    normalize4_fast(q);
    qmul(qn, q, qr);
    normalize4_fast(qn);
    w.qpos[NL + 3] = qn[0]; w.qpos[NL + 4] = qn[1]; w.qpos[NL + 5] = qn[2]; w.qpos[NL + 6] = qn[3];
  }
  GSYNC();
}

// [-1, 1] clip of an observation component
__device__ __forceinline__ real clip1(real x) { return fmin(fmax(x, -1.0), 1.0); }

// get_observation, env_sim.py:110-146 (state keys; cameras are out of this kernel)
template <int NL, int G>
__device__ __forceinline__ void write_obs(const Ws<NL>& w, const LModel<NL>& lm, const KModelDesc* m, int sub, double* obs_row) {
  for (int i = sub; i < NL; i += G) {
    obs_row[i] = clip1((w.qpos[i] - lm.range[i][0]) / (lm.range[i][1] - lm.range[i][0]));
    obs_row[NL + i] = clip1(w.qvel[i] / m->max_q_vel);
  }
  for (int c = sub; c < 7; c += G) {
    if (c < 3) obs_row[2 * NL + c] = clip1((w.qpos[NL + c] - m->cube_spawn_lo[c]) / (m->cube_spawn_hi[c] - m->cube_spawn_lo[c]));
    else obs_row[2 * NL + c] = w.qpos[NL + c];
  }
}

// lo + (hi - lo) * u with the product rounded before the sum: the cube spawn is compared bit-for-bit with the oracle
__device__ __forceinline__ real lerp_unfused(real lo, real hi, real u) {
#pragma clang fp contract(off)
  const real d = hi - lo;
  const real p = d * u;
  return lo + p;
}

// initialize_episode (env_sim.py:23-36) + mj_forward without actuation (dm_control after_reset)
template <int NL, int G, int SOLVER>
__device__ __forceinline__ void reset_env(Ws<NL>& w, const LModel<NL>& lm, const KModelDesc* m, int sub, uint64_t seed,
                                          int64_t genv, int episode, CReg<NL>& cr, real invm, Prof& pf) {
  constexpr int NV = Dim<NL>::NV;
  if (sub < NV) { w.qvel[sub] = 0; w.warm[sub] = 0; }
  if (sub < NL) { w.qpos[sub] = lm.q_home[sub]; w.ctrl[sub] = lm.q_home[sub]; }
  if (sub == 0) {
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t ctr[4] = {(uint32_t)genv, (uint32_t)((uint64_t)genv >> 32), (uint32_t)episode, 0}, o[4];
    philox4x32_10(ctr, key, o);
    real u0 = u53(o[0], o[1]), u1 = u53(o[2], o[3]);
    ctr[3] = 1;
    philox4x32_10(ctr, key, o);
    real u2 = u53(o[0], o[1]);
    w.qpos[NL] = lerp_unfused(m->cube_spawn_lo[0], m->cube_spawn_hi[0], u0);
    w.qpos[NL + 1] = lerp_unfused(m->cube_spawn_lo[1], m->cube_spawn_hi[1], u1);
    w.qpos[NL + 2] = lerp_unfused(m->cube_spawn_lo[2], m->cube_spawn_hi[2], u2);
    for (int c = 0; c < 4; c++) w.qpos[NL + 3 + c] = m->cube_quat0[c];
    w.bad = 0;
  }
  GSYNC();
  step1_products<NL, G, SOLVER>(w, lm, m, sub, cr, invm, pf);
  real a = solve<NL, G, SOLVER>(w, lm, m, sub, 0, cr, invm, pf);
  if (sub < NV) w.warm[sub] = a;
  GSYNC();
}

// contact points of slots that never became active are read (and multiplied by zero weights) by the slot-lane solver: give
// them finite values once per launch
template <int NL>
__device__ __forceinline__ void init_ws(Ws<NL>& w, int sub) {
  if (sub < Dim<NL>::NC) { w.c_pos[sub][0] = 0; w.c_pos[sub][1] = 0; w.c_pos[sub][2] = 0; w.c_dist[sub] = 0; }
}
// fused = before_step runs in this kernel: ctrl <- float32(ctrl) (env_sim.py:40) and qpos_ik <- qpos here
template <int NL, int G>
__device__ __forceinline__ void load_state(Ws<NL>& w, const KDeviceState& st, int env, int sub, bool fused) {
  constexpr int NV = Dim<NL>::NV, NQ = Dim<NL>::NQ;
  const int NE = st.num_envs;
  // (round 6) every column read of the env's state issued before the first is waited for: as loops, each HBM read was waited for
  // on its own (five to six round trips at the start of every wave)
  constexpr int KQ = (NQ + G - 1) / G, KV = (NV + G - 1) / G, KL = (NL + G - 1) / G;
  real q[KQ], v[KV], wm[KV], c[KL], qi[KL];
#pragma unroll
  for (int k = 0; k < KQ; k++) { const int i = sub + G * k; q[k] = st.qpos[(size_t)(i < NQ ? i : NQ - 1) * NE + env]; }
#pragma unroll
  for (int k = 0; k < KV; k++) { const int i = sub + G * k, ic = i < NV ? i : NV - 1; v[k] = st.qvel[(size_t)ic * NE + env]; wm[k] = st.warm[(size_t)ic * NE + env]; }
#pragma unroll
  for (int k = 0; k < KL; k++) {
    const int i = sub + G * k, ic = i < NL ? i : NL - 1;
    c[k] = st.ctrl[(size_t)ic * NE + env];
    qi[k] = 0.0;
    if (!fused) qi[k] = st.qpos_ik[(size_t)ic * NE + env];      // (wave-uniform: the split-launch path only)
  }
#pragma unroll
  for (int k = 0; k < KQ; k++) { const int i = sub + G * k; if (i < NQ) { w.qpos[i] = q[k]; if (fused && i < NL) w.qpos_ik[i] = q[k]; } }
#pragma unroll
  for (int k = 0; k < KV; k++) { const int i = sub + G * k; if (i < NV) { w.qvel[i] = v[k]; w.warm[i] = wm[k]; } }
#pragma unroll
  for (int k = 0; k < KL; k++) {
    const int i = sub + G * k;
    if (i < NL) {
      w.ctrl[i] = fused ? (real)(float)c[k] : c[k];
      if (!fused) w.qpos_ik[i] = qi[k];
    }
  }
  if (sub == 0) { w.bad = 0; w.work = 0; }
}
template <int NL> struct LdsIO {
  Ws<NL>& w; const KDeviceState& st; int env;
  __device__ __forceinline__ real qpos(int i) const { return w.qpos[i]; }
  __device__ __forceinline__ void set_ctrl(int i, real v) { w.ctrl[i] = v; }
  __device__ __forceinline__ void set_qpos_ik(int i, real v) { w.qpos_ik[i] = v; }
  __device__ __forceinline__ void set_diag(int arm, int nfev, int status) {
    st.ik_nfev[(size_t)arm * st.num_envs + env] = nfev; st.ik_status[(size_t)arm * st.num_envs + env] = status;
  }
};
template <int NL, int G>
__device__ __forceinline__ void store_state(const Ws<NL>& w, const KDeviceState& st, int env, int sub) {
  constexpr int NV = Dim<NL>::NV, NQ = Dim<NL>::NQ;
  const int NE = st.num_envs;
  for (int i = sub; i < NQ; i += G) st.qpos[(size_t)i * NE + env] = w.qpos[i];
  for (int i = sub; i < NV; i += G) { st.qvel[(size_t)i * NE + env] = w.qvel[i]; st.warm[(size_t)i * NE + env] = w.warm[i]; }
  for (int i = sub; i < NL; i += G) st.ctrl[(size_t)i * NE + env] = w.ctrl[i];
}

// per-link constants -> LDS: a flat copy of the image k_prepare_model built at kmanip_create (KDeviceModel::staged), one batch of
// 16-byte loads per lane, then a workgroup barrier (one wave: cheap)
template <int NL>
__device__ __forceinline__ void stage_model(LModel<NL>& lm, const KDeviceModel* dm) {
  static_assert(sizeof(LModel<NL>) % 16 == 0 && sizeof(LModel<NL>) <= KM_LMODEL_MAX, "the staged image is copied in 16-byte pieces");
  constexpr int N16 = (int)(sizeof(LModel<NL>) / 16);
  const uint4* src = reinterpret_cast<const uint4*>(dm->staged);
  uint4* dst = reinterpret_cast<uint4*>(&lm);
  uint4 v[(N16 + 63) / 64];
#pragma unroll
  for (int k = 0; k < (N16 + 63) / 64; k++) { const int i = threadIdx.x + 64 * k; v[k] = src[i < N16 ? i : N16 - 1]; }
#pragma unroll
  for (int k = 0; k < (N16 + 63) / 64; k++) { const int i = threadIdx.x + 64 * k; if (i < N16) dst[i] = v[k]; }
  __syncthreads();
}
// the image itself: per-link constants with all 64 lanes, the derived scalars on lane 0 (k_prepare_model only)
template <int NL>
__device__ __forceinline__ void build_lmodel(LModel<NL>& lm, const KDeviceModel* dm) {
  const KModelDesc* m = &dm->d;
  for (int i = threadIdx.x; i < NL; i += 64) {
    lm.parent[i] = m->link_parent[i]; lm.jtype[i] = m->jnt_type[i]; lm.forcelimited[i] = m->forcelimited[i];
    lm.anc[i] = dm->x.anc_mask[i]; lm.desc[i] = dm->x.desc_mask[i];
    for (int k = 0; k < 4; k++) lm.jump[k][i] = dm->x.jump[k][i];
    if (i == 0) {
      lm.fk_rounds = dm->x.fk_rounds; lm.split = dm->x.split;
      get_kb(m, m->con_def_solref, m->con_def_solimp, lm.kb[0][0], lm.kb[0][1]);
      get_kb(m, m->con_cube_solref, m->con_cube_solimp, lm.kb[1][0], lm.kb[1][1]);
      lm.imp0[0] = impedance(m->con_def_solimp, 0.0);
      lm.imp0[1] = impedance(m->con_cube_solimp, 0.0);
      stage_imp(lm.imp[0], m->con_def_solimp); stage_imp(lm.imp[1], m->con_cube_solimp);
      lm.cubew[0] = m->cube_invweight0[0]; lm.cubew[1] = m->cube_invweight0[1];
      lm.fric[0][0] = m->con_def_friction[0]; lm.fric[0][1] = m->con_def_friction[1];
      lm.fric[1][0] = m->con_cube_friction[0]; lm.fric[1][1] = m->con_cube_friction[1];
      lm.scale = 1.0 / (m->meaninertia * (NL + 6));
      const real muc = m->con_cube_friction[0], mud = m->con_def_friction[0], cw = m->cube_invweight0[0];
      lm.cornerA = cw + muc * muc * cw;
      for (int sp = 0; sp < Dim<NL>::NSPH; sp++) {
        const real lw = sp < m->nsphere ? m->body_invweight0[m->sphere_link[sp]][0] : 0.0;
        lm.sphA[0][sp] = (cw + lw) + muc * muc * (cw + lw);
        lm.sphA[1][sp] = lw + mud * mud * lw;
      }
    }
    if (i < Dim<NL>::NSPH) {           // (NSPH <= NL)
      const int sp = i < m->nsphere ? i : 0;          // (unused candidates: finite copies, never tested)
      lm.sph_link[i] = m->sphere_link[sp]; lm.sph_rad[i] = m->sphere_radius[sp];
      for (int c = 0; c < 3; c++) { lm.sph_pos[i][c] = m->sphere_pos[sp][c]; lm.sph_seg[i][c] = m->sphere_seg[sp][c]; }
      if (i == 0) lm.nsph = m->nsphere < Dim<NL>::NSPH ? m->nsphere : Dim<NL>::NSPH;
    }
    lm.dofw[i] = m->dof_invweight0[i];
    lm.floss[i] = m->frictionloss[i]; lm.kp[i] = m->kp[i]; lm.mass[i] = m->mass[i]; lm.q_home[i] = m->q_home[i];
    for (int c = 0; c < 3; c++) { lm.pos[i][c] = m->link_pos[i][c]; lm.jaxis[i][c] = m->jnt_axis[i][c]; lm.com[i][c] = m->com[i][c]; lm.inertia[i][c] = m->inertia[i][c]; }
    for (int c = 0; c < 4; c++) lm.quat[i][c] = m->link_quat[i][c];
    { real qn[4] = {m->link_quat[i][0], m->link_quat[i][1], m->link_quat[i][2], m->link_quat[i][3]}, Rm[9]; normalize4(qn); quat2mat(Rm, qn); for (int c = 0; c < 9; c++) lm.R[i][c] = Rm[c]; }
    for (int c = 0; c < 2; c++) { lm.range[i][c] = m->jnt_range[i][c]; lm.ctrlrange[i][c] = m->ctrlrange[i][c]; lm.forcerange[i][c] = m->forcerange[i][c]; }
  }
  __syncthreads();
}

// get_reward, env_sim.py:148-179, from the kinematics and contacts of a trailing mj_step1 (fk_parallel + collide_parallel)
template <int NL, int G>
__device__ __forceinline__ real env_reward(Ws<NL>& w, const KModelDesc* m, int sub) {
  constexpr int NV = Dim<NL>::NV;
  real v2 = gsum<G>(sub < NV ? w.qvel[sub] * w.qvel[sub] : 0.0);
  GSYNC();
  real rew = -m->reward_vel_penalty * km_sqrt(v2);
  for (int arm = 1; arm >= 0; arm--) {
    if (!m->arm_present[arm] || !m->arm_has_grip[arm]) continue;
    const int l = m->arm_site_link[arm];
    real so[3] = {m->arm_site_pos[arm][0], m->arm_site_pos[arm][1], m->arm_site_pos[arm][2]}, sp[3];
    mat_vec3(sp, w.k.xmat[l], so);
    real df[3] = {w.qpos[NL] - (sp[0] + w.k.xpos[l][0]), w.qpos[NL + 1] - (sp[1] + w.k.xpos[l][1]), w.qpos[NL + 2] - (sp[2] + w.k.xpos[l][2])};
    rew += m->reward_grip_dist * (1.0 / (km_sqrt(dot3(df, df)) + m->epsilon));
  }
  if (m->touch_reward_enabled && (w.contact_mask & KM_CON_FINGERS_CUBE(NL))) {      // a FINGER on the cube (palm / link spheres do not count)
    rew += m->reward_touch_cube;
    if (!w.touch_ct) rew += m->reward_lift_cube;
  }
  return rew;
}

// ---------------------------------------------------------------------------------------------
// EPB = envs per single-wave workgroup (<= 64 / G).  Fewer envs per wave = more waves per SIMD: the kernel is
// bound by LDS/dependent-issue latency, so waves of different envs hide each other's waits.
template <int NL, int G, int SOLVER, int EPB, bool CHUNK>
__global__ __launch_bounds__(64) void k_step(const KDeviceModel* __restrict__ dm, KDeviceState st, const float* __restrict__ act,
                                             double* __restrict__ obs, double* __restrict__ reward, uint8_t* __restrict__ done,
                                             int nchunk) {
  constexpr int NV = Dim<NL>::NV, NQ = Dim<NL>::NQ;
  __shared__ Ws<NL> ws[EPB];
  __shared__ LModel<NL> lm;
  const KModelDesc* m = &dm->d;
  const int lane = threadIdx.x, grp = lane / G, sub = lane % G;
  // wave slot -> env: the identity behind the XCD-aware block mapping, or -- launches of several residency rounds -- the
  // predicted-cost order of k_sort_envs with workgroup 0 first (longest-processing-time-first dispatch); or the previous
  // launch's dispatch list.  Looked up BEFORE the model is staged: the list's two dependent loads wait beside the staging's own.
  int slot = (st.slot_env ? (int)blockIdx.x : xcd_block(blockIdx.x, gridDim.x)) * EPB + grp;
  int env = -1;
  unsigned long long heavy_mask = 0, s1_mask = 0, s2_mask = 0;
  if (st.spread_in) {
    // SPREAD (KDeviceState): the 64 flags of this wave's block of 64 consecutive envs, one per lane (every lane votes: before any exit)
    const int blk = (slot - grp) / 64;
    const int fl = st.spread_in[blk * 64 + lane];
    heavy_mask = __ballot((fl & 1) != 0);
    s1_mask = __ballot((fl & 2) != 0);
    s2_mask = __ballot((fl & 4) != 0);
  }
  if (grp >= EPB) {
  } else if (st.spread_in) {
    const int wv = (slot - grp) / EPB;                        // this wave's index in slot space (xcd_block keeps an XCD's waves together)
    env = (wv / (64 / EPB)) * 64 + spread_pick(heavy_mask, s1_mask, s2_mask, wv % (64 / EPB), grp, EPB);
  } else if (st.disp_in) {
    // dispatch list (KDeviceState; kmanip_api.hip): SPREAD, or the heavy-first experiment -- the first workgroups hold the envs
    // predicted heavy, disp_heavy_epb of them per wave (1: no wave-mates to wait for at the IK, at the solves, in the joint loop),
    // light envs fill the following waves EPB at a time
    const int N = st.num_envs, nh = min(st.disp_in[0], st.disp_cap), hepb = st.disp_heavy_epb;
    const int b = blockIdx.x;
    int idx = -1;
    // (never seen: the list is filled by the previous launch's waves, one entry each.  A list that does not add up -- a launch
    // that was aborted half way -- must not lose or duplicate an env: the identity map instead)
    const bool ident = nh + st.disp_in[1] != N;
    if (ident) {
      idx = EPB * b + grp;
    } else if (hepb == 0) {
      // SPREAD: the grid stays one wave per EPB envs; wave b < nh takes heavy env b into its lane group 0 and light envs into the
      // others, so that no wave holds two envs predicted heavy (two coupled envs in one wave run the joint loop for the longer of
      // their iteration counts, with twice the chance of a straggler: those waves end the launch)
      idx = b < nh ? (grp == 0 ? b : nh + (EPB - 1) * b + grp - 1) : EPB * b + grp;
    } else {
      const int nhw = (nh + hepb - 1) / hepb;
      if (b < nhw) { if (grp < hepb && b * hepb + grp < nh) idx = b * hepb + grp; }
      else idx = nh + (b - nhw) * EPB + grp;
    }
    if (idx >= 0 && idx < N) env = ident ? idx : st.disp_in[KM_DISP_HDR + idx];
    slot = b * EPB + grp;                               // (diagnostics: wave_clk is sized for the grid)
  } else if (slot < st.num_envs) {
    env = st.slot_env ? st.slot_env[slot] : slot;
  }
  stage_model<NL>(lm, dm);
  if (env < 0) return;                                 // whole group exits together
  Ws<NL>& w = ws[grp];
  real invm = 0;                       // diagonal of M^-1 for the cube dof owned by this lane
  if (sub >= NL && sub < NV) invm = sub < NL + 3 ? 1.0 / m->cube_mass : 1.0 / m->cube_inertia[sub - NL - 3];
  Prof pf;
  pf.start();
  const unsigned long long t_wave0 = st.wave_clk ? __builtin_amdgcn_s_memtime() : 0ull;
  const unsigned long long r_wave0 = st.wave_clk ? __builtin_amdgcn_s_memrealtime() : 0ull;
  const bool fused = act != nullptr;
#ifdef KM_DEBUG_NANFILL   // diagnostic build (-DKM_DEBUG_NANFILL=<value>): poison the workspace, so that a read of LDS this launch did not write shows
  { double* wp = reinterpret_cast<double*>(&w); for (int i = sub; i < (int)(sizeof(Ws<NL>) / 8); i += G) wp[i] = KM_DEBUG_NANFILL; GSYNC(); }
#endif
  init_ws<NL>(w, sub);
  load_state<NL, G>(w, st, env, sub, fused);
  int step_idx = st.step_idx[env], episode = st.episode[env];
  const size_t NE = (size_t)st.num_envs;
  GSYNC();
  pf.ph(29);
  // nchunk control steps per launch (kmanip_step: 1).  With a chunk of pre-supplied actions every wave runs its envs
  // through all of them without meeting the other waves at a launch boundary, so the batch advances at the MEAN wave
  // speed instead of the slowest wave's; the state stays in LDS between the steps of a chunk.
  const int nsteps = CHUNK ? nchunk : 1;      // (the single-step kernel keeps its register allocation: no outer loop)
  int heavy_next = 0;                         // this env's class for the next launch's dispatch table (the state it ENDS the step in)
  int table_next = 0;                         // ... and: a sphere on the table (SPREAD's second class)
  for (int kc = 0; kc < nsteps; kc++) {
  if (fused) {
    if (kc > 0) {
      // what load_state does for the first step: ctrl <- float32(ctrl) (env_sim.py:40), qpos_ik <- qpos
      if (sub < NL) { w.ctrl[sub] = (real)(float)w.ctrl[sub]; w.qpos_ik[sub] = w.qpos[sub]; }
      if (sub == 0) { w.bad = 0; w.work = 0; }
      GSYNC();
    }
    // ---- KManipTask.before_step: 8 lanes per arm, one arm per 16-lane DPP row of the group (lanes 0-7 of row 0: right arm;
    // of row 1, in the two-row groups: left arm), the rest idle.
    // Fused here so that an env whose IK needs many evaluations delays only its own wave, not the whole batch.
    // (chunk kernels: the lane index is made opaque once per control step, like the dof index per sub-step below -- otherwise the
    // IK's per-lane chain constants, invariant across the steps of a chunk, are hoisted out of the chunk loop and kept alive
    // through the physics: the two-arm chunk kernels sat at the 512-register cap with 268 / 116 bytes of scratch)
    int subk = sub;
    if constexpr (CHUNK) asm volatile("" : "+v"(subk));
    const int arm = subk / GS;
    if (subk % GS < GI && arm < KM_MAX_ARMS && (NL > 10 || arm == 0) && m->arm_present[arm]) {
      LdsIO<NL> io{w, st, env};
      const float* arow = act + ((size_t)kc * NE + env) * m->act_dim;
      if (m->arm_nq[arm] == 7) coop_before_step<7>(dm, arm, subk % GS, arow, io, &pf);
      else coop_before_step<6>(dm, arm, subk % GS, arow, io, &pf);
    }
    GSYNC();
    pf.ph(30);
  }
  int bad = 0;
  heavy_next = 0; table_next = 0;
  const int nsub = m->n_sub_steps;
  for (int s = 0; s < nsub; s++) {
    // the lane's dof index, opaque to the optimiser once per sub-step: everything derived from it (LDS addresses, per-link
    // constants, masks) is recomputed inside the sub-step instead of being hoisted out of this loop and kept alive -- or
    // shuttled through AGPRs -- across all ten (380 -> 318 registers, measured)
    int subv = sub; asm volatile("" : "+v"(subv));
    CReg<NL> cr;                       // (one per sub-step: nothing of it can be carried round the loops)
    step1_products<NL, G, SOLVER>(w, lm, m, subv, cr, invm, pf);      // s == 0: products of the pre-IK state (stale mj_step2)
    real a = solve<NL, G, SOLVER>(w, lm, m, subv, 1, cr, invm, pf);
    pf.ph(28);
    int lb = (sub < NV) && (!isfinite(a) || fabs(a) > 1e10);   // mjWARN_BADQACC
    bad = gor<G>(lb) | w.bad;
    if (bad) break;
    if (s == 0) {                                        // the IK teleported the arm (ik_mujoco.py:34,67)
      if (sub < NL) w.qpos[sub] = w.qpos_ik[sub];
      GSYNC();
    }
    integrate<NL, G>(w, m, sub, a);
    pf.ph(27);
  }
  if (!bad) {
    int lb = 0;
    for (int i = sub; i < NQ; i += G) lb |= !isfinite(w.qpos[i]);
    bad = gor<G>(lb);
  }
  uint8_t dn = 0;
  real rew = 0;
  double* obs_row = obs + ((size_t)kc * NE + env) * m->obs_dim;
  if (!bad) {
    // trailing mj_step1: kinematics + collision feed reward and the contact mask
    fk_parallel<NL, G>(w, lm, sub);
    const int near_cube = collide_parallel<NL, G, true>(w, lm, m, sub, st.near_margin);
    heavy_next = near_cube;
    // the cost score of an env that is not heavy (spread_pick): bit 1 a sphere on the table, bit 0 a cube that does not rest on four corners
    // (KMANIP_SPREAD_TABLE, A/B: 0 no score; 1 the table bit only; 2 both as ONE class; 3 = default: both bits, four classes)
    {
      const int tb = (w.contact_mask & KM_CON_ANY_SPHERE_TABLE) != 0, cb = __popc(w.contact_mask & KM_CON_ANY_CUBE_TABLE) != 4;
      table_next = st.spread_table == 0 ? 0 : st.spread_table == 1 ? 2 * tb : st.spread_table == 2 ? 2 * (tb | cb) : 2 * tb + cb;
    }
    if constexpr (KM_WORK_COUNTERS(NL)) { if (sub == 0 && near_cube) w.work |= 1 << 30; }     // (bit 30: a collider on or close to the cube)
    rew = env_reward<NL, G>(w, m, sub);
    write_obs<NL, G>(w, lm, m, sub, obs_row);
    if (sub == 0) st.contact_mask[env] = w.contact_mask;
  } else {
    dn |= KM_DONE_DIVERGED;
    for (int i = sub; i < m->obs_dim; i += G) obs_row[i] = 0;
    if (sub == 0) st.contact_mask[env] = 0;
  }
  step_idx += 1;
  if (step_idx >= m->max_episode_steps) dn |= KM_DONE_TRUNCATED;
  if (dn && (m->auto_reset || bad)) {
    episode += 1; step_idx = 0;
    heavy_next = 0; table_next = 0;               // (the respawned cube is nowhere near the home pose)
    GSYNC();
    pf.ph(31);
    CReg<NL> cr;
    reset_env<NL, G, SOLVER>(w, lm, m, sub, st.seed, st.env_id_offset + env, episode, cr, invm, pf);
    write_obs<NL, G>(w, lm, m, sub, obs_row);
    pf.ph(32);
  }
  if (sub == 0) {
    reward[(size_t)kc * NE + env] = rew; done[(size_t)kc * NE + env] = dn;
    if (!CHUNK && st.rd_rec) { st.rd_rec[2 * (size_t)env] = rew; st.rd_rec[2 * (size_t)env + 1] = (real)dn; }
  }
  GSYNC();
  }   // chunk
  if (sub == 0) {
    st.step_idx[env] = step_idx; st.episode[env] = episode;
    if (st.sim_time) st.sim_time[env] = step_idx * st.control_dt;
    if (st.spread_out) st.spread_out[env] = (uint8_t)((heavy_next != 0) | (table_next << 1));      // (a chunk: the state its LAST step ends in)
    if (!CHUNK && st.disp_out) {
      // register for the next launch: heavy envs from the front of the list (at most disp_cap of them), the others from the back
      int pos = -1;
      if (heavy_next) { const int i = atomicAdd(&st.disp_out[0], 1); if (i < st.disp_cap) pos = i; }
      if (pos < 0) pos = st.num_envs - 1 - atomicAdd(&st.disp_out[1], 1);
      st.disp_out[KM_DISP_HDR + pos] = env;
      if (blockIdx.x == 0 && grp == 0) { st.disp_zero[0] = 0; st.disp_zero[1] = 0; }
    }
    if constexpr (KM_WORK_COUNTERS(NL)) st.work[env] = w.work;      // (the last control step's: what the next launch's slot order is predicted from)
    // (diagnostics: core-clock cycles in the low 40 bits; above them the wave's START on the constant 100 MHz clock, 24 bits)
    if (st.wave_clk) st.wave_clk[slot] = ((__builtin_amdgcn_s_memtime() - t_wave0) & 0xFFFFFFFFFFull) | ((r_wave0 & 0xFFFFFFull) << 40);
  }
  store_state<NL, G>(w, st, env, sub);
  pf.ph(31);
  pf.flush();
}

// KManipEnvSim.k_reset for the envs selected by mask (NULL = all)
template <int NL, int G, int SOLVER, int EPB>
__global__ __launch_bounds__(64) void k_reset(const KDeviceModel* __restrict__ dm, KDeviceState st,
                                              const uint8_t* __restrict__ mask, double* __restrict__ obs) {
  constexpr int NV = Dim<NL>::NV;
  __shared__ Ws<NL> ws[EPB];
  __shared__ LModel<NL> lm;
  stage_model<NL>(lm, dm);
  const KModelDesc* m = &dm->d;
  const int lane = threadIdx.x, grp = lane / G, sub = lane % G;
  const int env = xcd_block(blockIdx.x, gridDim.x) * EPB + grp;
  if (grp >= EPB || env >= st.num_envs) return;
  if (mask && !mask[env]) return;
  Ws<NL>& w = ws[grp];
  CReg<NL> cr;
  real invm = 0;
  if (sub >= NL && sub < NV) invm = sub < NL + 3 ? 1.0 / m->cube_mass : 1.0 / m->cube_inertia[sub - NL - 3];
  int episode = st.episode[env] + 1;
  Prof pf;
  pf.start();
  init_ws<NL>(w, sub);
  reset_env<NL, G, SOLVER>(w, lm, m, sub, st.seed, st.env_id_offset + env, episode, cr, invm, pf);
  if (obs) write_obs<NL, G>(w, lm, m, sub, obs + (size_t)env * m->obs_dim);
  if (sub == 0) { st.step_idx[env] = 0; st.episode[env] = episode; st.contact_mask[env] = 0; if (st.sim_time) st.sim_time[env] = 0; }
  GSYNC();
  store_state<NL, G>(w, st, env, sub);
}

// get_observation + get_reward of the CURRENT state (no step): what a trailing mj_step1 and the two task methods give
template <int NL, int G, int EPB>
__global__ __launch_bounds__(64) void k_observe(const KDeviceModel* __restrict__ dm, KDeviceState st, double* __restrict__ obs,
                                                double* __restrict__ reward) {
  __shared__ Ws<NL> ws[EPB];
  __shared__ LModel<NL> lm;
  stage_model<NL>(lm, dm);
  const KModelDesc* m = &dm->d;
  const int lane = threadIdx.x, grp = lane / G, sub = lane % G;
  const int env = xcd_block(blockIdx.x, gridDim.x) * EPB + grp;
  if (grp >= EPB || env >= st.num_envs) return;
  Ws<NL>& w = ws[grp];
  init_ws<NL>(w, sub);
  load_state<NL, G>(w, st, env, sub, false);
  GSYNC();
  // a state restored from a diverged checkpoint: what k_step reports for such an env -- zero observation and reward, no contacts
  // (the kinematics of a non-finite state would put garbage into the mask the diagnostics and the next cost sort read)
  int lb = 0;
  for (int i = sub; i < Dim<NL>::NQ; i += G) lb |= !isfinite(w.qpos[i]);
  for (int i = sub; i < Dim<NL>::NV; i += G) lb |= !isfinite(w.qvel[i]);
  if (gor<G>(lb)) {
    if (obs) for (int i = sub; i < m->obs_dim; i += G) obs[(size_t)env * m->obs_dim + i] = 0;
    if (sub == 0) { st.contact_mask[env] = 0; if (reward) reward[env] = 0; }
    return;
  }
  fk_parallel<NL, G>(w, lm, sub);
  collide_parallel<NL, G>(w, lm, m, sub);
  const real rew = env_reward<NL, G>(w, m, sub);
  if (obs) write_obs<NL, G>(w, lm, m, sub, obs + (size_t)env * m->obs_dim);
  if (sub == 0) { st.contact_mask[env] = w.contact_mask; if (reward) reward[env] = rew; }
}

template <int NL, int G, int SOLVER, int EPB>
static void launch_step_e(const KDeviceModel* dm, const KDeviceState& st, const float* act, double* obs, double* reward, uint8_t* done, int nchunk, hipStream_t stream) {
  if (nchunk > 1) {
    if constexpr (EPB == 64 / G) hipLaunchKernelGGL((k_step<NL, G, SOLVER, EPB, true>), dim3((st.num_envs + EPB - 1) / EPB), dim3(64), 0, stream, dm, st, act, obs, reward, done, nchunk);
  } else {
    int grid = (st.num_envs + EPB - 1) / EPB;
    if (st.disp_in && st.disp_heavy_epb > 0)       // room for disp_cap heavy envs at disp_heavy_epb per wave next to the light ones at EPB per wave (surplus workgroups exit at once)
      grid = (st.disp_cap + st.disp_heavy_epb - 1) / st.disp_heavy_epb + (st.num_envs - st.disp_cap + EPB - 1) / EPB + 1;
    hipLaunchKernelGGL((k_step<NL, G, SOLVER, EPB, false>), dim3(grid), dim3(64), 0, stream, dm, st, act, obs, reward, done, 1);
  }
}
template <int NL, int G, int SOLVER, int EPB>
static void launch_reset_e(const KDeviceModel* dm, const KDeviceState& st, const uint8_t* mask, double* obs, hipStream_t stream) {
  hipLaunchKernelGGL((k_reset<NL, G, SOLVER, EPB>), dim3((st.num_envs + EPB - 1) / EPB), dim3(64), 0, stream, dm, st, mask, obs);
}
template <int NL, int G, int SOLVER>
static void launch_step_t(const KDeviceModel* dm, const KDeviceState& st, const float* act, double* obs, double* reward, uint8_t* done, int nchunk, hipStream_t stream) {
  const int epb = km_step_epb(st.num_envs, 64 / G, nchunk);
  if constexpr (64 / G >= 4) if (epb == 4) return launch_step_e<NL, G, SOLVER, 4>(dm, st, act, obs, reward, done, nchunk, stream);
  if (epb == 2) return launch_step_e<NL, G, SOLVER, 2>(dm, st, act, obs, reward, done, nchunk, stream);
  launch_step_e<NL, G, SOLVER, 1>(dm, st, act, obs, reward, done, nchunk, stream);
}
template <int NL, int G, int SOLVER>
static void launch_reset_t(const KDeviceModel* dm, const KDeviceState& st, const uint8_t* mask, double* obs, hipStream_t stream) {
  const int epb = km_pick_epb(st.num_envs, 64 / G);
  if constexpr (64 / G >= 4) if (epb == 4) return launch_reset_e<NL, G, SOLVER, 4>(dm, st, mask, obs, stream);
  if (epb == 2) return launch_reset_e<NL, G, SOLVER, 2>(dm, st, mask, obs, stream);
  launch_reset_e<NL, G, SOLVER, 1>(dm, st, mask, obs, stream);
}
// kmanip_create, once: the LDS image of the model constants into KDeviceModel::staged (one workgroup)
template <int NL>
__global__ __launch_bounds__(64) void k_prepare_model(KDeviceModel* dm) {
  __shared__ LModel<NL> lm;
  { unsigned char* z = reinterpret_cast<unsigned char*>(&lm); for (int i = threadIdx.x; i < (int)sizeof(LModel<NL>); i += 64) z[i] = 0; }      // (padding bytes: defined)
  __syncthreads();
  build_lmodel<NL>(lm, dm);
  const uint4* src = reinterpret_cast<const uint4*>(&lm);
  uint4* dst = reinterpret_cast<uint4*>(dm->staged);
  for (int i = threadIdx.x; i < (int)(sizeof(LModel<NL>) / 16); i += 64) dst[i] = src[i];
}
template <int NL, int G>
static void launch_observe_t(const KDeviceModel* dm, const KDeviceState& st, double* obs, double* reward, hipStream_t stream) {
  constexpr int EPB = 64 / G;
  hipLaunchKernelGGL((k_observe<NL, G, EPB>), dim3((st.num_envs + EPB - 1) / EPB), dim3(64), 0, stream, dm, st, obs, reward);
}
// ---- one (NL, G, SOLVER) variant per translation unit (the Makefile compiles this file four times, in parallel)
#ifndef KM_VAR_NL
#error "compile with -DKM_VAR_NL=<10|20> -DKM_VAR_G=<16|32> -DKM_VAR_SOLVER=<0|1>"
#endif
#define KM_CAT4_(a, b, c, d) a##b##_##c##_##d
#define KM_CAT4(a, b, c, d) KM_CAT4_(a, b, c, d)
void KM_CAT4(kmanip_launch_step_, KM_VAR_NL, KM_VAR_G, KM_VAR_SOLVER)(const KDeviceModel* dm, const KDeviceState& st, const float* act,
                                                                    double* obs, double* reward, uint8_t* done, int nchunk, hipStream_t stream) {
  launch_step_t<KM_VAR_NL, KM_VAR_G, KM_VAR_SOLVER>(dm, st, act, obs, reward, done, nchunk, stream);
}
void KM_CAT4(kmanip_launch_reset_, KM_VAR_NL, KM_VAR_G, KM_VAR_SOLVER)(const KDeviceModel* dm, const KDeviceState& st,
                                                                     const uint8_t* mask, double* obs, hipStream_t stream) {
  launch_reset_t<KM_VAR_NL, KM_VAR_G, KM_VAR_SOLVER>(dm, st, mask, obs, stream);
}
#if KM_VAR_SOLVER == 1      // (solver-independent: one copy per link-count class)
void KM_CAT4(kmanip_launch_prepare_, KM_VAR_NL, KM_VAR_G, KM_VAR_SOLVER)(KDeviceModel* dm, hipStream_t stream) {
  hipLaunchKernelGGL((k_prepare_model<KM_VAR_NL>), dim3(1), dim3(64), 0, stream, dm);
}
void KM_CAT4(kmanip_launch_observe_, KM_VAR_NL, KM_VAR_G, KM_VAR_SOLVER)(const KDeviceModel* dm, const KDeviceState& st, double* obs,
                                                                       double* reward, hipStream_t stream) {
  launch_observe_t<KM_VAR_NL, KM_VAR_G>(dm, st, obs, reward, stream);
}
#endif
