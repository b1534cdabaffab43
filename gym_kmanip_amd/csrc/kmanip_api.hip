// kmanip_api.hip -- host side of the C ABI declared in include/kmanip.h.
// Owns the device model, the struct-of-arrays env state and the launch sequence of one control step:
//   k_prepare (ctrl float32 round trip, qpos_ik = qpos)  ->  k_before_step_coop (decode + IK)  ->  k_step (physics,
//   reward, obs, done, auto-reset).  No CPU fallback exists: every entry point fails loudly without a HIP device.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "kmanip_device.hpp"

#define KM_VERSION "kmanip-hip 0.30 (gfx950, f64)"

static thread_local std::string g_create_error;

struct KHandle_ {
  KModelDesc desc;
  KDeviceModel* dmodel = nullptr;
  KDeviceState st{};
  int device = 0;
  int num_envs = 0;
  std::string err;
  std::vector<hipEvent_t> ev;   // 4 events per timed step
  std::vector<char> ev_render;  // the step rendered in the step (its fourth event was recorded)
  bool timing = false;
  int timing_every = 1;         // events around every timing_every-th step (the argument of kmanip_enable_timing)
  long timing_count = 0;
  double* rd_rec[2] = {nullptr, nullptr};   // kmanip_bind_reward_done_record: the two record buffers
  int rd_sel = 0;                           // kmanip_select_reward_done_record: the one the next kmanip_step fills
  int timed_steps = 0;
  double* qpos_snap[2] = {nullptr, nullptr};   // kmanip_snapshot_render_state: copies of qpos the renders can read instead of the live state
  int render_src = -1;                         // kmanip_set_render_source: -1 = live state, 0 / 1 = that snapshot
  bool last_step_timed = false; // the last kmanip_step recorded its events (the ring was not full): a render that follows may add its leg
  // render target bound to the step (BASELINE config 5: "depth render in the step"): kmanip_step then also renders
  int step_cam = -1, step_h = 0, step_w = 0;
  float* step_depth = nullptr;
  bool ik_unfused = false;      // KMANIP_IK_UNFUSED=1: before_step as its own launch (A/B timing only)
  // wave slots in predicted-cost order (k_sort_envs) for launches of several residency rounds; KMANIP_COST_SORT=0 / 1 overrides
  int32_t* slot_env = nullptr;
  bool cost_sort = false;
  KCostWeights cost_w{18, 1, 2000, 0, 0, 100};     // work units per: IK evaluation, Newton work unit, collider near the cube; bin width
  // heavy-first dispatch (KDeviceState::disp_*): three rotating tables, the step counter that rotates them
  uint8_t* spread_flags[2] = {nullptr, nullptr};   // SPREAD (KDeviceState::spread_*): the two flag arrays, rotated by every single-step launch
  unsigned spread_k = 0;
  int32_t* disp_tab[3] = {nullptr, nullptr, nullptr};
  unsigned disp_k = 0;
  int wave_slots = 0;           // entries of st.wave_clk (one per lane group of the step launch's grid)
  int last_epb = 0;             // envs per wave of the LAST step launch (a chunk launch always takes the full shape): what the slot -> env maps of kmanip_dbg_wave_clocks are rebuilt with
  std::vector<void*> allocs;
};

// Every entry point works on the handle's device and leaves the caller's current device as it found it (a
// multi-device process -- e.g. torch with several GPUs -- must not have its current device changed under it).
struct DevGuard {
  int prev = -1;
  bool ok = true;
  explicit DevGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) ok = hipSetDevice(dev) == hipSuccess; else prev = -1;
  }
  ~DevGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
#define KM_ENTER(h)                                                                         \
  DevGuard dev_guard_((h)->device);                                                         \
  if (!dev_guard_.ok) { (h)->err = "hipSetDevice failed"; return -1; }

// device scratch freed on every exit path
struct DevBuf {
  void* p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  template <class T> T* as() const { return (T*)p; }
};

#define HIPCHK(h, call)                                                                     \
  do {                                                                                      \
    hipError_t e_ = (call);                                                                 \
    if (e_ != hipSuccess) {                                                                 \
      (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                         \
      return (int)e_ ? (int)e_ : -1;                                                        \
    }                                                                                       \
  } while (0)

static int validate(const KModelDesc* d, std::string& err) {
  if (d->ik_max_nfev < 0) { err = "ik_max_nfev must be 0 (the reference's 100 n) or a positive cap"; return -1; }
  if (d->nlink != 10 && d->nlink != 20) { err = "nlink must be 10 or 20 (KManipSoloArm / DualArm / Torso)"; return -1; }
  if (d->nsphere < 0 || d->nsphere > KM_MAX_SPHERES || d->nsphere > 6 * (d->nlink / 10)) { err = "too many collision spheres (at most 6 per 10 links: one collision lane each)"; return -1; }
  for (int s = 0; s < d->nsphere; s++)
    if (d->sphere_link[s] < 0 || d->sphere_link[s] >= d->nlink || !(d->sphere_radius[s] > 0)) { err = "collision sphere on a missing link or with radius <= 0"; return -1; }
  if (d->obs_dim != 2 * d->nlink + 7) { err = "obs_dim != 2*nlink+7"; return -1; }
  // (+-INFINITY is the infinite plane; a NaN bound would pass the depth renderer's min-chain rectangle test)
  if (std::isnan(d->table_rect[0]) || std::isnan(d->table_rect[1]) || std::isnan(d->table_rect[2]) || std::isnan(d->table_rect[3]) ||
      !(d->table_rect[0] <= d->table_rect[1]) || !(d->table_rect[2] <= d->table_rect[3])) { err = "table_rect must be x_lo <= x_hi, y_lo <= y_hi (no NaN; +-INFINITY = infinite plane)"; return -1; }
  if (d->n_sub_steps < 1 || d->solver_iterations < 0) { err = "bad n_sub_steps / solver_iterations"; return -1; }
  for (const double* si : {d->con_def_solimp, d->con_cube_solimp}) {
    const double power = si[4] < 1 ? 1 : si[4];
    if (power != 1 && power != 2) { err = "solimp power must be 1 or 2 (MuJoCo's default is 2; the device impedance spline has no pow)"; return -1; }
  }
  for (int i = 0; i < d->nlink; i++)
    if (d->link_parent[i] >= i || d->link_parent[i] < -1) { err = "links must be ordered parents-first"; return -1; }
  int nik = 0;
  for (int a = 0; a < KM_MAX_ARMS; a++) {
    if (!d->arm_present[a]) continue;
    if (d->arm_nq[a] != 6 && d->arm_nq[a] != 7) { err = "arm_nq must be 6 or 7"; return -1; }
    if (nik && nik != d->arm_nq[a]) { err = "both arms must have the same number of IK unknowns"; return -1; }
    nik = d->arm_nq[a];
  }
  return 0;
}

// ancestor masks and IK chains (derived data, kept out of the ABI struct)
static void h_quat2mat(const double* q, double* m) {
  double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  double w = q[0] / n, x = q[1] / n, y = q[2] / n, z = q[3] / n;
  m[0] = w * w + x * x - y * y - z * z; m[1] = 2 * (x * y - w * z); m[2] = 2 * (x * z + w * y);
  m[3] = 2 * (x * y + w * z); m[4] = w * w - x * x + y * y - z * z; m[5] = 2 * (y * z - w * x);
  m[6] = 2 * (x * z - w * y); m[7] = 2 * (y * z + w * x); m[8] = w * w - x * x - y * y + z * z;
}

static int build_aux(const KModelDesc* d, KModelAux* x, std::string& err) {
  memset(x, 0, sizeof(*x));
  for (int i = 0; i < d->nlink; i++)
    if (d->jnt_axis[i][0] != 0 || d->jnt_axis[i][1] != 0 || d->jnt_axis[i][2] != 1) {
      err = "the IK kernel assumes joint axes along local z (true for every reference model: axis=\"0 0 1\")"; return -1;
    }
  for (int i = 0; i < d->nlink; i++) {
    uint32_t mk = 0;
    for (int j = i; j >= 0; j = d->link_parent[j]) mk |= 1u << j;
    x->anc_mask[i] = mk;
  }
  int depth = 1;
  for (int i = 0; i < d->nlink; i++) {
    int dep = 0;
    for (int j = i; j >= 0; j = d->link_parent[j]) { x->desc_mask[j] |= 1u << i; dep++; }
    if (dep > depth) depth = dep;
    for (int k = 0; k < 4; k++) {
      int a = i;
      for (int t = 0; t < (1 << k) && a >= 0; t++) a = d->link_parent[a];
      x->jump[k][i] = a;
    }
  }
  if (depth > 16) { err = "kinematic tree deeper than 16 links"; return -1; }
  x->fk_rounds = 0;
  while ((1 << x->fk_rounds) < depth) x->fk_rounds++;
  // block split of the joint-space inertia: the most balanced s such that no link >= s has an ancestor < s
  x->split = 0;
  if (!getenv("KMANIP_NO_BLOCK_SPLIT")) {
    int best = d->nlink + 1;
    for (int s = 1; s < d->nlink; s++) {
      bool ok = true;
      for (int i = s; i < d->nlink && ok; i++) ok = (x->anc_mask[i] & ((1u << s) - 1u)) == 0;
      const int big = s > d->nlink - s ? s : d->nlink - s;
      // only the two layouts the device code is written for (its row-per-block paths pick the second block's columns at
      // compile-time offsets 10 or 11): DualArm 10 + 10, Torso 11 + 9.  Any other forest runs the full two-row code (split = 0).
      if (ok && big <= KM_BLOCK_MAX && big < best && (s == 10 || s == 11)) { best = big; x->split = s; }
    }
  }
  if (x->split != 0 && x->split != 10 && x->split != 11) { err = "internal: block split other than 10 / 11"; return -1; }
  for (int i = 0; i < d->nlink; i++) {
    double q[4] = {d->link_quat[i][0], d->link_quat[i][1], d->link_quat[i][2], d->link_quat[i][3]};
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (!(n > 0)) { err = "zero link quaternion"; return -1; }
    for (int c = 0; c < 4; c++) q[c] /= n;
    h_quat2mat(q, x->link_R[i]);
  }
  for (int c = 0; c < KM_MAX_CAMS; c++) x->cam_tanhalf[c] = d->cam_present[c] ? tan(0.5 * d->cam_fovy[c] * (M_PI / 180.0)) : 1.0;
  for (int s = 0; s < d->nsphere; s++)
    if (d->sphere_visible[s]) {
      if (x->nvis == KM_RENDER_MAXVIS) { err = "the camera renders draw at most 4 visible spheres (the finger tips)"; return -1; }
      x->vis_sphere[x->nvis++] = s;
    }
  for (int a = 0; a < KM_MAX_ARMS; a++) {
    if (!d->arm_present[a]) continue;
    int chain[KM_MAX_LINKS], n = 0;
    for (int j = d->arm_site_link[a]; j >= 0; j = d->link_parent[j]) chain[n++] = j;
    if (n > KM_MAX_CHAIN) { err = "IK chain longer than KM_MAX_CHAIN"; return -1; }
    if (n > d->arm_nq[a] + 1) { err = "at most one fixed joint may follow the IK unknowns on the site's chain"; return -1; }
    x->chain_len[a] = n;
    h_quat2mat(d->arm_site_quat[a], x->site_R[a]);
    for (int k = 0; k < n; k++) {
      int l = chain[n - 1 - k];
      x->chain_link[a][k] = l;
      int xi = -1;
      for (int i = 0; i < d->arm_nq[a]; i++) if (d->arm_q_id[a][i] == l) xi = i;
      x->chain_xidx[a][k] = xi;
      h_quat2mat(d->link_quat[l], x->chain_R[a][k]);
      // the kernels rely on: unknown i sits at chain position i, fixed joints (if any) come after
      if ((k < d->arm_nq[a]) != (xi == k)) { err = "IK mask must be the leading links of the site's chain, in order"; return -1; }
    }
  }
  return 0;
}

extern "C" {

int kmanip_model_desc_size(void) { return (int)sizeof(KModelDesc); }
const char* kmanip_version(void) { return KM_VERSION; }

int kmanip_create(const KModelDesc* desc, int num_envs, int device, uint64_t seed, int64_t env_id_offset, KHandle* out) {
  if (!desc || !out || num_envs <= 0) { g_create_error = "kmanip_create: bad arguments"; return -1; }
  KHandle_* h = new (std::nothrow) KHandle_();
  if (!h) { g_create_error = "out of memory"; return -1; }
  h->desc = *desc;
  h->device = device;
  h->num_envs = num_envs;
  KDeviceModel hm;
  hm.d = *desc;
  if (validate(desc, h->err) != 0 || build_aux(desc, &hm.x, h->err) != 0) { g_create_error = h->err; delete h; return -2; }
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0 || device >= ndev) {
    g_create_error = std::string("kmanip_create: no usable HIP device (") + hipGetErrorString(e) + "); this library has no CPU path";
    delete h; return -3;
  }
#define CR(call) do { hipError_t e2 = (call); if (e2 != hipSuccess) { g_create_error = std::string(#call) + ": " + hipGetErrorString(e2); kmanip_destroy(h); return -4; } } while (0)
  DevGuard dev_guard_(device);
  if (!dev_guard_.ok) { g_create_error = "kmanip_create: hipSetDevice failed"; delete h; return -4; }
  const int nl = desc->nlink, nv = nl + 6, nq = nl + 7;
  const size_t N = (size_t)num_envs;
  auto dalloc = [&](void** p, size_t bytes) -> hipError_t {
    hipError_t r = hipMalloc(p, bytes);
    if (r == hipSuccess) { h->allocs.push_back(*p); r = hipMemset(*p, 0, bytes); }
    return r;
  };
  CR(dalloc((void**)&h->dmodel, sizeof(KDeviceModel)));
  CR(hipMemcpy(h->dmodel, &hm, sizeof(KDeviceModel), hipMemcpyHostToDevice));
  kmanip_launch_prepare_model(h->dmodel, *desc, nullptr);       // the kernels' LDS image of the model constants, built once (KDeviceModel::staged)
  CR(hipGetLastError());
  CR(dalloc((void**)&h->st.qpos, sizeof(double) * nq * N));
  CR(dalloc((void**)&h->st.qvel, sizeof(double) * nv * N));
  CR(dalloc((void**)&h->st.ctrl, sizeof(double) * nl * N));
  CR(dalloc((void**)&h->st.warm, sizeof(double) * nv * N));
  CR(dalloc((void**)&h->st.qpos_ik, sizeof(double) * nl * N));
  CR(dalloc((void**)&h->st.step_idx, sizeof(int32_t) * N));
  CR(dalloc((void**)&h->st.episode, sizeof(int32_t) * N));
  CR(dalloc((void**)&h->st.contact_mask, sizeof(uint32_t) * N));
  CR(dalloc((void**)&h->st.ik_nfev, sizeof(int32_t) * 2 * N));
  CR(dalloc((void**)&h->st.ik_status, sizeof(int32_t) * 2 * N));
  // episode counter starts at -1 so that the first reset is episode 0 (matches the oracle's ko_reset(..., 0))
  CR(hipMemset(h->st.episode, 0xFF, sizeof(int32_t) * N));
  h->st.num_envs = num_envs;
  h->st.env_id_offset = env_id_offset;
  h->st.seed = seed;
  h->st.sim_time = nullptr;
  h->st.rd_rec = nullptr;
  h->st.control_dt = desc->n_sub_steps * desc->timestep;
  { const char* e = getenv("KMANIP_IK_UNFUSED"); h->ik_unfused = e && e[0] == '1'; }
  h->st.slot_env = nullptr;
  h->st.wave_clk = nullptr;
  h->st.spread_in = nullptr; h->st.spread_out = nullptr;
  // "near the cube": what flags an env heavy (SPREAD) / sets the sort's proximity bit (k_sort_envs).  Swept on one box
  // (profiles/r05_near_margin.txt): the single-arm launch is best at 1.5 cm (7.18 M; 1 cm 7.17, 2.5 cm 7.14, 0 = in contact only 7.01);
  // the two-arm sort at 2.5-3 cm on the DualArm (3.91 -> 3.97 M) and flat on the Torso (4.86 / 4.85 / 4.83 M at 1.5 / 2.5 / 4 cm)
  h->st.near_margin = nl > 10 ? 0.025 : 0.015;
  if (const char* e = getenv("KMANIP_NEAR_MARGIN")) { const double v = atof(e); if (v >= 0 && v < 1) h->st.near_margin = v; }
  h->st.disp_in = nullptr; h->st.disp_out = nullptr; h->st.disp_zero = nullptr; h->st.disp_cap = 0; h->st.disp_heavy_epb = 1;
  h->wave_slots = num_envs;
  {
    // Which envs share a wave (the single-arm Newton kernel at widths whose launch is about one residency round of multi-env waves).
    // Every step notes which envs end it with a collider on or within 1.5 cm of the cube ("heavy": 12 % of the envs, 99 % of the
    // next step's coupled ones); the next launch uses that to choose its wave-mates.  An env's bits do not depend on its slot or
    // its wave-mates (tests), so this is scheduling only.
    //  * SPREAD (default at >= 2048 envs: two or four envs per wave; KMANIP_SPREAD=0 turns it off): flags, one byte per env
    //    (KDeviceState::spread_*: bit 0 heavy; bits 1-2 a cost score of the others: a cube that does not rest on four corners, a
    //    sphere on the table).  A wave reads the 64 flags of its block of 64 consecutive envs (three ballots) and the block's waves
    //    deal its envs out (spread_pick): the j-th heavy env to lane group 0 of wave j, so that no wave holds two heavy ones (a wave
    //    with two coupled envs runs the joint loop for the longer of their iteration counts: those waves ended the launches); the
    //    others in ascending score order, so that the heavy waves' other groups take the block's plainest envs and envs of a kind
    //    sit together in the later waves.  k_step 0.5896 -> 0.5690 ms at 4096 envs (heavy-only flags 0.5751, + table bit 0.5716),
    //    0.5361 -> 0.5302 ms at 2048 (profiles/r05_spread_dispatch.txt).  A permutation INSIDE each block whatever the flags
    //    say: the cache lines a block touches are those of the identity map (a first version dealt from launch-wide lists filled by
    //    atomics in completion order: 0.5853 ms, HBM traffic 5.6 -> 17 MB a launch).
    //  * HEAVY-FIRST with variable occupancy (experiment, KMANIP_HEAVY_DISPATCH=1 KMANIP_HEAVY_EPB=1|2|4|0, KMANIP_HEAVY_CAP = most
    //    envs dispatched as heavy, default num_envs / 16): launch-wide lists (KDeviceState::disp_*), heavy envs first and
    //    KMANIP_HEAVY_EPB to a wave (0: the list-based spread).  Loses (profiles/r05_heavy_dispatch.txt): a coupled env
    //    alone in a wave ends after 1.0-1.2 M cycles against 1.4-1.5 M in a four-env wave, but 12 % flagged envs are 35 % more waves
    //    than SIMD slots, whose late starters end last.
    //  * KMANIP_HEAVY_DISPATCH=0: neither; the identity map.
    const bool single_newton = nl == 10 && desc->solver == KM_SOLVER_NEWTON;
    bool spread = single_newton && num_envs >= 2048 && num_envs % 64 == 0 && !getenv("KMANIP_HEAVY_DISPATCH");
    if (const char* e = getenv("KMANIP_SPREAD")) spread = spread && e[0] == '1';
    if (const char* e = getenv("KMANIP_COST_SORT")) if (e[0] == '1') spread = false;      // (the sorted slot order is a map of its own)
    if (spread) for (int t = 0; t < 2; t++) CR(dalloc((void**)&h->spread_flags[t], (size_t)num_envs));
    h->st.spread_table = 3;
    if (const char* e = getenv("KMANIP_SPREAD_TABLE")) h->st.spread_table = atoi(e);
    bool on = false;
    int hepb = 1;
    if (const char* e = getenv("KMANIP_HEAVY_DISPATCH")) on = single_newton && num_envs >= 2048 && e[0] == '1';
    if (on) {
      int cap = num_envs / 16 > 64 ? num_envs / 16 : 64;
      if (const char* e = getenv("KMANIP_HEAVY_CAP")) { const int v = atoi(e); if (v > 0) cap = v; }
      if (const char* e = getenv("KMANIP_HEAVY_EPB")) { const int v = atoi(e); if (v == 0 || v == 1 || v == 2 || v == 4) hepb = v; }
      if (hepb == 0) {                 // SPREAD: at most one heavy env per wave, every wave of the plain grid can take one
        const int waves = num_envs / (num_envs >= 4096 ? 4 : 2);
        if (!getenv("KMANIP_HEAVY_CAP") || cap > waves) cap = waves;
      }
      if (cap > num_envs) cap = num_envs;
      h->st.disp_cap = cap; h->st.disp_heavy_epb = hepb;
      std::vector<int32_t> init(KM_DISP_HDR + N, 0);
      init[1] = num_envs;                                   // table 0: nobody heavy, the light list is the identity
      for (int i = 0; i < num_envs; i++) init[KM_DISP_HDR + i] = i;
      for (int t = 0; t < 3; t++) {
        CR(dalloc((void**)&h->disp_tab[t], sizeof(int32_t) * (KM_DISP_HDR + N)));
        if (t == 0) CR(hipMemcpy(h->disp_tab[0], init.data(), sizeof(int32_t) * (KM_DISP_HDR + N), hipMemcpyHostToDevice));
      }
      if (hepb > 0) h->wave_slots = 4 * (cap + num_envs + 4);      // (an upper bound of 4 lane groups x the grid of any launch shape; SPREAD keeps the plain grid)
    }
  }
  if (const char* e = getenv("KMANIP_WAVE_CLOCKS")) if (e[0] == '1') CR(dalloc((void**)&h->st.wave_clk, sizeof(unsigned long long) * h->wave_slots));
  CR(dalloc((void**)&h->slot_env, sizeof(int32_t) * N));
  CR(dalloc((void**)&h->st.work, sizeof(int32_t) * N));
  {
    // more waves than SIMD slots (1024) at two envs per wave: the two-arm models (their kernels carry the work counters the
    // order is predicted from; the single-arm kernel ships without them -- KMANIP_COST_SORT=1 still sorts it by its IK counts)
    h->cost_sort = nl > 10 && num_envs > 2048;
    if (const char* e = getenv("KMANIP_COST_SORT")) h->cost_sort = e[0] == '1';
    if (const char* e = getenv("KMANIP_COST_W")) {       // diagnostic: "ik,work,near-cube,armtab,cubetab,binwidth"
      KCostWeights w = h->cost_w;
      // (negative weights would make a cost negative; the kernel clamps the bin, and they are refused here)
      if (sscanf(e, "%d,%d,%d,%d,%d,%d", &w.ik, &w.work, &w.coupled, &w.armtab, &w.cubetab, &w.binw) == 6 && w.binw > 0 &&
          w.ik >= 0 && w.work >= 0 && w.coupled >= 0 && w.armtab >= 0 && w.cubetab >= 0) h->cost_w = w;
      else { g_create_error = "kmanip_create: KMANIP_COST_W must be six comma-separated integers >= 0 with a bin width > 0"; kmanip_destroy(h); return -2; }
    }
  }
  // the initialisation above ran on the null stream; the caller's (non-blocking) streams must not start before it
  CR(hipDeviceSynchronize());
#undef CR
  *out = h;
  return 0;
}

void kmanip_destroy(KHandle h) {
  if (!h) return;
  DevGuard dev_guard_(h->device);
  for (void* p : h->allocs) (void)hipFree(p);
  for (hipEvent_t e : h->ev) (void)hipEventDestroy(e);
  delete h;
}

const char* kmanip_last_error(KHandle h) { return h ? h->err.c_str() : g_create_error.c_str(); }
int kmanip_num_envs(KHandle h) { return h ? h->num_envs : 0; }

int kmanip_reset(KHandle h, const uint8_t* mask_dev, double* obs_dev, void* stream) {
  if (!h) return -1;
  KM_ENTER(h);
  kmanip_launch_reset(h->dmodel, h->desc, h->st, mask_dev, 0, obs_dev, (hipStream_t)stream);
  HIPCHK(h, hipGetLastError());
  return 0;
}

// Diagnostics (include/kmanip_debug.h, not part of the boundary; KMANIP_WAVE_CLOCKS=1 at create): per wave slot, the ticks its wave spent in the last
// k_step and the env it held -- HOST arrays of kmanip_dbg_wave_slots(h) entries --, and every env's work counter (num_envs entries).  Synchronous.
int kmanip_dbg_wave_clocks(KHandle h, unsigned long long* clk, int32_t* slot_env, int32_t* work) {
  if (!h) return -1;
  if (clk && !h->st.wave_clk) { h->err = "kmanip_dbg_wave_clocks: clk needs KMANIP_WAVE_CLOCKS=1 at create"; return -1; }
  KM_ENTER(h);
  HIPCHK(h, hipDeviceSynchronize());
  const size_t N = (size_t)h->num_envs;
  if (clk) HIPCHK(h, hipMemcpy(clk, h->st.wave_clk, sizeof(unsigned long long) * h->wave_slots, hipMemcpyDeviceToHost));
  if (slot_env) {
    if (h->disp_tab[0]) {
      // heavy-first dispatch: rebuild the LAST launch's slot -> env map (4 slots per workgroup, -1 = empty lane group) from its table
      std::vector<int32_t> tab(KM_DISP_HDR + N);
      HIPCHK(h, hipMemcpy(tab.data(), h->disp_tab[(h->disp_k + 2) % 3], sizeof(int32_t) * (KM_DISP_HDR + N), hipMemcpyDeviceToHost));
      const int epb = h->last_epb ? h->last_epb : km_step_epb(h->num_envs, 4, 1), hepb = h->st.disp_heavy_epb;
      const int nh = tab[0] < h->st.disp_cap ? tab[0] : h->st.disp_cap, nhw = hepb ? (nh + hepb - 1) / hepb : 0;
      for (int sl = 0; sl < h->wave_slots; sl++) {
        const int b = sl / epb, grp = sl % epb;
        int idx = -1;
        if (hepb == 0) { const int i = b < nh ? (grp == 0 ? b : nh + (epb - 1) * b + grp - 1) : epb * b + grp; if (i < (int)N) idx = i; }
        else if (b < nhw) { if (grp < hepb && b * hepb + grp < nh) idx = b * hepb + grp; }
        else { const int i = nh + (b - nhw) * epb + grp; if (i < (int)N) idx = i; }
        slot_env[sl] = idx >= 0 ? tab[KM_DISP_HDR + idx] : -1;
      }
    } else if (h->spread_flags[0]) {
      // SPREAD: the LAST launch's map from the flags it read (slot = wave index in slot space x envs per wave + lane group)
      std::vector<uint8_t> fl(N);
      HIPCHK(h, hipMemcpy(fl.data(), h->spread_flags[(h->spread_k + 1) & 1], N, hipMemcpyDeviceToHost));
      const int epb = h->last_epb ? h->last_epb : km_step_epb(h->num_envs, 4, 1);      // (a chunk launch at 2048 envs runs four envs per wave, a step two)
      for (size_t blk = 0; blk < N / 64; blk++) {
        unsigned long long M = 0, S1 = 0, S2 = 0;
        for (int i = 0; i < 64; i++) {
          const unsigned long long f = fl[blk * 64 + i];
          M |= (f & 1) << i; S1 |= ((f >> 1) & 1) << i; S2 |= ((f >> 2) & 1) << i;
        }
        for (int j = 0; j < 64 / epb; j++)
          for (int g = 0; g < epb; g++) slot_env[blk * 64 + j * epb + g] = (int32_t)(blk * 64) + spread_pick(M, S1, S2, j, g, epb);
      }
    } else if (h->cost_sort) HIPCHK(h, hipMemcpy(slot_env, h->slot_env, sizeof(int32_t) * N, hipMemcpyDeviceToHost));
    else for (size_t i = 0; i < N; i++) slot_env[i] = (int32_t)i;                     // the identity map
  }
  if (work) HIPCHK(h, hipMemcpy(work, h->st.work, sizeof(int32_t) * N, hipMemcpyDeviceToHost));
  return 0;
}
// number of entries of kmanip_dbg_wave_clocks' clk / slot_env arrays (num_envs, or more with the heavy-first dispatch)
int kmanip_dbg_wave_slots(KHandle h) { return h ? h->wave_slots : 0; }

int kmanip_observe(KHandle h, double* obs_dev, double* reward_dev, void* stream) {
  if (!h) { g_create_error = "kmanip_observe: null handle"; return -1; }
  KM_ENTER(h);
  kmanip_launch_observe(h->dmodel, h->desc, h->st, obs_dev, reward_dev, (hipStream_t)stream);
  HIPCHK(h, hipGetLastError());
  return 0;
}

static int step_impl(KHandle h, int nchunk, const float* act_dev, double* obs_dev, double* reward_dev, uint8_t* done_dev, void* stream) {
  if (!h) { g_create_error = "kmanip_step: null handle"; return -1; }
  if (!act_dev || !obs_dev || !reward_dev || !done_dev) { h->err = "kmanip_step: null buffer"; return -1; }
  KM_ENTER(h);
  hipStream_t s = (hipStream_t)stream;
  const bool tm = h->timing && (h->timing_count++ % h->timing_every) == 0 && h->timed_steps < KM_TIMING_SLOTS;
  h->last_step_timed = tm;
  hipEvent_t* ev = tm ? &h->ev[4 * (size_t)h->timed_steps] : nullptr;
  // product path: ONE launch, before_step (decode + IK) fused into k_step so that no device-wide barrier sits between
  // an env's IK and its physics; the split launches remain for A/B timing (KMANIP_IK_UNFUSED=1)
  const bool split = h->ik_unfused;
  h->st.rd_rec = nullptr;
  if (nchunk == 1 && h->rd_rec[0]) h->st.rd_rec = h->rd_rec[h->rd_sel];     // (the CALLER alternates: kmanip_select_reward_done_record)
  if (split && nchunk != 1) { h->err = "kmanip_step_chunk needs the fused path (unset KMANIP_IK_UNFUSED)"; return -1; }
  if (split) {
    if (tm) HIPCHK(h, hipEventRecord(ev[0], s));
    kmanip_launch_ik_coop(h->dmodel, h->desc, h->st, act_dev, s);
  }
  h->st.slot_env = nullptr;
  if (h->cost_sort) {                        // (one small launch: counting sort of the envs by their last step's diagnostics)
    kmanip_launch_sort_envs(h->st, h->slot_env, h->cost_w, s);
    h->st.slot_env = h->slot_env;
  }
  h->st.spread_in = nullptr; h->st.spread_out = nullptr;
  if (h->spread_flags[0]) {       // SPREAD: this launch (one step or a chunk of them) reads the flags the last launch wrote, and writes the other array
    h->st.spread_in = h->spread_flags[h->spread_k & 1]; h->st.spread_out = h->spread_flags[(h->spread_k + 1) & 1];
    h->spread_k++;
  }
  h->st.disp_in = nullptr; h->st.disp_out = nullptr; h->st.disp_zero = nullptr;
  if (h->disp_tab[0] && nchunk == 1) {       // heavy-first dispatch: this launch reads table k, fills k + 1, clears the counters of k + 2
    h->st.disp_in = h->disp_tab[h->disp_k % 3]; h->st.disp_out = h->disp_tab[(h->disp_k + 1) % 3]; h->st.disp_zero = h->disp_tab[(h->disp_k + 2) % 3];
    h->disp_k++;
  }
  h->last_epb = km_step_epb(h->num_envs, h->desc.nlink <= 10 ? 4 : 2, nchunk);
  if (tm) HIPCHK(h, hipEventRecord(ev[1], s));        // (fused path: two events per step, each costs the stream a barrier packet)
  kmanip_launch_step(h->dmodel, h->desc, h->st, split ? nullptr : act_dev, obs_dev, reward_dev, done_dev, nchunk, s);
  if (tm) HIPCHK(h, hipEventRecord(ev[2], s));
  const bool render = h->step_depth && nchunk == 1;
  if (render)                             // the observation's camera branch (env_sim.py:140-145) of the state just produced
    kmanip_launch_render_depth(h->dmodel, h->st, h->step_cam, h->step_h, h->step_w, h->step_depth, s);
  if (tm) {
    if (render) HIPCHK(h, hipEventRecord(ev[3], s));
    h->ev_render[h->timed_steps] = render;
    h->timed_steps++;
  }
  HIPCHK(h, hipGetLastError());
  return 0;
}

int kmanip_step(KHandle h, const float* act_dev, double* obs_dev, double* reward_dev, uint8_t* done_dev, void* stream) {
  return step_impl(h, 1, act_dev, obs_dev, reward_dev, done_dev, stream);
}

int kmanip_step_chunk(KHandle h, int nsteps, const float* act_dev, double* obs_dev, double* reward_dev, uint8_t* done_dev, void* stream) {
  if (nsteps <= 0) { if (h) h->err = "kmanip_step_chunk: nsteps must be positive"; return -1; }
  return step_impl(h, nsteps, act_dev, obs_dev, reward_dev, done_dev, stream);
}

int kmanip_render_depth(KHandle h, int cam, int height, int width, float* depth_dev, void* stream) {
  if (!h || !depth_dev || cam < 0 || cam >= KM_MAX_CAMS || height <= 0 || width <= 0) { if (h) h->err = "kmanip_render_depth: bad arguments"; return -1; }
  if (!h->desc.cam_present[cam]) { h->err = "kmanip_render_depth: this model has no such camera"; return -1; }
  KM_ENTER(h);
  KDeviceState st = h->st;
  if (h->render_src >= 0) st.qpos = h->qpos_snap[h->render_src];
  kmanip_launch_render_depth(h->dmodel, st, cam, height, width, depth_dev, (hipStream_t)stream);
  HIPCHK(h, hipGetLastError());
  return 0;
}

int kmanip_snapshot_render_state(KHandle h, int slot, void* stream) {
  if (!h) { g_create_error = "kmanip_snapshot_render_state: null handle"; return -1; }
  if (slot < 0 || slot > 1) { h->err = "kmanip_snapshot_render_state: slot must be 0 or 1"; return -1; }
  KM_ENTER(h);
  const size_t bytes = sizeof(double) * (size_t)(h->desc.nlink + 7) * h->num_envs;
  if (!h->qpos_snap[slot]) {
    void* p = nullptr;
    HIPCHK(h, hipMalloc(&p, bytes));
    h->allocs.push_back(p);
    h->qpos_snap[slot] = (double*)p;
  }
  HIPCHK(h, hipMemcpyAsync(h->qpos_snap[slot], h->st.qpos, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return 0;
}

int kmanip_set_render_source(KHandle h, int slot) {
  if (!h) { g_create_error = "kmanip_set_render_source: null handle"; return -1; }
  if (slot < -1 || slot > 1) { h->err = "kmanip_set_render_source: slot must be -1 (live state), 0 or 1"; return -1; }
  if (slot >= 0 && !h->qpos_snap[slot]) { h->err = "kmanip_set_render_source: no snapshot was taken into that slot"; return -1; }
  h->render_src = slot;
  return 0;
}

int kmanip_render_rgb_multi(KHandle h, int ncam, const int* cams, const int* heights, const int* widths, uint8_t* const* rgb_dev,
                            void* stream) {
  if (!h) { g_create_error = "kmanip_render_rgb: null handle"; return -1; }
  if (ncam <= 0 || ncam > KM_MAX_CAMS || !cams || !heights || !widths || !rgb_dev) { h->err = "kmanip_render_rgb: bad arguments"; return -1; }
  KRenderJobs jobs;
  jobs.n = ncam;
  for (int i = 0; i < ncam; i++) {
    if (!rgb_dev[i] || cams[i] < 0 || cams[i] >= KM_MAX_CAMS || heights[i] <= 0 || widths[i] <= 0) { h->err = "kmanip_render_rgb: bad arguments"; return -1; }
    if (!h->desc.cam_present[cams[i]]) { h->err = "kmanip_render_rgb: this model has no such camera"; return -1; }
    jobs.cam[i] = cams[i]; jobs.height[i] = heights[i]; jobs.width[i] = widths[i]; jobs.rgb[i] = rgb_dev[i];
  }
  KM_ENTER(h);
  KDeviceState st = h->st;
  if (h->render_src >= 0) st.qpos = h->qpos_snap[h->render_src];
  kmanip_launch_render_rgb(h->dmodel, st, jobs, (hipStream_t)stream);
  // kernel timing (kmanip_enable_timing): the camera observations rendered right after a timed step are that step's render leg --
  // its start is the event the step recorded after k_step, so the render costs the stream ONE more event, not a pair around it
  // (a render of a SNAPSHOT runs behind the steps, on a stream of its own: it is no leg of the step's stream)
  if (h->timing && h->render_src < 0 && h->last_step_timed && h->timed_steps > 0 && !h->ev_render[h->timed_steps - 1]) {
    HIPCHK(h, hipEventRecord(h->ev[4 * (size_t)(h->timed_steps - 1) + 3], (hipStream_t)stream));
    h->ev_render[h->timed_steps - 1] = 1;
  }
  HIPCHK(h, hipGetLastError());
  return 0;
}

int kmanip_render_rgb(KHandle h, int cam, int height, int width, uint8_t* rgb_dev, void* stream) {
  return kmanip_render_rgb_multi(h, 1, &cam, &height, &width, &rgb_dev, stream);
}

int kmanip_bind_step_depth(KHandle h, int cam, int height, int width, float* depth_dev) {
  if (!h) { g_create_error = "kmanip_bind_step_depth: null handle"; return -1; }
  if (!depth_dev) { h->step_depth = nullptr; h->step_cam = -1; return 0; }
  if (cam < 0 || cam >= KM_MAX_CAMS || height <= 0 || width <= 0 || !h->desc.cam_present[cam]) { h->err = "kmanip_bind_step_depth: bad camera / size"; return -1; }
  h->step_cam = cam; h->step_h = height; h->step_w = width; h->step_depth = depth_dev;
  return 0;
}

int kmanip_scripted_action(KHandle h, float* act_dev, void* stream) {
  if (!h || !act_dev) { if (h) h->err = "kmanip_scripted_action: null buffer"; return -1; }
  if (!h->desc.arm_present[0] || h->desc.act_col[KM_ACT_EER_POS] < 0) {
    h->err = "kmanip_scripted_action: this env id has no eer_pos action (the scripted policy drives the right EE delta)"; return -1;
  }
  KM_ENTER(h);
  kmanip_launch_scripted_action(h->dmodel, h->st, act_dev, (hipStream_t)stream);
  HIPCHK(h, hipGetLastError());
  return 0;
}

int kmanip_sample_action(KHandle h, float* act_dev, int ahead, void* stream) {
  if (!h || !act_dev || ahead < 0) { if (h) h->err = "kmanip_sample_action: null buffer / negative ahead"; return -1; }
  if (h->desc.act_dim > 16) { h->err = "kmanip_sample_action: act_dim > 16"; return -1; }
  KM_ENTER(h);
  kmanip_launch_sample_action(h->dmodel, h->st, act_dev, ahead, (hipStream_t)stream);
  HIPCHK(h, hipGetLastError());
  return 0;
}

int kmanip_enable_timing(KHandle h, int enable) {
  if (!h) return -1;
  KM_ENTER(h);
  if (enable && h->ev.empty()) {
    h->ev.resize(4 * KM_TIMING_SLOTS, nullptr);
    h->ev_render.assign(KM_TIMING_SLOTS, 0);
    for (auto& e : h->ev) HIPCHK(h, hipEventCreate(&e));
  }
  h->timing = enable != 0;
  h->timing_every = enable > 1 ? enable : 1;
  h->timing_count = 0;
  h->timed_steps = 0;
  return 0;
}
int kmanip_timing_summary(KHandle h, double* ik_ms_sum, double* dyn_ms_sum, double* render_ms_sum, int32_t* nsteps) {
  if (!h) return -1;
  KM_ENTER(h);
  HIPCHK(h, hipDeviceSynchronize());
  double a = 0, b = 0, c = 0;
  for (int k = 0; k < h->timed_steps; k++) {
    float m1 = 0, m2 = 0, m3 = 0;
    if (h->ik_unfused) HIPCHK(h, hipEventElapsedTime(&m1, h->ev[4 * k], h->ev[4 * k + 1]));
    HIPCHK(h, hipEventElapsedTime(&m2, h->ev[4 * k + 1], h->ev[4 * k + 2]));
    if (h->ev_render[k]) HIPCHK(h, hipEventElapsedTime(&m3, h->ev[4 * k + 2], h->ev[4 * k + 3]));
    a += m1; b += m2; c += m3;
  }
  if (ik_ms_sum) *ik_ms_sum = a;
  if (dyn_ms_sum) *dyn_ms_sum = b;
  if (render_ms_sum) *render_ms_sum = c;
  if (nsteps) *nsteps = h->timed_steps;
  h->timed_steps = 0;
  return 0;
}

// env-major host <-> component-major device transposes
static int pull(KHandle h, const double* dev, double* host, int ncomp) {
  if (!host) return 0;
  const size_t N = (size_t)h->num_envs;
  std::vector<double> tmp(N * ncomp);
  HIPCHK(h, hipMemcpy(tmp.data(), dev, sizeof(double) * N * ncomp, hipMemcpyDeviceToHost));
  for (size_t e = 0; e < N; e++) for (int k = 0; k < ncomp; k++) host[e * ncomp + k] = tmp[(size_t)k * N + e];
  return 0;
}
static int push(KHandle h, double* dev, const double* host, int ncomp) {
  if (!host) return 0;
  const size_t N = (size_t)h->num_envs;
  std::vector<double> tmp(N * ncomp);
  for (size_t e = 0; e < N; e++) for (int k = 0; k < ncomp; k++) tmp[(size_t)k * N + e] = host[e * ncomp + k];
  HIPCHK(h, hipMemcpy(dev, tmp.data(), sizeof(double) * N * ncomp, hipMemcpyHostToDevice));
  return 0;
}

int kmanip_get_state(KHandle h, double* qpos, double* qvel, double* ctrl, double* qacc_warm, int32_t* step_idx) {
  if (!h) return -1;
  KM_ENTER(h);
  HIPCHK(h, hipDeviceSynchronize());
  const int nl = h->desc.nlink;
  int rc = 0;
  if ((rc = pull(h, h->st.qpos, qpos, nl + 7))) return rc;
  if ((rc = pull(h, h->st.qvel, qvel, nl + 6))) return rc;
  if ((rc = pull(h, h->st.ctrl, ctrl, nl))) return rc;
  if ((rc = pull(h, h->st.warm, qacc_warm, nl + 6))) return rc;
  if (step_idx) HIPCHK(h, hipMemcpy(step_idx, h->st.step_idx, sizeof(int32_t) * h->num_envs, hipMemcpyDeviceToHost));
  return 0;
}
int kmanip_set_state(KHandle h, const double* qpos, const double* qvel, const double* ctrl, const double* qacc_warm,
                     const int32_t* step_idx) {
  if (!h) return -1;
  KM_ENTER(h);
  HIPCHK(h, hipDeviceSynchronize());
  const int nl = h->desc.nlink;
  int rc = 0;
  if ((rc = push(h, h->st.qpos, qpos, nl + 7))) return rc;
  if ((rc = push(h, h->st.qvel, qvel, nl + 6))) return rc;
  if ((rc = push(h, h->st.ctrl, ctrl, nl))) return rc;
  if ((rc = push(h, h->st.warm, qacc_warm, nl + 6))) return rc;
  if (step_idx) HIPCHK(h, hipMemcpy(h->st.step_idx, step_idx, sizeof(int32_t) * h->num_envs, hipMemcpyHostToDevice));
  HIPCHK(h, hipDeviceSynchronize());      // pageable-memory copies may return early; order them before the caller's streams
  return 0;
}

int kmanip_get_episode(KHandle h, int32_t* episode) {
  if (!h || !episode) { if (h) h->err = "kmanip_get_episode: null pointer"; return -1; }
  KM_ENTER(h);
  HIPCHK(h, hipDeviceSynchronize());
  HIPCHK(h, hipMemcpy(episode, h->st.episode, sizeof(int32_t) * h->num_envs, hipMemcpyDeviceToHost));
  return 0;
}
int kmanip_set_episode(KHandle h, const int32_t* episode) {
  if (!h || !episode) { if (h) h->err = "kmanip_set_episode: null pointer"; return -1; }
  KM_ENTER(h);
  HIPCHK(h, hipDeviceSynchronize());
  HIPCHK(h, hipMemcpy(h->st.episode, episode, sizeof(int32_t) * h->num_envs, hipMemcpyHostToDevice));
  HIPCHK(h, hipDeviceSynchronize());
  return 0;
}

int kmanip_bind_sim_time(KHandle h, double* sim_time_dev) {
  if (!h) return -1;
  h->st.sim_time = sim_time_dev;
  return 0;
}

int kmanip_bind_reward_done_record(KHandle h, double* rec0_dev, double* rec1_dev) {
  if (!h) return -1;
  if ((rec0_dev == nullptr) != (rec1_dev == nullptr)) { h->err = "kmanip_bind_reward_done_record: two buffers or none"; return -1; }
  h->rd_rec[0] = rec0_dev; h->rd_rec[1] = rec1_dev; h->rd_sel = 0;
  return 0;
}

int kmanip_select_reward_done_record(KHandle h, int index) {
  if (!h) return -1;
  if (index < 0 || index > 1 || !h->rd_rec[0]) { h->err = "kmanip_select_reward_done_record: index 0 / 1 of two bound buffers"; return -1; }
  h->rd_sel = index;
  return 0;
}

int kmanip_set_seed(KHandle h, uint64_t seed, int restart_episodes) {
  if (!h) return -1;
  KM_ENTER(h);
  h->st.seed = seed;
  if (restart_episodes) {
    HIPCHK(h, hipDeviceSynchronize());
    HIPCHK(h, hipMemset(h->st.episode, 0xFF, sizeof(int32_t) * (size_t)h->num_envs));   // -1: the next reset is episode 0
    HIPCHK(h, hipDeviceSynchronize());
  }
  return 0;
}

int kmanip_get_counters(KHandle h, int32_t* step_idx_dev, int32_t* episode_dev, void* stream) {
  if (!h) return -1;
  KM_ENTER(h);
  const size_t bytes = sizeof(int32_t) * (size_t)h->num_envs;
  if (step_idx_dev) HIPCHK(h, hipMemcpyAsync(step_idx_dev, h->st.step_idx, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  if (episode_dev) HIPCHK(h, hipMemcpyAsync(episode_dev, h->st.episode, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return 0;
}

int kmanip_get_diag(KHandle h, uint32_t* contact_mask, int32_t* ik_nfev, int32_t* ik_status) {
  if (!h) return -1;
  KM_ENTER(h);
  HIPCHK(h, hipDeviceSynchronize());
  const size_t N = (size_t)h->num_envs;
  if (contact_mask) HIPCHK(h, hipMemcpy(contact_mask, h->st.contact_mask, sizeof(uint32_t) * N, hipMemcpyDeviceToHost));
  std::vector<int32_t> tmp(2 * N);
  if (ik_nfev) {
    HIPCHK(h, hipMemcpy(tmp.data(), h->st.ik_nfev, sizeof(int32_t) * 2 * N, hipMemcpyDeviceToHost));
    for (size_t e = 0; e < N; e++) { ik_nfev[2 * e] = tmp[e]; ik_nfev[2 * e + 1] = tmp[N + e]; }
  }
  if (ik_status) {
    HIPCHK(h, hipMemcpy(tmp.data(), h->st.ik_status, sizeof(int32_t) * 2 * N, hipMemcpyDeviceToHost));
    for (size_t e = 0; e < N; e++) { ik_status[2 * e] = tmp[e]; ik_status[2 * e + 1] = tmp[N + e]; }
  }
  return 0;
}

int kmanip_ik(KHandle h, int arm, int n, double* qpos, const double* goal_pos, const double* goal_quat, double* q_out,
              int32_t* nfev, int32_t* status) {
  if (!h) { g_create_error = "kmanip_ik: null handle"; return -1; }
  if (arm < 0 || arm >= KM_MAX_ARMS || !h->desc.arm_present[arm] || n <= 0 || !qpos || !goal_pos || !goal_quat || !q_out) {
    h->err = "kmanip_ik: bad arguments"; return -1;
  }
  KM_ENTER(h);
  const int nq = h->desc.nlink + 7, nik = h->desc.arm_nq[arm];
  DevBuf dq, dgp, dgq, dqo, dnf, dst;        // freed on every exit path
  HIPCHK(h, hipMalloc(&dq.p, sizeof(double) * n * nq));
  HIPCHK(h, hipMalloc(&dgp.p, sizeof(double) * n * 3));
  HIPCHK(h, hipMalloc(&dgq.p, sizeof(double) * n * 4));
  HIPCHK(h, hipMalloc(&dqo.p, sizeof(double) * n * nik));
  HIPCHK(h, hipMalloc(&dnf.p, sizeof(int32_t) * n));
  HIPCHK(h, hipMalloc(&dst.p, sizeof(int32_t) * n));
  HIPCHK(h, hipMemcpy(dq.p, qpos, sizeof(double) * n * nq, hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy(dgp.p, goal_pos, sizeof(double) * n * 3, hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy(dgq.p, goal_quat, sizeof(double) * n * 4, hipMemcpyHostToDevice));
  HIPCHK(h, hipDeviceSynchronize());
  kmanip_launch_ik_coop_standalone(h->dmodel, h->desc, arm, n, dq.as<double>(), dgp.as<double>(), dgq.as<double>(), dqo.as<double>(), dnf.as<int32_t>(), dst.as<int32_t>(), nullptr);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipDeviceSynchronize());
  HIPCHK(h, hipMemcpy(qpos, dq.p, sizeof(double) * n * nq, hipMemcpyDeviceToHost));
  HIPCHK(h, hipMemcpy(q_out, dqo.p, sizeof(double) * n * nik, hipMemcpyDeviceToHost));
  if (nfev) HIPCHK(h, hipMemcpy(nfev, dnf.p, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
  if (status) HIPCHK(h, hipMemcpy(status, dst.p, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
  return 0;
}

int kmanip_ik_eval(KHandle h, int arm, int n, const double* qpos, const double* goal_pos, const double* goal_quat,
                   double* res, double* jac) {
  if (!h) { g_create_error = "kmanip_ik_eval: null handle"; return -1; }
  if (arm < 0 || arm >= KM_MAX_ARMS || !h->desc.arm_present[arm] || n <= 0 || !qpos || !goal_pos || !goal_quat || !res || !jac) {
    h->err = "kmanip_ik_eval: bad arguments"; return -1;
  }
  KM_ENTER(h);
  const int nq = h->desc.nlink + 7, nik = h->desc.arm_nq[arm], mrow = 6 + 2 * nik;
  DevBuf dq, dgp, dgq, dres, djac;
  HIPCHK(h, hipMalloc(&dq.p, sizeof(double) * n * nq));
  HIPCHK(h, hipMalloc(&dgp.p, sizeof(double) * n * 3));
  HIPCHK(h, hipMalloc(&dgq.p, sizeof(double) * n * 4));
  HIPCHK(h, hipMalloc(&dres.p, sizeof(double) * n * mrow));
  HIPCHK(h, hipMalloc(&djac.p, sizeof(double) * n * mrow * nik));
  HIPCHK(h, hipMemcpy(dq.p, qpos, sizeof(double) * n * nq, hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy(dgp.p, goal_pos, sizeof(double) * n * 3, hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy(dgq.p, goal_quat, sizeof(double) * n * 4, hipMemcpyHostToDevice));
  HIPCHK(h, hipDeviceSynchronize());
  kmanip_launch_ik_eval_coop(h->dmodel, h->desc, arm, n, dq.as<double>(), dgp.as<double>(), dgq.as<double>(), dres.as<double>(), djac.as<double>(), nullptr);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipDeviceSynchronize());
  HIPCHK(h, hipMemcpy(res, dres.p, sizeof(double) * n * mrow, hipMemcpyDeviceToHost));
  HIPCHK(h, hipMemcpy(jac, djac.p, sizeof(double) * n * mrow * nik, hipMemcpyDeviceToHost));
  return 0;
}

}  // extern "C"
