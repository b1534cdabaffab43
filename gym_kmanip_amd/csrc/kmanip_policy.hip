// kmanip_policy.hip -- the scripted "move towards the cube" policy of the reference's synthetic-data example
// (gym_kmanip/examples/2_synthetic_data.py:28-41), evaluated for every env on device so that data generation needs
// no device->host round trip per step:  action["eer_pos"] = (cube_pos - eer_site_xpos) / |.|.
// One lane per env (a 10-link serial FK is ~1 k FLOP; 4096 envs = 64 waves): the kernel is launch-latency sized, not
// a hot path, and reads the struct-of-arrays state with coalesced columns.
#include "kmanip_device.hpp"

__global__ __launch_bounds__(64) void k_scripted_action(const KDeviceModel* __restrict__ dm, KDeviceState st, float* __restrict__ act) {
  const KModelDesc* m = &dm->d;
  const int env = blockIdx.x * 64 + threadIdx.x, NE = st.num_envs;
  if (env >= NE) return;
  const int arm = 0;                                        // right arm: "eer_site_pos"
  const int clen = dm->x.chain_len[arm];
  real pos[3] = {0, 0, 0}, mat[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  for (int k = 0; k < clen; k++) {
    const int l = dm->x.chain_link[arm][k];
    const double* Rl = dm->x.chain_R[arm][k];
    real lp[3] = {m->link_pos[l][0], m->link_pos[l][1], m->link_pos[l][2]}, t[3], R1[9];
    mat_vec3(t, mat, lp);
    pos[0] += t[0]; pos[1] += t[1]; pos[2] += t[2];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) R1[3 * i + j] = mat[3 * i] * Rl[j] + mat[3 * i + 1] * Rl[3 + j] + mat[3 * i + 2] * Rl[6 + j];
    const real q = st.qpos[(size_t)l * NE + env];
    if (m->jnt_type[l] == KM_JNT_SLIDE) {
      for (int i = 0; i < 9; i++) mat[i] = R1[i];
      pos[0] += R1[2] * q; pos[1] += R1[5] * q; pos[2] += R1[8] * q;
    } else {
      real sn, cs;
      sincos(q, &sn, &cs);
      for (int i = 0; i < 3; i++) {
        mat[3 * i] = cs * R1[3 * i] + sn * R1[3 * i + 1];
        mat[3 * i + 1] = cs * R1[3 * i + 1] - sn * R1[3 * i];
        mat[3 * i + 2] = R1[3 * i + 2];
      }
    }
  }
  real so[3] = {m->arm_site_pos[arm][0], m->arm_site_pos[arm][1], m->arm_site_pos[arm][2]}, sp[3];
  mat_vec3(sp, mat, so);
  const int nl = m->nlink;
  real d[3];
  for (int c = 0; c < 3; c++) d[c] = st.qpos[(size_t)(nl + c) * NE + env] - (sp[c] + pos[c]);
  const real inv = 1.0 / sqrt(dot3(d, d));                  // (numpy: raw_action /= np.linalg.norm(raw_action))
  float* a = act + (size_t)env * m->act_dim + m->act_col[KM_ACT_EER_POS];
  a[0] = (float)(d[0] * inv); a[1] = (float)(d[1] * inv); a[2] = (float)(d[2] * inv);
}

void kmanip_launch_scripted_action(const KDeviceModel* dm, const KDeviceState& st, float* act, hipStream_t stream) {
  hipLaunchKernelGGL(k_scripted_action, dim3((st.num_envs + 63) / 64), dim3(64), 0, stream, dm, st, act);
}

// ---------------------------------------------------------------------------------------------
// action_space.sample() for every env, on device: what the reference's rollout loops feed env.step with
// (examples/2_log_with_h5py.py:22-26, 3_save_to_video.py:20-27: `action = env.action_space.sample()`, a Dict of Box(-1, 1,
// float32) keys, env_base.py:151-188).  Counter-based: Philox4x32-10 keyed by the handle's seed, counter = (global env id,
// episode, KM_ACT_CTR3(step, block)) -- SURVEY 8d's stream, so the CPU oracle (ko_sample_action) draws identical bits for
// the same (seed, env, episode, step) and a run's actions do not depend on the shard layout.  `ahead` control steps into the
// future under the TimeLimit-only episode structure (the reference never terminates early: every episode is exactly
// max_episode_steps long), so a caller can lay out the actions of the next K steps before stepping (bench.py).
// One lane per (env, block of 4 columns).
__global__ __launch_bounds__(256) void k_sample_action(const KDeviceModel* __restrict__ dm, KDeviceState st, float* __restrict__ act, int ahead) {
  const KModelDesc* m = &dm->d;
  const int nblk = (m->act_dim + 3) / 4;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= st.num_envs * nblk) return;
  const int env = i / nblk, blk = i - env * nblk;
  const int T = m->max_episode_steps;
  const int s0 = st.step_idx[env] + ahead;
  const int step = s0 % T, episode = st.episode[env] + s0 / T;
  const int64_t genv = st.env_id_offset + env;
  const uint32_t key[2] = {(uint32_t)st.seed, (uint32_t)(st.seed >> 32)};
  const uint32_t ctr[4] = {(uint32_t)genv, (uint32_t)((uint64_t)genv >> 32), (uint32_t)episode, KM_ACT_CTR3(step, blk)};
  uint32_t o[4];
  philox4x32_10(ctr, key, o);
  float* a = act + (size_t)env * m->act_dim + 4 * blk;
  for (int c = 0; c < 4 && 4 * blk + c < m->act_dim; c++) a[c] = km_action_from_u32(o[c]);
}

void kmanip_launch_sample_action(const KDeviceModel* dm, const KDeviceState& st, float* act, int ahead, hipStream_t stream) {
  const int n = st.num_envs * 4;          // act_dim <= 16: at most 4 blocks per env (idle lanes return)
  hipLaunchKernelGGL(k_sample_action, dim3((n + 255) / 256), dim3(256), 0, stream, dm, st, act, ahead);
}

// ---------------------------------------------------------------------------------------------------------------------
// Wave-slot assignment by predicted cost (round 4).  When a launch has more waves than the chip has SIMD slots (the two-arm
// models at 8192 envs: 4096 waves on 1024 slots), the hardware hands a freed slot the NEXT workgroup in index order, so the
// launch ends at  total / slots + (how late the longest waves started): measured 5.8 M clocks against a mean-bound 4.0 M
// (DualArm @ 8192).  Longest-processing-time-first fixes exactly that: workgroup 0 gets the envs expected to take longest.
// The predictor is what the env's LAST step left in the diagnostics -- contacts persist over steps (a sphere on the cube = the
// coupled 26-dof Newton loop, a sphere on the table = the arm problem off its Woodbury shortcut), the IK's evaluation count
// partly (an "IK failed" start stays failed) -- quantised to 32 bins; envs of a bin also share waves, so a wave's two envs wait
// less for each other.  An env's bits do not depend on its slot or its wave-mates (tests compare shards and launch shapes).
// One workgroup, counting sort, deterministic: thread t owns envs t, t + 256, ...
#define KM_SORT_BINS 32
#define KM_SORT_UNROLL 8
__global__ __launch_bounds__(256) void k_sort_envs(KDeviceState st, int32_t* __restrict__ slot_env, KCostWeights w) {
  __shared__ int cnt[KM_SORT_BINS][256];
  __shared__ int base[KM_SORT_BINS];
  const int t = threadIdx.x, N = st.num_envs;
  // bins of KM_SORT_UNROLL envs at a time: their loads are independent and go out together (one workgroup walks the whole batch:
  // a dependent load per iteration would cost a memory round trip per 256 envs)
  auto bins = [&](int e0, int (&b)[KM_SORT_UNROLL]) {
    uint32_t mask[KM_SORT_UNROLL]; int nf0[KM_SORT_UNROLL], nf1[KM_SORT_UNROLL], wk[KM_SORT_UNROLL];
#pragma unroll
    for (int k = 0; k < KM_SORT_UNROLL; k++) {
      const int e = e0 + 256 * k, ec = e < N ? e : 0;
      mask[k] = st.contact_mask[ec]; nf0[k] = st.ik_nfev[ec]; nf1[k] = st.ik_nfev[(size_t)N + ec]; wk[k] = st.work[ec];
    }
#pragma unroll
    for (int k = 0; k < KM_SORT_UNROLL; k++) {
      const int nf = max(nf0[k], nf1[k]);                              // (the two arms' solves run side by side, one per DPP row)
      // work bit 30: a collider within KM_NEAR_MARGIN of the cube (or on it) -- the coupled Newton loop is on or about to start
      const int cost = w.ik * nf + w.work * (wk[k] & 0x3FFFFFFF) + ((wk[k] >> 30) ? w.coupled : 0)
                       + ((mask[k] & KM_CON_ANY_SPHERE_TABLE) ? w.armtab : 0) + ((mask[k] & KM_CON_ANY_CUBE_TABLE) ? w.cubetab : 0);
      b[k] = e0 + 256 * k < N ? KM_SORT_BINS - 1 - max(0, min(cost / w.binw, KM_SORT_BINS - 1)) : -1;      // bin 0 = heaviest; clamped both ways: cnt[][] / base[] are indexed by it
    }
  };
  for (int b = 0; b < KM_SORT_BINS; b++) cnt[b][t] = 0;
  for (int e0 = t; e0 < N; e0 += 256 * KM_SORT_UNROLL) {
    int b[KM_SORT_UNROLL];
    bins(e0, b);
#pragma unroll
    for (int k = 0; k < KM_SORT_UNROLL; k++) if (b[k] >= 0) cnt[b[k]][t]++;
  }
  __syncthreads();
  // exclusive prefix over the 256 threads of every bin: wave v takes bins v, v + 4, ...; a lane scans four neighbouring threads'
  // counts itself and the 64 lane sums by shuffles
  {
    const int lane = t & 63, wv = t >> 6;
    for (int b = wv; b < KM_SORT_BINS; b += 4) {
      const int c0 = cnt[b][4 * lane], c1 = cnt[b][4 * lane + 1], c2 = cnt[b][4 * lane + 2], c3 = cnt[b][4 * lane + 3];
      const int s = c0 + c1 + c2 + c3;
      int inc = s;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(inc, d, 64); if (lane >= d) inc += v; }
      const int ex = inc - s;
      cnt[b][4 * lane] = ex; cnt[b][4 * lane + 1] = ex + c0; cnt[b][4 * lane + 2] = ex + c0 + c1; cnt[b][4 * lane + 3] = ex + c0 + c1 + c2;
      if (lane == 63) base[b] = inc;
    }
  }
  __syncthreads();
  if (t == 0) { int acc = 0; for (int b = 0; b < KM_SORT_BINS; b++) { const int c = base[b]; base[b] = acc; acc += c; } }
  __syncthreads();
  for (int e0 = t; e0 < N; e0 += 256 * KM_SORT_UNROLL) {
    int b[KM_SORT_UNROLL];
    bins(e0, b);
#pragma unroll
    for (int k = 0; k < KM_SORT_UNROLL; k++) if (b[k] >= 0) slot_env[base[b[k]] + cnt[b[k]][t]++] = e0 + 256 * k;
  }
}
void kmanip_launch_sort_envs(const KDeviceState& st, int32_t* slot_env, const KCostWeights& w, hipStream_t stream) {
  hipLaunchKernelGGL(k_sort_envs, dim3(1), dim3(256), 0, stream, st, slot_env, w);
}
