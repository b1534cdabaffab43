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
