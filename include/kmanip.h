/*
 * kmanip.h -- C ABI of the MI355X-native batched gym-kmanip hot path.
 *
 * The reference (kscalelabs/gym-kmanip) has no FFI: its backend seam is the Python object
 * returned by `env_sim.new(gym_env)` (reference gym_kmanip/env_base.py:192-200), on which
 * KManipEnv only ever calls k_reset() (env_base.py:221), k_step(action) (env_base.py:242),
 * k_render(cam) (env_base.py:217) and k_close() (env_base.py:266).  This header is what a
 * third backend (`env_hip.new`) binds with ctypes; every entry point cites the reference
 * interface it replaces.  INTEGRATION.md shows the reference-side stub.
 *
 * Conventions
 *   - plain C, no torch / HIP types in signatures; device buffers are passed as void* device
 *     pointers, the stream as void* (hipStream_t) -- NULL = the null stream.
 *   - every function returns 0 on success, nonzero on error; kmanip_last_error() gives text.
 *   - numerical blow-up of one env is DATA (bit 1 of its done byte), not an error.
 *   - one handle <-> one device; calls on a handle are serialised by the caller
 *     (the reference is single-threaded and non-re-entrant: ik_mujoco.py:34,67).
 *
 * Arithmetic type: float64 on device, like the reference (MuJoCo mjtNum = double,
 * gym_kmanip/__init__.py:50 OBS_DTYPE = float64).  Actions are float32 (__init__.py:51).
 */
#ifndef KMANIP_H
#define KMANIP_H

#include <stdint.h>

/* The library is built with -fvisibility=hidden: the entry points declared in this header (and the diagnostics of
 * kmanip_debug.h) are the ONLY symbols it exports (tests/test_abi.py compares the dynamic symbol table with the two headers). */
#if defined(__GNUC__)
#define KMANIP_API __attribute__((visibility("default")))
#else
#define KMANIP_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define KM_MAX_LINKS   20   /* 1-DoF robot links == robot dofs == nu == q_len (10 solo, 20 dual/torso) */
#define KM_MAX_ARMS    2    /* arm 0 = right ("eer"), arm 1 = left ("eel")                         */
#define KM_MAX_IK      7    /* IK unknowns per arm (7 solo/dual, 6 torso)                          */
#define KM_MAX_SPHERES 12   /* sphere colliders per model: per arm two fingers, the palm and three joint housings */
/* Contact slots the solver keeps per step: 4 cube corners on the table, KM_SPHERE_SLOTS sphere-cube contacts and
 * KM_SPHERE_TABLE_SLOTS sphere-table contacts -- the first penetrating spheres in sphere-index order (fingers come first); a
 * further penetrating sphere is dropped for that sub-step (its mask bit stays clear), like the 5th+ cube corner.  Two per arm
 * and kind, except three per arm on the table for the two-arm models, whose hands rest on it (measured under random actions:
 * the single arm never has more than two spheres on the table, the Torso up to six). */
#define KM_SPHERE_SLOTS(nlink) (2 * ((nlink) / 10))
#define KM_SPHERE_TABLE_SLOTS(nlink) ((nlink) >= 20 ? 6 : 2)
#define KM_MAX_CAMS    4    /* cameras: 0 = grip_r, 1 = grip_l, 2 = top, 3 = head (__init__.py:157-161) */
enum { KM_CAM_GRIP_R = 0, KM_CAM_GRIP_L = 1, KM_CAM_TOP = 2, KM_CAM_HEAD = 3 };
#define KM_NQ_CUBE     7
#define KM_NV_CUBE     6

/* action-key columns in the flat [num_envs, act_dim] float32 action matrix; key order is the
 * insertion order of the action Dict space, reference env_base.py:151-188 */
enum {
  KM_ACT_EEL_POS = 0, KM_ACT_EEL_ORN, KM_ACT_EER_POS, KM_ACT_EER_ORN,
  KM_ACT_GRIP_L, KM_ACT_GRIP_R, KM_ACT_QPOS_R, KM_ACT_QPOS_L, KM_ACT_NKEYS
};

enum { KM_JNT_HINGE = 0, KM_JNT_SLIDE = 1 };
enum { KM_SOLVER_PGS = 0, KM_SOLVER_NEWTON = 1 };

/* done byte */
#define KM_DONE_TRUNCATED 1u  /* step_idx reached max_episode_steps (TimeLimit, __init__.py:28,247) */
#define KM_DONE_DIVERGED  2u  /* NaN / |qacc| > 1e10 (dm_control raises PhysicsError instead)       */

/* contact-mask bits (uint32 per env), bit-exact parity target */
#define KM_CON_CUBE_TABLE(c)   (1u << (c))          /* c = 0..7 cube corner vs table plane (<=4 kept) */
#define KM_CON_SPHERE_CUBE(s)  (1u << (8 + (s)))    /* s = sphere index 0..11 (fingers first, then link spheres): sphere vs cube */
#define KM_CON_SPHERE_TABLE(s) (1u << (20 + (s)))   /* sphere vs table plane                                         */
#define KM_CON_ANY_CUBE_TABLE   0x000000FFu
#define KM_CON_ANY_SPHERE_CUBE  0x000FFF00u
#define KM_CON_ANY_SPHERE_TABLE 0xFFF00000u
/* the finger spheres come first in the sphere list, two per arm: "cube touches a gripper finger" (env_sim.py:164-175) */
#define KM_CON_FINGERS_CUBE(nlink) (((1u << (2 * ((nlink) / 10))) - 1u) << 8)

typedef struct KModelDesc {
  /* ---- sizes */
  int32_t nlink;                 /* robot 1-DoF links; nq = nlink + 7, nv = nlink + 6, nu = nlink */
  int32_t narm;                  /* arms present in act_list (0..2)                                */
  int32_t nsphere;
  int32_t act_dim;               /* columns of the action matrix                                   */
  int32_t obs_dim;               /* 2*nlink + 7                                                    */
  int32_t max_episode_steps;     /* 64, __init__.py:28                                             */
  int32_t n_sub_steps;           /* control_timestep / timestep = 10, __init__.py:30, env_sim.py:210 */
  int32_t solver_iterations;     /* PGS sweeps cap (MuJoCo default 100)                            */
  int32_t touch_reward_enabled;  /* 0 reproduces the reference's dead touch/lift terms (SURVEY finding 4) */
  int32_t auto_reset;            /* 1: envs whose done byte is set are reset inside kmanip_step    */
  int32_t act_col[KM_ACT_NKEYS]; /* first column of each action key, -1 if the key is absent       */
  int32_t solver;                /* KM_SOLVER_NEWTON (MuJoCo's default, what the reference runs) or KM_SOLVER_PGS */
  /* OPT-IN, NOT THE REFERENCE: cap on the function evaluations of one ik() call.  0 (default) = the reference's
   * scipy.optimize.least_squares default max_nfev = 100 * n (ik_mujoco.py:129-135 passes none): a start in a flat region lets the
   * TRF crawl for 350-650 evaluations, and a batch's launch waits for it (one launch in ~40 at 4096 envs runs 2-7 x long).  A
   * throughput caller may bound it (e.g. 64): the IK then returns its best accepted point with status 0, like SciPy at max_nfev;
   * every parity test and the bench's headline run with 0. */
  int32_t ik_max_nfev;
  int32_t pad0_;

  /* ---- kinematic tree (link i <-> dof i <-> qpos i <-> actuator i), parents before children */
  int32_t link_parent[KM_MAX_LINKS];       /* -1 = fixed to world                                  */
  int32_t jnt_type[KM_MAX_LINKS];
  int32_t forcelimited[KM_MAX_LINKS];
  int32_t pad1_[KM_MAX_LINKS];
  double  link_pos[KM_MAX_LINKS][3];       /* pose in parent link frame (world if parent == -1)    */
  double  link_quat[KM_MAX_LINKS][4];      /* wxyz                                                 */
  double  jnt_axis[KM_MAX_LINKS][3];       /* local                                                */
  double  jnt_range[KM_MAX_LINKS][2];
  double  frictionloss[KM_MAX_LINKS];
  double  kp[KM_MAX_LINKS];
  double  ctrlrange[KM_MAX_LINKS][2];
  double  forcerange[KM_MAX_LINKS][2];
  double  mass[KM_MAX_LINKS];              /* surrogate inertials (build-owned)                    */
  double  com[KM_MAX_LINKS][3];            /* link frame                                           */
  double  inertia[KM_MAX_LINKS][3];        /* diagonal, link-frame axes, about com                 */
  double  q_home[KM_MAX_LINKS];            /* float32-rounded, __init__.py:53-122                  */

  /* ---- arms: arm 0 = right, arm 1 = left */
  int32_t arm_present[KM_MAX_ARMS];
  int32_t arm_nq[KM_MAX_ARMS];
  int32_t arm_q_id[KM_MAX_ARMS][KM_MAX_IK + 1]; /* q_id_r_mask / q_id_l_mask, __init__.py:125-136  */
  int32_t arm_grip_id[KM_MAX_ARMS][2];          /* ctrl_id_*_grip                                  */
  int32_t arm_site_link[KM_MAX_ARMS];
  int32_t arm_mode[KM_MAX_ARMS];                /* 0 = none, 1 = EE-delta + IK, 2 = joint-delta    */
  int32_t arm_has_grip[KM_MAX_ARMS];
  int32_t pad2_[2];
  double  arm_site_pos[KM_MAX_ARMS][3];         /* site pose in its link frame                     */
  double  arm_site_quat[KM_MAX_ARMS][4];

  /* ---- colliders (surrogates): finger + link spheres, table top (height + rectangle), cube box */
  int32_t sphere_link[KM_MAX_SPHERES];
  int32_t sphere_visible[KM_MAX_SPHERES];       /* 1: drawn by the camera renders (fingers); 0: collision only    */
  double  sphere_pos[KM_MAX_SPHERES][3];
  double  sphere_radius[KM_MAX_SPHERES];
  /* Capsule sections: a non-zero sphere_seg[s] makes candidate s, AGAINST THE CUBE, the closest point of the link-fixed segment
   * [sphere_pos, sphere_pos + sphere_seg] to the cube centre (a sphere of the same radius sliding along the link): the forearm /
   * elbow housings are the ends of capsules between consecutive joint origins.  Against the table plane a capsule touches in
   * its end spheres, which are candidates of their own, so that test stays on sphere_pos. */
  double  sphere_seg[KM_MAX_SPHERES][3];
  double  table_z;
  /* The table top is the rectangle x_lo..x_hi, y_lo..y_hi at height table_z (world frame): 0.8 m x 0.4 m centred on the table body
   * -- the plane primitive the reference itself puts in the mesh's place (examples/4_teleop.py:82-84: "table is easier to construct
   * from base vuer plane primitive than load from stl", TABLE_SIZE 0.4 x 0.8 at body "table"; the long side lies along x: the cube
   * spawns at x up to 0.3, __init__.py:164-170).  A cube corner or a sphere touches the table only while its centre line meets the
   * rectangle; beside it, it falls.  +-INFINITY = the round-2 infinite plane. */
  double  table_rect[4];

  /* ---- cube (free body), scene.xml:17-21 */
  double  cube_mass;
  double  cube_inertia[3];
  double  cube_half[3];
  double  cube_frictionloss;
  double  cube_quat0[4];
  double  cube_spawn_lo[3];                     /* __init__.py:164-170                             */
  double  cube_spawn_hi[3];

  /* ---- contact parameters after MuJoCo's pair mixing */
  double  con_cube_solref[2];                   /* pairs involving the cube (condim 4)             */
  double  con_cube_solimp[5];
  double  con_cube_friction[3];                 /* tangential, torsional, rolling                  */
  double  con_def_solref[2];                    /* finger-table pairs (condim 3), joint limits, friction loss */
  double  con_def_solimp[5];
  double  con_def_friction[3];

  /* ---- options / constants */
  double  timestep;                             /* 0.002 (MuJoCo default, no <option> in any XML)  */
  double  gravity[3];
  double  solver_tolerance;                     /* 1e-8                                            */
  double  ik_res_rad, ik_res_reg_prev, ik_res_reg_home, ik_jac_rad, ik_jac_reg; /* __init__.py:37-41 */
  double  ee_pos_delta[3], ee_orn_delta[3];     /* __init__.py:174-187                             */
  double  q_pos_delta;                          /* __init__.py:196                                 */
  double  ee_s_min, ee_s_max, ee_s_delta;       /* __init__.py:199-201                             */
  double  max_q_vel;                            /* pi, __init__.py:31                              */
  double  epsilon;                              /* 1e-6, __init__.py:192                           */
  double  reward_vel_penalty, reward_grip_dist, reward_touch_cube, reward_lift_cube; /* :205-208  */

  /* ---- cameras, all mode="targetbody": gripper cameras on the hand links tracking the EE site body
   * (arm_r_body.xml:68 / arm_l_body.xml:68 / torso_body.xml:104,173) and the world-fixed `top` / `head` cameras
   * tracking the table (_env_solo_arm.xml:11-12 and siblings).  Index = KM_CAM_*.  A link of -1 = world frame.
   * The rendered scene is the build's surrogate geometry = its collision primitives. */
  int32_t cam_present[KM_MAX_CAMS];
  int32_t cam_link[KM_MAX_CAMS];
  int32_t cam_target_link[KM_MAX_CAMS];
  double  cam_pos[KM_MAX_CAMS][3];          /* in cam_link's frame                                    */
  double  cam_target_pos[KM_MAX_CAMS][3];   /* target body origin in cam_target_link's frame          */
  double  cam_fovy[KM_MAX_CAMS];            /* degrees                                                */
  double  cam_znear, cam_zfar;              /* metres; depth is clipped to [znear, zfar], no hit = zfar */

  /* ---- quantities MuJoCo precomputes at qpos0 (mj_setConst): they set the regulariser of every constraint row
   * (efc_diagApprox -> R = (1-d)/d * diagApprox) and the solvers' termination scale 1/(meaninertia * nv).
   * dof_invweight0[i] = (M^-1)_ii; body_invweight0[b] = mean translational / rotational diagonal of J_b M^-1 J_b^T
   * with J_b the 6 x nv Jacobian at body b's centre of mass; the cube (free body): (1/m, mean 1/I_k). */
  double  dof_invweight0[KM_MAX_LINKS];
  double  body_invweight0[KM_MAX_LINKS][2];
  double  cube_invweight0[2];
  double  meaninertia;                      /* trace(M(qpos0)) / nv, all nv = nlink + 6 dofs          */
} KModelDesc;

typedef struct KHandle_* KHandle;

/* sizeof(KModelDesc) as compiled into the library (host bindings check their mirror). */
KMANIP_API int kmanip_model_desc_size(void);

/* Replaces env_sim.new(gym_env) (reference env_sim.py:206-211): build `num_envs` simulated envs
 * on HIP device `device`.  `env_id_offset` is the global index of local env 0 (multi-GPU
 * sharding: RNG streams are keyed by the GLOBAL env id so results do not depend on the shard
 * layout).  The library owns model, state and scratch. */
KMANIP_API int kmanip_create(const KModelDesc* desc, int num_envs, int device, uint64_t seed,
                  int64_t env_id_offset, KHandle* out);
/* Launch shape (no effect on results: an env's bits depend neither on its wave-mates nor on the order its wave is dispatched in --
 * tests compare shards, launch shapes and orders bit for bit).  A step is ONE launch of single-wave workgroups holding 4 (10-link
 * models) or 2 (20-link models) envs each.  When a two-arm handle has more waves than the GPU has SIMD slots (more than 2048 envs),
 * kmanip_step first orders the envs by the cost their last step predicts and dispatches the longest waves first (k_sort_envs, one
 * small extra launch).  Diagnostic environment variables, read at create: KMANIP_COST_SORT=0/1 (force that order
 * off / on), KMANIP_COST_W (its weights), KMANIP_EPB (envs per wave), KMANIP_IK_UNFUSED=1 (before_step as its own launch),
 * KMANIP_NO_BLOCK_SPLIT=1 (two-arm inertia as one block), KMANIP_WAVE_CLOCKS=1 (per-wave cycle counts for tests/tools/wave_times.py).
 * Throughput note: one batch's launch ends with its slowest wave and leaves about half the SIMD time idle; handles are
 * independent and every entry point takes the caller's stream, so two or more batches kept in flight on different streams fill
 * it (gym_kmanip_amd/pipeline.py). */

/* Replaces KManipEnvSim.k_reset (env_sim.py:190-194) -> KManipTask.initialize_episode
 * (env_sim.py:23-36): reset envs whose mask byte is nonzero (NULL = all) to the home pose,
 * zero velocity and a fresh cube spawn; writes their observation rows.  mask/obs are device
 * pointers: mask uint8[num_envs], obs double[num_envs, obs_dim] (may be NULL). */
KMANIP_API int kmanip_reset(KHandle h, const uint8_t* mask_dev, double* obs_dev, void* stream);

/* Replaces KManipEnvSim.k_step (env_sim.py:196-200): one control step for every env =
 * KManipTask.before_step (env_sim.py:38-108, incl. ik_mujoco.ik) + physics.step(10)
 * + get_reward (env_sim.py:148-179) + get_observation (env_sim.py:110-146).
 * act_dev float[num_envs, act_dim]; obs_dev double[num_envs, obs_dim];
 * reward_dev double[num_envs]; done_dev uint8[num_envs] (KM_DONE_* bits).  All device memory,
 * owned by the caller. */
KMANIP_API int kmanip_step(KHandle h, const float* act_dev, double* obs_dev, double* reward_dev,
                uint8_t* done_dev, void* stream);

/* State access for parity tests / checkpointing (state is
 * (qpos, qvel, ctrl, qacc_warmstart, time)); HOST pointers, env-major
 * [num_envs, nq|nv|nu|nv], step_idx int32[num_envs]; any pointer may be NULL. Synchronous. */
KMANIP_API int kmanip_get_state(KHandle h, double* qpos, double* qvel, double* ctrl, double* qacc_warm,
                     int32_t* step_idx);
KMANIP_API int kmanip_set_state(KHandle h, const double* qpos, const double* qvel, const double* ctrl,
                     const double* qacc_warm, const int32_t* step_idx);

/* The episode counter completes the checkpoint: the cube-spawn stream is keyed (seed, global env id, episode), so a
 * restored run draws the same spawns as the original only if `episode` is restored too.  HOST int32[num_envs]. Synchronous. */
KMANIP_API int kmanip_get_episode(KHandle h, int32_t* episode);
KMANIP_API int kmanip_set_episode(KHandle h, const int32_t* episode);

/* Asynchronous device-to-device copy of the per-env counters into caller-owned DEVICE buffers (int32[num_envs] each, either
 * may be NULL) on `stream`: lets the k_step seam return sim_time (= step_idx * control_timestep, env_sim.py:194,200) as a
 * device tensor without synchronising. */
KMANIP_API int kmanip_get_counters(KHandle h, int32_t* step_idx_dev, int32_t* episode_dev, void* stream);

/* Bind a caller-owned DEVICE buffer double[num_envs] that every kmanip_step / kmanip_reset also fills with the env's
 * simulation time (dm_control's data.time, the last element of k_step's return tuple, env_sim.py:194,200) =
 * steps since the env's last reset x control_timestep.  NULL unbinds.  The buffer must outlive the binding. */
KMANIP_API int kmanip_bind_sim_time(KHandle h, double* sim_time_dev);

/* The multi-GPU learner's per-step exchange is one packed record per env, (reward, done as a double), all-gathered across the
 * ranks (gym_kmanip_amd/dist.py).  Bound here, every kmanip_step writes that record itself into ONE of two
 * caller-owned double[num_envs, 2] buffers -- the one chosen by kmanip_select_reward_done_record (rec0 after the bind) -- so that
 * the exchange costs the step's stream no packing kernel.  reward_dev / done_dev are written as always.  NULL, NULL unbinds;
 * kmanip_step_chunk does not write records.
 * ORDERING INVARIANT (the caller's): a collective that still READS buffer b must have completed -- or the step's stream must have
 * been made to wait for it -- BEFORE the kmanip_step that fills b is enqueued; two buffers only mean that step k may overlap
 * the exchange of step k-1, not that no wait is needed (dist.RewardDoneGather.before_step).  The library keeps no counter of
 * its own: a step without an exchange (evaluation, a failed launch) cannot put the two sides out of phase. */
KMANIP_API int kmanip_bind_reward_done_record(KHandle h, double* rec0_dev, double* rec1_dev);
/* index 0 / 1: the bound buffer the following kmanip_step calls fill. */
KMANIP_API int kmanip_select_reward_done_record(KHandle h, int index);

/* KManipTask.get_observation + get_reward (env_sim.py:110-179) of every env's CURRENT state, without stepping -- what
 * dm_control evaluates after a physics.forward(): obs_dev double[num_envs, obs_dim], reward_dev double[num_envs] (either may be
 * NULL); the contact masks kmanip_get_diag returns are refreshed too.  An env whose qpos / qvel hold a non-finite value (a restored
 * diverged checkpoint) gets what kmanip_step reports for a diverged env: zero observation, zero reward, an empty contact mask.  Used by the parity tests against fixtures made from the
 * reference's own Python (tests/golden/ref_obs_*.npz) and by callers that restore a checkpoint with kmanip_set_state. */
KMANIP_API int kmanip_observe(KHandle h, double* obs_dev, double* reward_dev, void* stream);

/* KManipEnv.reset(seed=...) (env_base.py:219-220): re-key the cube-spawn stream.  restart_episodes != 0 also rewinds every
 * env's episode counter so that the next kmanip_reset draws episode 0 of the new seed (reset(seed=s) is then reproducible). */
KMANIP_API int kmanip_set_seed(KHandle h, uint64_t seed, int restart_episodes);

/* Per-env diagnostics of the last kmanip_step (HOST pointers, may be NULL):
 * contact_mask uint32 (KM_CON_* bits, from the trailing mj_step1), ik_nfev int32[num_envs, 2],
 * ik_status int32[num_envs, 2]. Synchronous. */
KMANIP_API int kmanip_get_diag(KHandle h, uint32_t* contact_mask, int32_t* ik_nfev, int32_t* ik_status);

/* Kernel timing with HIP events recorded on the launch stream (bench.py roofline leg).  While enabled, every kmanip_step
 * records an event before and after k_step, one more after the bound in-step render (or after the first kmanip_render_rgb[_multi]
 * call that follows the step: the camera observations of a *Vision id), and -- only on the KMANIP_IK_UNFUSED=1 A/B
 * path -- one before its stand-alone decode/IK launches (an event record costs the stream about 5 us, so the product path takes
 * the two it needs), into a ring of `KM_TIMING_SLOTS` steps.  kmanip_timing_summary synchronises the device and returns the
 * summed durations in milliseconds of the three legs (stand-alone IK: 0 on the product path; k_step; the step's render:
 * kmanip_bind_step_depth's or the RGB render called after the step, 0 without one) over the recorded steps, then clears the ring.  Any output pointer may be NULL.
 * `enable` = k > 1 records every k-th step only (the first step after the call, then every k-th): the events' own cost -- the ~5 us
 * above, 1 % of a 4096-env step -- then falls on one step in k, and the averages are over the sampled steps (`*nsteps` = their count). */
#define KM_TIMING_SLOTS 1024
KMANIP_API int kmanip_enable_timing(KHandle h, int enable);
KMANIP_API int kmanip_timing_summary(KHandle h, double* ik_ms_sum, double* dyn_ms_sum, double* render_ms_sum, int32_t* nsteps);

/* nsteps control steps in ONE launch, for callers that already hold the next nsteps actions of every env (action-chunking
 * policies such as ACT, scripted / replayed action streams): exactly the result of nsteps consecutive kmanip_step calls,
 * with act_dev float[nsteps, num_envs, act_dim], obs_dev double[nsteps, num_envs, obs_dim], reward_dev double[nsteps,
 * num_envs], done_dev uint8[nsteps, num_envs].  Without a launch boundary per step the waves do not wait for the
 * batch's slowest env at every step, so throughput follows the mean wave rather than the slowest one. */
KMANIP_API int kmanip_step_chunk(KHandle h, int nsteps, const float* act_dev, double* obs_dev, double* reward_dev,
                      uint8_t* done_dev, void* stream);

/* Standalone batched IK (ik_mujoco.ik, reference ik_mujoco.py:100-155) for parity tests:
 * qpos HOST double[n, nq] (in: current; out: qpos after the IK's last evaluation),
 * goal_pos double[n,3], goal_quat double[n,4] (wxyz), arm 0/1; q_out double[n, arm_nq]
 * (the clipped result.x that the reference writes into ctrl). Synchronous. */
KMANIP_API int kmanip_ik(KHandle h, int arm, int n, double* qpos, const double* goal_pos,
              const double* goal_quat, double* q_out, int32_t* nfev, int32_t* status);

/* ik_res / ik_jac (reference ik_mujoco.py:20-53 / :56-97) as the device IK evaluates them, at x = qpos[q_mask] with
 * q_pos_prev = x, for parity tests: qpos HOST double[n, nq], goal_pos [n,3], goal_quat [n,4] (wxyz);
 * res double[n, 6 + 2*arm_nq], jac double[n, (6 + 2*arm_nq) x arm_nq] row-major.  Synchronous. */
KMANIP_API int kmanip_ik_eval(KHandle h, int arm, int n, const double* qpos, const double* goal_pos,
                   const double* goal_quat, double* res, double* jac);

/* Replaces KManipEnvSim.k_render (env_sim.py:187-188) / the camera branch of get_observation
 * (env_sim.py:140-145) for the gripper cameras, as BASELINE.json config 5 defines it: a height x width
 * float32 DEPTH image (metres along the optical axis) of every env's current state.
 * depth_dev: float[num_envs, height, width] device memory owned by the caller. */
KMANIP_API int kmanip_render_depth(KHandle h, int cam, int height, int width, float* depth_dev, void* stream);

/* The same cameras as uint8 RGB, what dm_control's physics.render(height, width, camera_id) returns for the camera
 * observations of the *Vision env ids (env_sim.py:140-145; shapes env_base.py:140-146, cameras __init__.py:157-161) and for
 * KManipEnv.render() (env_base.py:215-217, the `top` camera): rgb_dev uint8[num_envs, height, width, 3], caller-owned
 * device memory.  Lambert shading of the surrogate scene under the reference's lights (scene.xml:8-13). */
KMANIP_API int kmanip_render_rgb(KHandle h, int cam, int height, int width, uint8_t* rgb_dev, void* stream);
/* ncam (<= KM_MAX_CAMS) of those images in ONE launch: the whole camera branch of a *Vision observation (head 480 x 640 + grip
 * 40 x 60 per arm: env_base.py:140-146) costs the step's stream one launch instead of one per camera.  cams / heights / widths /
 * rgb_dev are HOST arrays of ncam entries; rgb_dev[i] is device memory uint8[num_envs, heights[i], widths[i], 3]. */
KMANIP_API int kmanip_render_rgb_multi(KHandle h, int ncam, const int* cams, const int* heights, const int* widths, uint8_t* const* rgb_dev,
                            void* stream);

/* Rendering BEHIND the steps (a data-generation loop whose policy does not look at the images: the reference's scripted heuristic,
 * examples/2_synthetic_data.py:28-41, logs them and acts on the state).  A render reads nothing of the state but qpos:
 * kmanip_snapshot_render_state copies qpos into snapshot `slot` (0 or 1) on `stream` -- the step's stream, after the step whose
 * images are wanted -- and kmanip_set_render_source(h, slot) makes the kmanip_render_* calls that follow read that copy
 * (-1: the live state again, the default; host-side switch, not stream-ordered).  The caller can then issue the render on a
 * SECOND stream while the next kmanip_step runs on the first: the render's workgroups take the SIMDs the step's early-finishing
 * waves free (one 2048-env step + head and grip images: 0.96 ms in sequence, 0.67 ms this way; gym_kmanip_amd/pipeline.py
 * RenderBehind).  Ordering is the caller's: the render stream waits for the copy (an event), and a slot is not overwritten before
 * the render that reads it has finished.  kmanip_bind_step_depth's in-step render always reads the live state. */
KMANIP_API int kmanip_snapshot_render_state(KHandle h, int slot, void* stream);
KMANIP_API int kmanip_set_render_source(KHandle h, int slot);

/* BASELINE config 5 ("64x64 gripper-cam depth render in the step"): bind a caller-owned device buffer
 * float[num_envs, height, width]; every kmanip_step then ends by rendering camera `cam` of the state it produced into it,
 * on the step's stream (one C call per control step).  depth_dev == NULL unbinds. */
KMANIP_API int kmanip_bind_step_depth(KHandle h, int cam, int height, int width, float* depth_dev);

/* The scripted data-generation policy of reference examples/2_synthetic_data.py:28-41, for every env, on device:
 * act_dev float[num_envs, act_dim] arrives holding action_space.sample() (the caller draws it) and leaves with its
 * eer_pos columns overwritten by the unit vector from the right end-effector site to the cube centre
 * (cube_pos - site("eer_site_pos").xpos, normalised), evaluated at the env's current state.  The reference
 * stores that vector as float64 in the action dict; the flat action buffer is float32 (ACT_DTYPE).
 * Returns an error for env ids without an eer_pos action (the *QPos ids). */
KMANIP_API int kmanip_scripted_action(KHandle h, float* act_dev, void* stream);

/* action_space.sample() for every env, on device -- what the reference's rollout loops feed env.step with
 * (examples/2_log_with_h5py.py:22-26, 3_save_to_video.py:20-27; spaces env_base.py:151-188: every key a Box(-1, 1, float32)):
 * act_dev float[num_envs, act_dim] is filled with U[-1, 1) float32 from a counter-based stream, Philox4x32-10 keyed by the
 * handle's seed with counter (global env id, episode, step): the action of an env at a given (episode, step) does not depend
 * on the shard layout or on what was drawn before, and the CPU oracle draws identical bits (SURVEY 8d's synthetic inputs).
 * `ahead` >= 0 draws the action the env will need `ahead` control steps from now, assuming TimeLimit-only episodes (the
 * reference never terminates early), so the next K actions can be laid out before stepping. */
KMANIP_API int kmanip_sample_action(KHandle h, float* act_dev, int ahead, void* stream);

KMANIP_API int kmanip_num_envs(KHandle h);
KMANIP_API const char* kmanip_last_error(KHandle h);   /* h may be NULL: error of the last failed create */
KMANIP_API const char* kmanip_version(void);

/* Replaces KManipEnvSim.k_close (env_sim.py:202-203). */
KMANIP_API void kmanip_destroy(KHandle h);

#ifdef __cplusplus
}
#endif
#endif /* KMANIP_H */
