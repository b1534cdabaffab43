// kmanip_render.hip -- camera renders of every env's current state: float32 depth (BASELINE.json config 5) and uint8 RGB
// (the camera observations of the *Vision env ids and KManipEnv.render()).
//
// Replaces the camera branch of KManipTask.get_observation / KManipEnvSim.k_render (reference
// gym_kmanip/env_sim.py:140-145,187-188 -> dm_control physics.render(height, width, camera_id)).  Cameras are the
// reference's four, all mode="targetbody": grip_r / grip_l on the hand links tracking the EE site body (fovy 20:
// arm_r_body.xml:68, arm_l_body.xml:68, torso_body.xml:104,173) and the world-fixed top / head tracking the table
// (fovy 78: _env_solo_arm.xml:14-15 and siblings).  The rendered scene is the build's surrogate geometry = its collision
// primitives (cube box, table plane, finger spheres): the reference's robot meshes are absent from the checkout
// (DESIGN.md section 5), so pixel parity with MuJoCo's OpenGL renderer is not defined; parity is against the oracle's
// restatement of the same ray caster.
//   depth : metres along the optical axis, clipped to [znear, zfar], no hit = zfar
//   rgb   : Lambert shading of the reference's material colours (cube rgba 1 0 0, table rgba .2 .2 .2: scene.xml:15,20)
//           under the reference's lights (scene.xml:8-13: headlight ambient 0.4 + MuJoCo's default headlight diffuse 0.4,
//           three directional lights of diffuse 0.3), no specular / shadows / fog; background black
//
// One workgroup (256 lanes) per env.  Forward kinematics run one link per lane with ceil(log2(depth)) rounds of pointer
// jumping through LDS (the same scheme as k_step's fk_parallel), lane 0 builds the camera frame, then the lanes ray-cast.
//   depth (k_render_depth): 128 lanes per env, pixels p = lane, lane + 128, ...; float64 ray maths -- the depth parity bar
//     against the float64 oracle is 1e-6 m, which float32 intersection arithmetic (cancellation ~1e-5 m) would not meet.
//   rgb (k_render_rgb, round 3): the *Vision observations are 480 x 640 x 3 + 40 x 60 x 3 bytes per env-step (__init__.py:
//     158-161), 929 KB -- the one genuinely HBM-sized output of the path.  Round 2 cast every pixel in float64 (~150 FP64
//     operations per pixel: 5 x the time the stores need) and wrote three separate bytes per pixel.  Now: (1) float32 ray
//     maths (the parity bar is one grey level); (2) the cube and the visible spheres are projected to screen-space bounding
//     rectangles once per env, and a pixel outside all of them can only see the table plane or the background: ~15
//     operations (most of a head / top image); (3) a lane shades FOUR consecutive pixels of a row and stores their 12 bytes as
//     three dwords (768 contiguous bytes per wave-instruction group); (4) the per-ray divides of the slab and plane tests are
//     v_rcp_f32 on quantities that are linear in the pixel coordinates.
#include "kmanip_device.hpp"
#include <stdlib.h>

struct RenderScene {
  real xpos[KM_MAX_LINKS][3], xmat[KM_MAX_LINKS][9];
  real cam_o[3], cam_x[3], cam_y[3], cam_z[3];
  real cube_p[3], cube_R[9];
  real sph[KM_MAX_SPHERES][3];
  real focal;
  real cube_q[8];              // the cube's pose (qpos[nl .. nl + 6]) as read by lanes 0-6 at the top of the set-up
};
// what the camera / sphere lanes need from global memory, read at the top of render_fk together with the kinematics' inputs
struct RenderPre { int sl, cl, tl; real sp[3], cp[3], tp[3], tanhalf; };

// mj_kinematics, one link per lane + pointer jumping (block-wide barriers: the workgroup is 4 waves).
// Round 6: every global read of the set-up is issued at the top and waited for once -- the joint angle (HBM), the link's constants
// and its jump table for all four rounds (L2): read where they were used, each round's `jump[k][i]` cost another L2 round trip
// behind a barrier, and a launch whose 2048 workgroups all start together hides none of it (the 64 x 64 depth render: 8 of 39 us).
__device__ __forceinline__ void render_fk(const KDeviceModel* dm, const KDeviceState& st, int env, int cam, RenderScene* sc, RenderPre& pre) {
  const KModelDesc* m = &dm->d;
  const int nl = m->nlink, NE = st.num_envs, i = threadIdx.x;
  const bool on = i < nl;
  const int ii = on ? i : 0;
  real R[9], p[3], Rl[9];
  int ja[4];
  // lanes 0-6: one component of the cube's pose each; lanes 64..: one collision sphere each; the camera's constants (wave-uniform)
  real cube_c = st.qpos[(size_t)(nl + (i < 7 ? i : 0)) * NE + env];
  {
    const int s = (i >= 64 && i < 64 + m->nsphere) ? i - 64 : 0;
    pre.sl = m->sphere_link[s];
    pre.sp[0] = m->sphere_pos[s][0]; pre.sp[1] = m->sphere_pos[s][1]; pre.sp[2] = m->sphere_pos[s][2];
    pre.cl = m->cam_link[cam]; pre.tl = m->cam_target_link[cam];
    for (int c = 0; c < 3; c++) { pre.cp[c] = m->cam_pos[cam][c]; pre.tp[c] = m->cam_target_pos[cam][c]; }
    pre.tanhalf = dm->x.cam_tanhalf[cam];
  }
  {
    const double* Rg = dm->x.link_R[ii];                // (normalised link_quat as a matrix: built once per model on the host)
#pragma unroll
    for (int c = 0; c < 9; c++) Rl[c] = Rg[c];
  }
  p[0] = m->link_pos[ii][0]; p[1] = m->link_pos[ii][1]; p[2] = m->link_pos[ii][2];
  real q = st.qpos[(size_t)ii * NE + env];
  int jt = m->jnt_type[ii];
#pragma unroll
  for (int k = 0; k < 4; k++) ja[k] = dm->x.jump[k][ii];
  const int rounds = dm->x.fk_rounds;
  km_pin(Rl); km_pin(p); km_pin(q, cube_c); km_pin_i(jt); km_pin_i(ja[0], ja[1]); km_pin_i(ja[2], ja[3]); km_pin(pre.sp); km_pin_i(pre.sl);
  if (i < 7) sc->cube_q[i] = cube_c;
  if (on) {
    if (jt == KM_JNT_SLIDE) {
#pragma unroll
      for (int c = 0; c < 9; c++) R[c] = Rl[c];
      p[0] += Rl[2] * q; p[1] += Rl[5] * q; p[2] += Rl[8] * q;
    } else {
      real sn, cs;
      km_sincos(q, &sn, &cs);
#pragma unroll
      for (int a = 0; a < 3; a++) {
        R[3 * a] = cs * Rl[3 * a] + sn * Rl[3 * a + 1];
        R[3 * a + 1] = cs * Rl[3 * a + 1] - sn * Rl[3 * a];
        R[3 * a + 2] = Rl[3 * a + 2];
      }
    }
#pragma unroll
    for (int c = 0; c < 9; c++) sc->xmat[i][c] = R[c];
    sc->xpos[i][0] = p[0]; sc->xpos[i][1] = p[1]; sc->xpos[i][2] = p[2];
  }
  __syncthreads();
  for (int k = 0; k < rounds; k++) {
    const int a = on ? (k == 0 ? ja[0] : (k == 1 ? ja[1] : (k == 2 ? ja[2] : ja[3]))) : -1;
    if (a >= 0) {
      real A[9], pa[3], Rn[9], t[3];
#pragma unroll
      for (int c = 0; c < 9; c++) A[c] = sc->xmat[a][c];
      pa[0] = sc->xpos[a][0]; pa[1] = sc->xpos[a][1]; pa[2] = sc->xpos[a][2];
      mat_vec3(t, A, p);
      p[0] = t[0] + pa[0]; p[1] = t[1] + pa[1]; p[2] = t[2] + pa[2];
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) Rn[3 * r + c] = A[3 * r] * R[c] + A[3 * r + 1] * R[3 + c] + A[3 * r + 2] * R[6 + c];
#pragma unroll
      for (int c = 0; c < 9; c++) R[c] = Rn[c];
    }
    __syncthreads();
    if (a >= 0) {
#pragma unroll
      for (int c = 0; c < 9; c++) sc->xmat[i][c] = R[c];
      sc->xpos[i][0] = p[0]; sc->xpos[i][1] = p[1]; sc->xpos[i][2] = p[2];
    }
    __syncthreads();
  }
}

// Per-env scene after the FK: lane 0 builds the camera frame (mj_camlight, targetbody) and the cube pose, lanes 64.. one sphere
// centre each (another wave: in parallel with lane 0).  Caller synchronises afterwards.
__device__ __forceinline__ void render_camera(const KDeviceModel* dm, const KDeviceState& st, int env, int cam, int height, RenderScene* sc, const RenderPre& pre) {
  const KModelDesc* m = &dm->d;
  const int t = threadIdx.x;
  if (t == 0) {
    // camera frame: z = (cam - target)/|.|, x = (0,0,1) x z, y = z x x; a link of -1 = world frame
    const int cl = pre.cl, tl = pre.tl;
    real co[3], to[3], v[3];
    real cp[3] = {pre.cp[0], pre.cp[1], pre.cp[2]};
    real tp[3] = {pre.tp[0], pre.tp[1], pre.tp[2]};
    if (cl < 0) { co[0] = cp[0]; co[1] = cp[1]; co[2] = cp[2]; }
    else { mat_vec3(v, sc->xmat[cl], cp); co[0] = sc->xpos[cl][0] + v[0]; co[1] = sc->xpos[cl][1] + v[1]; co[2] = sc->xpos[cl][2] + v[2]; }
    if (tl < 0) { to[0] = tp[0]; to[1] = tp[1]; to[2] = tp[2]; }
    else { mat_vec3(v, sc->xmat[tl], tp); to[0] = sc->xpos[tl][0] + v[0]; to[1] = sc->xpos[tl][1] + v[1]; to[2] = sc->xpos[tl][2] + v[2]; }
    real z[3] = {co[0] - to[0], co[1] - to[1], co[2] - to[2]}, up[3] = {0, 0, 1}, x[3], y[3];
    normalize3_fast(z);
    cross3(x, up, z); normalize3_fast(x);
    cross3(y, z, x); normalize3_fast(y);
    for (int c = 0; c < 3; c++) { sc->cam_o[c] = co[c]; sc->cam_x[c] = x[c]; sc->cam_y[c] = y[c]; sc->cam_z[c] = z[c]; }
    sc->focal = (0.5 * height) / pre.tanhalf;
    real cq[4];
    for (int c = 0; c < 3; c++) sc->cube_p[c] = sc->cube_q[c];
    for (int c = 0; c < 4; c++) cq[c] = sc->cube_q[3 + c];
    normalize4_fast(cq);
    quat2mat(sc->cube_R, cq);
  } else if (t >= 64 && t < 64 + m->nsphere) {
    const int s = t - 64, l = pre.sl;
    real sl[3] = {pre.sp[0], pre.sp[1], pre.sp[2]}, v[3];
    mat_vec3(v, sc->xmat[l], sl);
    sc->sph[s][0] = sc->xpos[l][0] + v[0]; sc->sph[s][1] = sc->xpos[l][1] + v[1]; sc->sph[s][2] = sc->xpos[l][2] + v[2];
  }
}

// ---- RGB -----------------------------------------------------------------------------------------------------------
#define KM_RGB_MAXSPH KM_RENDER_MAXVIS        // visible spheres (the two finger tips per arm)
struct RgbScene {              // float32 view of the scene for the pixel loop, built by lane 0 from the float64 RenderScene
  float o[3], X[3], Y[3], Z[3], inv_f, tz, zfar;
  float ol[3], DX[3], DY[3], DZ[3], half[3], R[9];       // cube: camera origin and the ray basis in the cube frame, half sizes, rotation
  float oc[KM_RGB_MAXSPH][3], cc[KM_RGB_MAXSPH], ir[KM_RGB_MAXSPH];   // spheres: origin - centre, |oc|^2 - r^2, 1 / r
  int nsph;
  // the table top seen from the camera: a ray direction d = X dx + Y dy - Z passes through the rectangle iff the four edge functions
  // te_a[i] dx + te_b[i] dy + te_c[i] (triple products of d with consecutive corners as seen from the camera, oriented so that the
  // rectangle's centre is positive) are all positive; te_i[i] = -1 / te_a[i] turns a row's values into its column span
  float te_a[4], te_b[4], te_c[4], te_i[4];
  int ubox[4];                     // union of the rectangles below
  int box[1 + KM_RGB_MAXSPH][4];   // screen-space bounding rectangle of the cube [0] and of every visible sphere: r0, r1, c0, c1 (inclusive)
  float tab_L;                 // directional-light sum on the table's normal
};
// pixel (row, col) of the world point P; false if it is not safely in front of the camera
__device__ __forceinline__ bool rgb_project(const RenderScene& sc, const real* P, int height, int width, real& row, real& col) {
  const real pc[3] = {P[0] - sc.cam_o[0], P[1] - sc.cam_o[1], P[2] - sc.cam_o[2]};
  const real zc = -dot3(pc, sc.cam_z);
  if (!(zc > 1e-3)) return false;
  const real s = sc.focal / zc;
  col = dot3(pc, sc.cam_x) * s + 0.5 * width - 0.5;
  row = -dot3(pc, sc.cam_y) * s + 0.5 * height - 0.5;
  return true;
}
// The per-env set-up of the pixel loop, spread over the first 17 lanes of the workgroup (one lane did all of it in ~22 us, a tenth of
// a 2048-image launch at two residency rounds): lanes 0-7 project one cube corner each, lanes 8-11 one visible sphere each (its ray
// constants and rectangle), lanes 12-15 one table edge each, lane 16 the camera / cube-frame scalars; lane 0 then folds the corners
// into the cube's rectangle and the rectangles into their union.
struct RgbTmp { real row[8], col[8]; int ok[8]; };
__device__ __forceinline__ void rgb_scene(const KDeviceModel* dm, const RenderScene& sc, int height, int width, RgbScene* g, RgbTmp* tmp, int t) {
  const KModelDesc* m = &dm->d;
  // bounding rectangles, one per object (an object that is not safely in front of the camera gets the whole image): the cube's
  // eight corners; a sphere's centre +- a conservative projected radius
  auto put = [&](int o, bool ok, real r0, real r1, real c0, real c1) {
    if (!ok) { g->box[o][0] = 0; g->box[o][1] = height - 1; g->box[o][2] = 0; g->box[o][3] = width - 1; return; }
    g->box[o][0] = (int)fmax(floor(r0) - 1, -1.0); g->box[o][1] = (int)fmin(ceil(r1) + 1, (real)height);
    g->box[o][2] = (int)fmax(floor(c0) - 1, -1.0); g->box[o][3] = (int)fmin(ceil(c1) + 1, (real)width);
  };
  if (t < 8) {
    const real loc[3] = {(t & 1 ? 1 : -1) * m->cube_half[0], (t & 2 ? 1 : -1) * m->cube_half[1], (t & 4 ? 1 : -1) * m->cube_half[2]};
    real P[3], row = 0, col = 0;
    mat_vec3(P, sc.cube_R, loc);
    P[0] += sc.cube_p[0]; P[1] += sc.cube_p[1]; P[2] += sc.cube_p[2];
    tmp->ok[t] = rgb_project(sc, P, height, width, row, col);
    tmp->row[t] = row; tmp->col[t] = col;
  } else if (t < 8 + KM_RGB_MAXSPH) {
    // the (t - 8)-th visible sphere (the list is built on the host; kmanip_create refuses models with more than four)
    const int s = t - 8 < dm->x.nvis ? dm->x.vis_sphere[t - 8] : -1;
    if (s >= 0) {
      const int ns = t - 8;
      const real rad = m->sphere_radius[s];
      const real oc[3] = {sc.cam_o[0] - sc.sph[s][0], sc.cam_o[1] - sc.sph[s][1], sc.cam_o[2] - sc.sph[s][2]};
      for (int c = 0; c < 3; c++) g->oc[ns][c] = (float)oc[c];
      g->cc[ns] = (float)(dot3(oc, oc) - rad * rad); g->ir[ns] = (float)(1.0 / rad);
      real row = 0, col = 0;
      const real zc = dot3(oc, sc.cam_z);                                  // depth of the centre along the optical axis
      const bool ok = rgb_project(sc, sc.sph[s], height, width, row, col) && zc - rad > 1e-3;
      const real pr = ok ? 1.5 * sc.focal * rad / (zc - rad) + 1.0 : 0.0;  // (off-axis spheres project to ellipses: generous)
      put(1 + ns, ok, row - pr, row + pr, col - pr, col + pr);
    }
  } else if (t < 16) {
    const int i = t - 12;
    const real* tr = m->table_rect;
    if (isfinite(tr[0]) && isfinite(tr[1]) && isfinite(tr[2]) && isfinite(tr[3])) {
      // corner i and its successor, counter-clockwise from (x_lo, y_lo)
      const real ax = (i == 0 || i == 3) ? tr[0] : tr[1], ay = i < 2 ? tr[2] : tr[3];
      const real bx = (i == 3 || i == 2) ? tr[0] : tr[1], by = (i == 0 || i == 3) ? tr[2] : tr[3];
      const real dz = m->table_z - sc.cam_o[2];
      const real Va[3] = {ax - sc.cam_o[0], ay - sc.cam_o[1], dz}, Vb[3] = {bx - sc.cam_o[0], by - sc.cam_o[1], dz};
      const real Vc[3] = {0.5 * (tr[0] + tr[1]) - sc.cam_o[0], 0.5 * (tr[2] + tr[3]) - sc.cam_o[1], dz};
      real n[3];
      cross3(n, Va, Vb);
      const real sgn = dot3(n, Vc) < 0 ? -1.0 : 1.0;
      const real a = sgn * dot3(n, sc.cam_x), b = sgn * dot3(n, sc.cam_y), c = -sgn * dot3(n, sc.cam_z);
      g->te_a[i] = (float)a; g->te_b[i] = (float)b; g->te_c[i] = (float)c;
      g->te_i[i] = g->te_a[i] != 0.0f ? -1.0f / g->te_a[i] : 0.0f;
    } else { g->te_a[i] = 0; g->te_b[i] = 0; g->te_c[i] = 1; g->te_i[i] = 0; }   // the infinite plane: always inside
  } else if (t == 16) {
    for (int c = 0; c < 3; c++) { g->o[c] = (float)sc.cam_o[c]; g->X[c] = (float)sc.cam_x[c]; g->Y[c] = (float)sc.cam_y[c]; g->Z[c] = (float)sc.cam_z[c]; }
    g->inv_f = (float)(1.0 / sc.focal); g->tz = (float)m->table_z; g->zfar = (float)m->cam_zfar;
    real rel[3] = {sc.cam_o[0] - sc.cube_p[0], sc.cam_o[1] - sc.cube_p[1], sc.cam_o[2] - sc.cube_p[2]}, v[3];
    matT_vec3(v, sc.cube_R, rel); for (int c = 0; c < 3; c++) g->ol[c] = (float)v[c];
    matT_vec3(v, sc.cube_R, sc.cam_x); for (int c = 0; c < 3; c++) g->DX[c] = (float)v[c];
    matT_vec3(v, sc.cube_R, sc.cam_y); for (int c = 0; c < 3; c++) g->DY[c] = (float)v[c];
    matT_vec3(v, sc.cube_R, sc.cam_z); for (int c = 0; c < 3; c++) g->DZ[c] = (float)v[c];
    for (int c = 0; c < 3; c++) g->half[c] = (float)m->cube_half[c];
    for (int c = 0; c < 9; c++) g->R[c] = (float)sc.cube_R[c];
    g->nsph = dm->x.nvis;
    g->tab_L = 0.3f * (0.57735026919f + 0.57735026919f + 0.70710678119f);    // sum_l max(0, L_l . (0,0,1)), scene.xml:11-13
  }
  __syncthreads();
  if (t == 0) {
    real r0 = 1e30, r1 = -1e30, c0 = 1e30, c1 = -1e30;
    bool ok = true;
    for (int k = 0; k < 8; k++) {
      ok = ok && tmp->ok[k];
      r0 = fmin(r0, tmp->row[k]); r1 = fmax(r1, tmp->row[k]); c0 = fmin(c0, tmp->col[k]); c1 = fmax(c1, tmp->col[k]);
    }
    put(0, ok, r0, r1, c0, c1);
    for (int k = 0; k < 4; k++) {
      int u = g->box[0][k];
      for (int o = 1; o <= g->nsph; o++) u = (k & 1) ? max(u, g->box[o][k]) : min(u, g->box[o][k]);
      g->ubox[k] = u;
    }
  }
}
// ---- depth (BASELINE config 5) -------------------------------------------------------------------------------------
// float64 ray maths (the parity bar against the float64 oracle is 1e-6 m; float32 intersection arithmetic cancels to ~1e-5 m).
// Round 4: the pixel loop was bound by what it FETCHED per pixel, not by its arithmetic -- scene constants re-read from LDS and
// model scalars from global memory inside the loop, a walk over all twelve collider spheres testing each one's `visible` flag.
// Now the per-env scene a ray meets is hoisted into registers once (camera basis and the same basis turned into the cube frame,
// so the slab test's direction is linear in the pixel coordinates; the visible spheres' centres from the host-built list), the
// divides are reciprocal + Newton steps, and nothing but the image is touched inside the loop.  (A float32 pre-classification
// with a float64 evaluation of the winner was built and measured first: 12 % of the pixels fell within its margins, i.e. nearly
// every wave ran both paths -- 0.155 ms against round 3's 0.089; dropped.)
struct DepthScene { real ol[3], DX[3], DY[3], DZ[3]; real oc[KM_RENDER_MAXVIS][3], cc[KM_RENDER_MAXVIS]; int nvis; };

template <bool COLFIXED>      // COLFIXED: blockDim.x is a whole number of image rows (the launcher knows)
__global__ __launch_bounds__(256, 4) void k_render_depth(const KDeviceModel* __restrict__ dm, KDeviceState st, int cam, int height, int width,
                                                      float* __restrict__ depth) {
  __shared__ RenderScene sc;
  __shared__ DepthScene ds;
  const KModelDesc* m = &dm->d;
  const int env = blockIdx.x;
  // model scalars of the pixel loop: wave-uniform reads, issued in front of the set-up (they used to start after its last barrier)
  const real zfar = m->cam_zfar, znear = m->cam_znear, tabz = m->table_z;
  const real rx0 = m->table_rect[0], rx1 = m->table_rect[1], ry0 = m->table_rect[2], ry1 = m->table_rect[3];
  const real hf0 = m->cube_half[0], hf1 = m->cube_half[1], hf2 = m->cube_half[2];
  RenderPre pre;
  render_fk(dm, st, env, cam, &sc, pre);
  render_camera(dm, st, env, cam, height, &sc, pre);
  __syncthreads();
  if (threadIdx.x < 4) {
    // ray origin and basis in the cube frame, one vector per lane
    const int k = threadIdx.x;
    const real rel[3] = {sc.cam_o[0] - sc.cube_p[0], sc.cam_o[1] - sc.cube_p[1], sc.cam_o[2] - sc.cube_p[2]};
    const real* src = k == 0 ? rel : (k == 1 ? sc.cam_x : (k == 2 ? sc.cam_y : sc.cam_z));
    real* dst = k == 0 ? ds.ol : (k == 1 ? ds.DX : (k == 2 ? ds.DY : ds.DZ));
    matT_vec3(dst, sc.cube_R, src);
  } else if (threadIdx.x >= 64 && threadIdx.x < 64 + KM_RENDER_MAXVIS) {
    const int ns = threadIdx.x - 64;
    if (ns == 0) ds.nvis = dm->x.nvis;
    if (ns < dm->x.nvis) {
      const int sp = dm->x.vis_sphere[ns];
      const real rad = m->sphere_radius[sp];
      const real oc[3] = {sc.cam_o[0] - sc.sph[sp][0], sc.cam_o[1] - sc.sph[sp][1], sc.cam_o[2] - sc.sph[sp][2]};
      ds.oc[ns][0] = oc[0]; ds.oc[ns][1] = oc[1]; ds.oc[ns][2] = oc[2];
      ds.cc[ns] = dot3(oc, oc) - rad * rad;
    }
  }
  __syncthreads();
  // ---- everything the pixel loop reads, in registers
  const real k0 = tabz - sc.cam_o[2], ox = sc.cam_o[0], oy = sc.cam_o[1];
  real X[3], Y[3], Z[3], ol[3], DX[3], DY[3], DZ[3];
  const real hf[3] = {hf0, hf1, hf2};
#pragma unroll
  for (int c = 0; c < 3; c++) {
    X[c] = sc.cam_x[c]; Y[c] = sc.cam_y[c]; Z[c] = sc.cam_z[c];
    ol[c] = ds.ol[c]; DX[c] = ds.DX[c]; DY[c] = ds.DY[c]; DZ[c] = ds.DZ[c];
  }
  // |camera - cube centre|^2 - (bounding radius)^2, the radius padded by a relative 1e-9 so that roundoff never rejects a grazing ray
  const real cube_cc = (ol[0] * ol[0] + ol[1] * ol[1] + ol[2] * ol[2]) - (hf[0] * hf[0] + hf[1] * hf[1] + hf[2] * hf[2]) * (1.0 + 1e-9);
  const int nvis = ds.nvis;
  real soc[KM_RENDER_MAXVIS][3], scc[KM_RENDER_MAXVIS];
#pragma unroll
  for (int s = 0; s < KM_RENDER_MAXVIS; s++) {
    const int k = s < nvis ? s : 0;                      // (unused entries: finite copies, never tested)
    soc[s][0] = ds.oc[k][0]; soc[s][1] = ds.oc[k][1]; soc[s][2] = ds.oc[k][2]; scc[s] = ds.cc[k];
  }
  const int npix = height * width;
  const real inv_f = 1.0 / sc.focal, hw = 0.5 * width, hh = 0.5 * height;
  float* __restrict__ out = depth + (size_t)env * npix;
  int r = threadIdx.x / width, c = threadIdx.x - r * width;         // row / column advance incrementally (no division in the loop)
  const int dr = blockDim.x / width, dc = blockDim.x - dr * width;
  // Round 5: when the workgroup's stride is a whole number of image rows (64-wide images: 128 lanes = two rows) a lane keeps its
  // COLUMN for the whole loop, so everything that is linear in the pixel coordinates has a per-lane constant part.
  // The loop is bound by the NUMBER of float64 instructions a pixel issues (a wave is one image row): what a ray rarely needs is
  // computed where it is needed -- the direction in the cube frame and the slab reciprocals behind the bounding-sphere test (itself
  // in the world frame: a rotation changes neither dot product), reciprocals take one Newton step (v_rcp_f64 is good to 2^-23: one
  // step gives 2^-46, the depth bar is 1e-6 m), and the final clamp is one v_med3_f32 after the conversion (rounding is monotonic).
  // Round 6 (column kept): the ray direction is d = e + Y dy with e fixed per lane, so every dot product with d is ONE FMA in dy
  // against two per-lane constants (|d|^2: two) instead of three; dy itself advances by a constant; the table's rectangle is
  // tested as half-width minus |offset from its centre| (three instructions for the four edges' seven).
  constexpr bool colfixed = COLFIXED;
  const real dxl = (c + 0.5 - hw) * inv_f;
  real ex[3], bx[3];
#pragma unroll
  for (int a = 0; a < 3; a++) { ex[a] = X[a] * dxl - Z[a]; bx[a] = DX[a] * dxl - DZ[a]; }
  const real rel[3] = {sc.cam_o[0] - sc.cube_p[0], sc.cam_o[1] - sc.cube_p[1], sc.cam_o[2] - sc.cube_p[2]};   // (|rel| = |ol|)
  real A0 = 0, A1 = 0, A2 = 0, B0 = 0, B1 = 0, S0[KM_RENDER_MAXVIS], S1[KM_RENDER_MAXVIS];
#pragma unroll
  for (int s = 0; s < KM_RENDER_MAXVIS; s++) { S0[s] = 0; S1[s] = 0; }
  if constexpr (colfixed) {
    A0 = dot3(ex, ex); A1 = 2.0 * dot3(ex, Y); A2 = dot3(Y, Y);
    B0 = dot3(ex, rel); B1 = dot3(Y, rel);
#pragma unroll
    for (int s = 0; s < KM_RENDER_MAXVIS; s++) { S0[s] = dot3(ex, soc[s]); S1[s] = dot3(Y, soc[s]); }
  }
  // table rectangle by centre and half-widths (an unbounded side: the four-edge form below)
  const bool centred = isfinite(rx0) && isfinite(rx1) && isfinite(ry0) && isfinite(ry1);
  const real tcx = centred ? 0.5 * (rx0 + rx1) : 0.0, tcy = centred ? 0.5 * (ry0 + ry1) : 0.0;
  const real thx = 0.5 * (rx1 - rx0), thy = 0.5 * (ry1 - ry0);
  const real oxc = ox - tcx, oyc = oy - tcy;
  const float znf = (float)znear, zff = (float)zfar;
  const real row0 = hh - 0.5;                            // dy = (row0 - r) / f
  real rd = (real)r;
  const real drd = (real)dr;
  real dyc = (row0 - rd) * inv_f;
  const real ddy = drd * inv_f;
  real zfv = zfar;
  asm volatile("" : "+v"(zfv));                          // (kept in a register pair: the selects below cannot take it from SGPRs next to VCC)
  auto rcp1 = [](real x) { real q = __builtin_amdgcn_rcp(x); return q + q * (1.0 - x * q); };
  // sqrt(x), x > 0, to 2^-46: v_rsq_f64 (2^-23) and one coupled step (g ~ sqrt x, h ~ 1 / (2 sqrt x): g += g (1/2 - g h)).
  // x = 0 gives NaN, which fails the comparisons of the hit it would have been (a ray tangent to a sphere to the last bit)
  auto sqrt1 = [](real x) { const real y = __builtin_amdgcn_rsq(x); const real g = x * y, h = 0.5 * y; return g + g * (0.5 - g * h); };
  // (a wave-uniform trip count: the loop control is scalar, the lane's pixel index one add)
  const int nit = (npix + (int)blockDim.x - 1) / (int)blockDim.x;
  float* __restrict__ op = out + threadIdx.x;
  const int pstep = blockDim.x;
  int p = threadIdx.x;
  for (int it = 0; it < nit; it++, p += pstep, op += pstep) {
    real dx, dy, d0, d1, d2, a2, bc;
    if constexpr (colfixed) {
      dx = dxl; dy = dyc; dyc -= ddy;
      d0 = __builtin_fma(Y[0], dy, ex[0]); d1 = __builtin_fma(Y[1], dy, ex[1]); d2 = __builtin_fma(Y[2], dy, ex[2]);
      a2 = __builtin_fma(dy, __builtin_fma(dy, A2, A1), A0);
      bc = __builtin_fma(dy, B1, B0);
    } else {
      dx = (c + 0.5 - hw) * inv_f; dy = -(r + 0.5 - hh) * inv_f;
      c += dc; r += dr;
      if (c >= width) { c -= width; r++; }
      d0 = X[0] * dx + Y[0] * dy - Z[0]; d1 = X[1] * dx + Y[1] * dy - Z[1]; d2 = X[2] * dx + Y[2] * dy - Z[2];
      a2 = d0 * d0 + d1 * d1 + d2 * d2;
      bc = d0 * rel[0] + d1 * rel[1] + d2 * rel[2];
    }
    // table top: the rectangle table_rect at z = table_z.  d2 == 0 makes t infinite or NaN, which fails `t > 0 && t < zfar`; the
    // rectangle test is then never looked at (table_rect holds no NaN: kmanip_create checks)
    real best = zfv;
    {
      const real t = k0 * rcp1(d2);
      real in;
      if (centred) {
        const real hx = __builtin_fma(t, d0, oxc), hy = __builtin_fma(t, d1, oyc);
        in = fmin(thx - fabs(hx), thy - fabs(hy));
      } else {
        const real hx = __builtin_fma(t, d0, ox), hy = __builtin_fma(t, d1, oy);
        in = fmin(fmin(hx - rx0, rx1 - hx), fmin(hy - ry0, ry1 - hy));
      }
      if (t > 0 && t < zfar && in >= 0) best = t;
    }
    // cube box: slab test in the cube frame -- only for rays that meet the box's bounding sphere (a wave is one image row or two,
    // so the test is coherent)
    if (bc * bc - a2 * cube_cc >= 0) {
      real dlv[3];
#pragma unroll
      for (int a = 0; a < 3; a++) dlv[a] = colfixed ? __builtin_fma(DY[a], dy, bx[a]) : DX[a] * dx + DY[a] * dy - DZ[a];
      real t0 = -INFINITY, t1 = INFINITY;
      bool ok = true;
#pragma unroll
      for (int a = 0; a < 3; a++) {
        const real dl = dlv[a];
        if (dl != 0) {
          const real inv = rcp1(dl);
          const real ta = (-hf[a] - ol[a]) * inv, tb = (hf[a] - ol[a]) * inv;
          t0 = fmax(t0, fmin(ta, tb)); t1 = fmin(t1, fmax(ta, tb));
        } else if (ol[a] < -hf[a] || ol[a] > hf[a]) ok = false;
      }
      if (ok && t0 <= t1 && t1 > 0) {
        const real t = t0 > 0 ? t0 : t1;
        if (t < best) best = t;
      }
    }
    // the visible spheres (finger tips): from the gripper camera they fill most of the image, so nearly every wave (one image
    // row) takes the hit path of every sphere -- 1 / |d|^2 once per pixel, a one-step square root
    const real ia2 = rcp1(a2);
#pragma unroll
    for (int s = 0; s < KM_RENDER_MAXVIS; s++) {
      if (s < nvis) {
        const real b = colfixed ? __builtin_fma(dy, S1[s], S0[s]) : d0 * soc[s][0] + d1 * soc[s][1] + d2 * soc[s][2];
        const real disc = b * b - a2 * scc[s];
        if (disc >= 0) {
          // nearest root; compared before the division: t < best  <=>  -b - sqrt(disc) < best * a2  (a2 > 0)
          const real num = -b - sqrt1(disc);
          if (num > 0 && num < best * a2) best = num * ia2;
        }
      }
    }
    if (COLFIXED || p < npix) *op = __builtin_amdgcn_fmed3f((float)best, znf, zff);     // (COLFIXED launches: npix is a multiple of the workgroup, checked by the launcher)
  }
}

// one pixel, float32: grey level * 255 of the three channels packed r | g << 8 | b << 16
// does the ray direction (dx, dy) pass through the table top?
__device__ __forceinline__ bool rgb_over_table(const RgbScene& g, float dx, float dy) {
  bool in = true;
#pragma unroll
  for (int i = 0; i < 4; i++) in = in && (g.te_a[i] * dx + (g.te_b[i] * dy + g.te_c[i])) > 0.0f;
  return in;
}
// the open interval of dx over which a row (dy) crosses the table top: lo >= hi = not at all
__device__ __forceinline__ void rgb_table_span(const RgbScene& g, float dy, float& lo, float& hi) {
  lo = -INFINITY; hi = INFINITY;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const float e = g.te_b[i] * dy + g.te_c[i], a = g.te_a[i], x = e * g.te_i[i];
    if (a > 0.0f) lo = fmaxf(lo, x);
    else if (a < 0.0f) hi = fminf(hi, x);
    else if (!(e > 0.0f)) lo = INFINITY;
  }
}

__device__ __forceinline__ uint32_t rgb_pixel(const RgbScene& g, float dx, float dy, uint32_t objs, bool tab) {
  const float dz = g.X[2] * dx + g.Y[2] * dy - g.Z[2];
  const float dd = dx * dx + dy * dy + 1.0f;                   // |d|^2: the camera axes are orthonormal
  float best = g.zfar;
  int mat = 0;
  float n0 = 0, n1 = 0, n2 = 1;
  if (tab && dz != 0.0f) { const float t = (g.tz - g.o[2]) * __builtin_amdgcn_rcpf(dz); if (t > 0 && t < best) { best = t; mat = 1; } }
  const float d0 = g.X[0] * dx + g.Y[0] * dy - g.Z[0], d1 = g.X[1] * dx + g.Y[1] * dy - g.Z[1];
  if (objs & 1u) {
    // cube box: slab test in the cube frame; the ray direction there is linear in (dx, dy)
    float t0 = -INFINITY, t1 = INFINITY, s0 = 0, s1 = 0;
    int a0 = 0, a1 = 0;
    bool ok = true;
#pragma unroll
    for (int a = 0; a < 3; a++) {
      const float dl = g.DX[a] * dx + g.DY[a] * dy - g.DZ[a], h = g.half[a], o = g.ol[a];
      if (dl != 0.0f) {
        const float inv = __builtin_amdgcn_rcpf(dl);
        float ta = (-h - o) * inv, tb = (h - o) * inv, sa = -1, sb = 1;
        if (ta > tb) { const float s = ta; ta = tb; tb = s; sa = 1; sb = -1; }
        if (ta > t0) { t0 = ta; a0 = a; s0 = sa; }
        if (tb < t1) { t1 = tb; a1 = a; s1 = sb; }
      } else if (o < -h || o > h) ok = false;
    }
    if (ok && t0 <= t1 && t1 > 0) {
      const bool front = t0 > 0;
      const float t = front ? t0 : t1;
      if (t < best) {
        best = t; mat = 2;
        const int ax = front ? a0 : a1;
        const float sg = front ? s0 : s1;
        n0 = sg * g.R[ax]; n1 = sg * g.R[3 + ax]; n2 = sg * g.R[6 + ax];
      }
    }
  }
  for (int s = 0; s < g.nsph; s++) {
    if (objs >> (1 + s) & 1u) {
      const float b = d0 * g.oc[s][0] + d1 * g.oc[s][1] + dz * g.oc[s][2], disc = b * b - dd * g.cc[s];
      if (disc >= 0) {
        const float t = (-b - __builtin_sqrtf(disc)) * __builtin_amdgcn_rcpf(dd);
        if (t > 0 && t < best) {
          best = t; mat = 3;
          n0 = (g.oc[s][0] + t * d0) * g.ir[s]; n1 = (g.oc[s][1] + t * d1) * g.ir[s]; n2 = (g.oc[s][2] + t * dz) * g.ir[s];
        }
      }
    }
  }
  if (mat == 0) return 0u;
  const float rs = __builtin_amdgcn_rsqf(dd);
  float I;
  if (mat == 1) I = 0.4f + 0.4f * fmaxf(0.0f, -dz * rs) + g.tab_L;
  else {
    const float r3 = 0.57735026919f, r2 = 0.70710678119f;
    I = 0.4f + 0.4f * fmaxf(0.0f, -(n0 * d0 + n1 * d1 + n2 * dz) * rs)
        + 0.3f * (fmaxf(0.0f, (-n0 - n1 + n2) * r3) + fmaxf(0.0f, (n0 - n1 + n2) * r3) + fmaxf(0.0f, (n1 + n2) * r2));
  }
  I = fminf(I, 1.0f) * 255.0f;
  if (mat == 1) { const uint32_t v = (uint32_t)(0.2f * I + 0.5f); return v | (v << 8) | (v << 16); }
  if (mat == 2) return (uint32_t)(I + 0.5f);                                         // cube: rgba 1 0 0
  const uint32_t v = (uint32_t)(0.647059f * I + 0.5f);
  return v | (v << 8) | (v << 16);
}

// grid = envs x jobs: blockIdx.y picks the camera / resolution / output buffer (all the cameras of a *Vision observation in one launch)
__global__ __launch_bounds__(256) void k_render_rgb(const KDeviceModel* __restrict__ dm, KDeviceState st, KRenderJobs jobs) {
  __shared__ RenderScene sc;
  __shared__ RgbScene g;
  __shared__ RgbTmp tmp;
  const int env = blockIdx.x, job = blockIdx.y;
  const int cam = jobs.cam[job], height = jobs.height[job], width = jobs.width[job];
  uint8_t* __restrict__ rgb = jobs.rgb[job];
  RenderPre pre;
  render_fk(dm, st, env, cam, &sc, pre);
  render_camera(dm, st, env, cam, height, &sc, pre);
  __syncthreads();
  rgb_scene(dm, sc, height, width, &g, &tmp, threadIdx.x);
  __syncthreads();
  const int npix = height * width;
  const float hw = 0.5f * width, hh = 0.5f * height, inv_f = g.inv_f;
  uint8_t* out = rgb + (size_t)env * npix * 3;
  if ((width & 3) == 0) {
    // four consecutive pixels of a row per lane: 12 bytes = three dwords.  A wave covers a 64 x 4 pixel tile (16 quads x 4 rows:
    // 192 contiguous bytes a row) rather than 256 pixels of one row, so that fewer waves straddle the objects' rectangles and
    // run both the full and the table-only path; the block walks 64 x 16 pixel tiles, tile row and column advancing
    // incrementally (no integer division)
    const int wq = width >> 2, tcols = (wq + 15) >> 4, ntile = tcols * ((height + 15) >> 4);
    uint32_t* out32 = reinterpret_cast<uint32_t*>(out);
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    int tr = 0, tc = 0;
    // Outside every bounding rectangle a ray can only meet the table plane, and the shaded grey needs no hit distance: with
    // s = sign(tz - o_z), the hit test 0 < (tz - o_z) / dz < zfar is  s dz > |tz - o_z| / zfar, and the Lambert term
    // 0.4 max(0, -dz / |d|) is clamp(s dz * rsq(|d|^2) * (-0.4 s)).  ~14 operations a pixel, one of them transcendental
    const float k0 = g.tz - g.o[2], sg = k0 < 0.0f ? -1.0f : 1.0f, thr = k0 != 0.0f ? fabsf(k0) / g.zfar : INFINITY;
    const float Xzs = sg * g.X[2], Yzs = sg * g.Y[2], Zzs = sg * g.Z[2], lam = -0.4f * sg, c1 = 0.4f + g.tab_L;
    const int nobj = g.nsph;
    // Round 5: the loop was bound by instruction issue, not by its stores (~95 instructions a quad: 0.38 of its 0.45 ms).  A lane
    // keeps its image row while the block walks the tile columns, so the row's table span (20 instructions) and its ray constants
    // are computed once per tile ROW; and a quad that lies wholly inside the span and above the horizon -- nearly all of them --
    // shades its four pixels without the three per-pixel tests.
    float lo = 0, hi = 0, dy = 0, rz = 0, rd = 0;
    bool row_ok = false;              // every pixel of this row that is inside the span also passes the horizon test
    for (int tile = 0; tile < ntile; tile++) {
      const int r = (tr << 4) + ty, qc = (tc << 4) + tx, c = qc << 2, q = r * wq + qc;
      if (tc == 0) {
        dy = -(r + 0.5f - hh) * inv_f;
        rgb_table_span(g, dy, lo, hi);
        rz = Yzs * dy - Zzs; rd = dy * dy + 1.0f;
        // s dz is linear in dx: above the threshold on the whole clipped span iff it is at both ends (+- a pixel of slack)
        const float xa = fmaxf(lo, (0.5f - hw) * inv_f) - inv_f, xb = fminf(hi, (width - 0.5f - hw) * inv_f) + inv_f;
        row_ok = (Xzs * xa + rz > thr) && (Xzs * xb + rz > thr);
      }
      if (++tc == tcols) { tc = 0; tr++; }
      if (r >= height || qc >= wq) continue;
      const float dx0 = (c + 0.5f - hw) * inv_f;
      uint32_t w0, w1, w2;
      if (r >= g.ubox[0] && r <= g.ubox[1] && c + 3 >= g.ubox[2] && c <= g.ubox[3]) {
        uint32_t objs = 0;                                  // objects whose bounding rectangle this quad touches
        for (int o = 0; o <= nobj; o++)
          objs |= (uint32_t)(r >= g.box[o][0] && r <= g.box[o][1] && c + 3 >= g.box[o][2] && c <= g.box[o][3]) << o;
        uint32_t px[4];
#pragma unroll
        for (int i = 0; i < 4; i++) { const float dx = (c + i + 0.5f - hw) * inv_f; px[i] = rgb_pixel(g, dx, dy, objs, dx > lo && dx < hi); }
        w0 = px[0] | (px[1] << 24); w1 = (px[1] >> 8) | (px[2] << 16); w2 = (px[2] >> 16) | (px[3] << 8);
      } else if (!(dx0 + 3.0f * inv_f > lo && dx0 < hi)) {
        w0 = 0; w1 = 0; w2 = 0;                                                        // beside the table: background
      } else {
        uint32_t v[4];
        if (row_ok && dx0 > lo && dx0 + 3.0f * inv_f < hi) {                  // the whole quad is table: no per-pixel tests
#pragma unroll
          for (int i = 0; i < 4; i++) {
            const float dx = dx0 + (float)i * inv_f;
            const float sdz = Xzs * dx + rz, dd = dx * dx + rd;
            const float a = __builtin_amdgcn_fmed3f(sdz * __builtin_amdgcn_rsqf(dd) * lam, 0.0f, 1.0f);
            v[i] = (uint32_t)(__builtin_amdgcn_fmed3f(a + c1, 0.0f, 1.0f) * 51.0f + 0.5f);  // table rgba .2 .2 .2: 255 * 0.2 = 51
          }
        } else {
#pragma unroll
          for (int i = 0; i < 4; i++) {
            const float dx = (c + i + 0.5f - hw) * inv_f;
            const float sdz = Xzs * dx + rz, dd = dx * dx + rd;
            const float a = __builtin_amdgcn_fmed3f(sdz * __builtin_amdgcn_rsqf(dd) * lam, 0.0f, 1.0f);
            const float I = __builtin_amdgcn_fmed3f(a + c1, 0.0f, 1.0f);
            v[i] = (sdz > thr && dx > lo && dx < hi) ? (uint32_t)(51.0f * I + 0.5f) : 0u;
          }
        }
        // grey pixels: the three dwords are byte replications of the four values
        w0 = __builtin_amdgcn_perm(v[1], v[0], 0x04000000u);                         // v0 v0 v0 v1
        w1 = __builtin_amdgcn_perm(v[2], v[1], 0x04040000u);                         // v1 v1 v2 v2
        w2 = __builtin_amdgcn_perm(v[3], v[2], 0x04040400u);                         // v2 v3 v3 v3
      }
      out32[3 * q] = w0; out32[3 * q + 1] = w1; out32[3 * q + 2] = w2;       // (one global_store_dwordx3; non-temporal stores measured 11 % slower)
    }
  } else {
    for (int p = threadIdx.x; p < npix; p += blockDim.x) {
      const int r = p / width, c = p - r * width;
      uint32_t objs = 0;
      for (int o = 0; o <= g.nsph; o++)
        objs |= (uint32_t)(r >= g.box[o][0] && r <= g.box[o][1] && c >= g.box[o][2] && c <= g.box[o][3]) << o;
      const float dx = (c + 0.5f - hw) * inv_f, dy = -(r + 0.5f - hh) * inv_f;
      const uint32_t v = rgb_pixel(g, dx, dy, objs, rgb_over_table(g, dx, dy));
      out[3 * (size_t)p] = (uint8_t)v; out[3 * (size_t)p + 1] = (uint8_t)(v >> 8); out[3 * (size_t)p + 2] = (uint8_t)(v >> 16);
    }
  }
}

void kmanip_launch_render_depth(const KDeviceModel* dm, const KDeviceState& st, int cam, int height, int width, float* depth,
                                hipStream_t stream) {
  // 128 lanes per env: two waves -- 2048 envs x 2 waves fill the chip's 4096 wave slots (120 registers: four waves per SIMD) in one round
  if (width > 0 && 128 % width == 0 && (height * width) % 128 == 0) hipLaunchKernelGGL(k_render_depth<true>, dim3(st.num_envs), dim3(128), 0, stream, dm, st, cam, height, width, depth);
  else hipLaunchKernelGGL(k_render_depth<false>, dim3(st.num_envs), dim3(128), 0, stream, dm, st, cam, height, width, depth);
}
void kmanip_launch_render_rgb(const KDeviceModel* dm, const KDeviceState& st, const KRenderJobs& jobs, hipStream_t stream) {
  hipLaunchKernelGGL(k_render_rgb, dim3(st.num_envs, jobs.n), dim3(256), 0, stream, dm, st, jobs);
}
