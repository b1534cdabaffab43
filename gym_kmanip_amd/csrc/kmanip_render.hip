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
// jumping through LDS (the same scheme as k_step's fk_parallel); then every lane ray-casts pixels p = lane, lane + 256,
// ... so each wave writes 64 consecutive pixels (coalesced).  The kernel is HBM-write bound: 4 B (depth) or 3 B (rgb) per
// pixel against ~100 FLOP of ray/primitive tests.  Ray maths stay in float64 like the rest of the path: the depth parity
// bar against the float64 oracle is 1e-6 m, which float32 intersection arithmetic (cancellation ~1e-5 m) would not meet,
// and the kernel is bound by its stores, not by FP64 issue.
#include "kmanip_device.hpp"

struct RenderScene {
  real xpos[KM_MAX_LINKS][3], xmat[KM_MAX_LINKS][9];
  real cam_o[3], cam_x[3], cam_y[3], cam_z[3];
  real cube_p[3], cube_R[9];
  real sph[KM_MAX_SPHERES][3];
  real focal;
};

// mj_kinematics, one link per lane + pointer jumping (block-wide barriers: the workgroup is 4 waves)
__device__ __forceinline__ void render_fk(const KDeviceModel* dm, const KDeviceState& st, int env, RenderScene* sc) {
  const KModelDesc* m = &dm->d;
  const int nl = m->nlink, NE = st.num_envs, i = threadIdx.x;
  const bool on = i < nl;
  real R[9], p[3];
  if (on) {
    real Rl[9], q4[4] = {m->link_quat[i][0], m->link_quat[i][1], m->link_quat[i][2], m->link_quat[i][3]};
    normalize4(q4);
    quat2mat(Rl, q4);
    p[0] = m->link_pos[i][0]; p[1] = m->link_pos[i][1]; p[2] = m->link_pos[i][2];
    const real q = st.qpos[(size_t)i * NE + env];
    if (m->jnt_type[i] == KM_JNT_SLIDE) {
#pragma unroll
      for (int c = 0; c < 9; c++) R[c] = Rl[c];
      p[0] += Rl[2] * q; p[1] += Rl[5] * q; p[2] += Rl[8] * q;
    } else {
      real sn, cs;
      sincos(q, &sn, &cs);
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
  const int rounds = dm->x.fk_rounds;
  for (int k = 0; k < rounds; k++) {
    const int a = on ? dm->x.jump[k][i] : -1;
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

__device__ __forceinline__ void render_camera(const KDeviceModel* dm, const KDeviceState& st, int env, int cam, int height, RenderScene* sc) {
  const KModelDesc* m = &dm->d;
  const int nl = m->nlink, NE = st.num_envs;
  // camera frame (mj_camlight, targetbody): z = (cam - target)/|.|, x = (0,0,1) x z, y = z x x; a link of -1 = world frame
  const int cl = m->cam_link[cam], tl = m->cam_target_link[cam];
  real co[3], to[3], t[3];
  real cp[3] = {m->cam_pos[cam][0], m->cam_pos[cam][1], m->cam_pos[cam][2]};
  real tp[3] = {m->cam_target_pos[cam][0], m->cam_target_pos[cam][1], m->cam_target_pos[cam][2]};
  if (cl < 0) { co[0] = cp[0]; co[1] = cp[1]; co[2] = cp[2]; }
  else { mat_vec3(t, sc->xmat[cl], cp); co[0] = sc->xpos[cl][0] + t[0]; co[1] = sc->xpos[cl][1] + t[1]; co[2] = sc->xpos[cl][2] + t[2]; }
  if (tl < 0) { to[0] = tp[0]; to[1] = tp[1]; to[2] = tp[2]; }
  else { mat_vec3(t, sc->xmat[tl], tp); to[0] = sc->xpos[tl][0] + t[0]; to[1] = sc->xpos[tl][1] + t[1]; to[2] = sc->xpos[tl][2] + t[2]; }
  real z[3] = {co[0] - to[0], co[1] - to[1], co[2] - to[2]}, up[3] = {0, 0, 1}, x[3], y[3];
  normalize3(z);
  cross3(x, up, z); normalize3(x);
  cross3(y, z, x); normalize3(y);
  for (int c = 0; c < 3; c++) { sc->cam_o[c] = co[c]; sc->cam_x[c] = x[c]; sc->cam_y[c] = y[c]; sc->cam_z[c] = z[c]; }
  sc->focal = (0.5 * height) / tan(0.5 * m->cam_fovy[cam] * (M_PI / 180.0));
  real cq[4];
  for (int c = 0; c < 3; c++) sc->cube_p[c] = st.qpos[(size_t)(nl + c) * NE + env];
  for (int c = 0; c < 4; c++) cq[c] = st.qpos[(size_t)(nl + 3 + c) * NE + env];
  normalize4(cq);
  quat2mat(sc->cube_R, cq);
  for (int s = 0; s < m->nsphere; s++) {
    const int l = m->sphere_link[s];
    real sl[3] = {m->sphere_pos[s][0], m->sphere_pos[s][1], m->sphere_pos[s][2]};
    mat_vec3(t, sc->xmat[l], sl);
    sc->sph[s][0] = sc->xpos[l][0] + t[0]; sc->sph[s][1] = sc->xpos[l][1] + t[1]; sc->sph[s][2] = sc->xpos[l][2] + t[2];
  }
}

// nearest hit of the ray o + t d with the surrogate scene: returns t (zfar if none), the surface normal and the material id
// (0 = none, 1 = table, 2 = cube, 3 = finger sphere)
__device__ __forceinline__ real cast_ray(const KModelDesc* m, const RenderScene& sc, const real* d, real zfar, real* nrm, int& mat) {
  real best = zfar;
  mat = 0;
  nrm[0] = 0; nrm[1] = 0; nrm[2] = 1;
  // table plane z = table_z
  if (d[2] != 0) { real t = (m->table_z - sc.cam_o[2]) / d[2]; if (t > 0 && t < best) { best = t; mat = 1; } }
  // cube box (slab test in the cube frame)
  {
    real rel[3] = {sc.cam_o[0] - sc.cube_p[0], sc.cam_o[1] - sc.cube_p[1], sc.cam_o[2] - sc.cube_p[2]}, ol[3], dl[3];
    matT_vec3(ol, sc.cube_R, rel);
    matT_vec3(dl, sc.cube_R, d);
    real t0 = -INFINITY, t1 = INFINITY;
    int a0 = 0, a1 = 0;
    real s0 = 0, s1 = 0;
    bool ok = true;
#pragma unroll
    for (int a = 0; a < 3; a++) {
      const real h = m->cube_half[a];
      if (dl[a] != 0) {
        real ta = (-h - ol[a]) / dl[a], tb = (h - ol[a]) / dl[a];
        real sa = -1, sb = 1;                      // outward normal sign of the face each root lies on
        if (ta > tb) { real s = ta; ta = tb; tb = s; sa = 1; sb = -1; }
        if (ta > t0) { t0 = ta; a0 = a; s0 = sa; }
        if (tb < t1) { t1 = tb; a1 = a; s1 = sb; }
      } else if (ol[a] < -h || ol[a] > h) ok = false;
    }
    if (ok && t0 <= t1 && t1 > 0) {
      const bool front = t0 > 0;
      const real t = front ? t0 : t1;
      if (t < best) {
        best = t; mat = 2;
        const int ax = front ? a0 : a1;
        const real sg = front ? s0 : s1;
        nrm[0] = sg * sc.cube_R[ax]; nrm[1] = sg * sc.cube_R[3 + ax]; nrm[2] = sg * sc.cube_R[6 + ax];
      }
    }
  }
  // finger spheres (the link spheres are collision-only surrogates: they would wall in the wrist cameras)
  for (int s = 0; s < m->nsphere; s++) {
    if (!m->sphere_visible[s]) continue;
    real oc[3] = {sc.cam_o[0] - sc.sph[s][0], sc.cam_o[1] - sc.sph[s][1], sc.cam_o[2] - sc.sph[s][2]};
    const real a = dot3(d, d), b = dot3(d, oc), cc = dot3(oc, oc) - m->sphere_radius[s] * m->sphere_radius[s];
    const real disc = b * b - a * cc;
    if (disc >= 0) {
      real t = (-b - sqrt(disc)) / a;
      if (t > 0 && t < best) {
        best = t; mat = 3;
        const real ir = 1.0 / m->sphere_radius[s];
        nrm[0] = (oc[0] + t * d[0]) * ir; nrm[1] = (oc[1] + t * d[1]) * ir; nrm[2] = (oc[2] + t * d[2]) * ir;
      }
    }
  }
  return best;
}

template <bool RGB>
__global__ __launch_bounds__(256) void k_render(const KDeviceModel* __restrict__ dm, KDeviceState st, int cam, int height, int width,
                                                float* __restrict__ depth, uint8_t* __restrict__ rgb) {
  __shared__ RenderScene sc;
  const KModelDesc* m = &dm->d;
  const int env = blockIdx.x;
  render_fk(dm, st, env, &sc);
  if (threadIdx.x == 0) render_camera(dm, st, env, cam, height, &sc);
  __syncthreads();
  const real zfar = m->cam_zfar, znear = m->cam_znear;
  const int npix = height * width;
  const real inv_f = 1.0 / sc.focal;
  // directions TO the three scene lights (scene.xml:11-13: dir = (1,1,-1), (-1,1,-1), (0,-1,-1), normalised)
  const real r3 = 0.57735026918962576451, r2 = 0.70710678118654752440;
  const real L[3][3] = {{-r3, -r3, r3}, {r3, -r3, r3}, {0, r2, r2}};
  const real col[4][3] = {{0, 0, 0}, {0.2, 0.2, 0.2}, {1, 0, 0}, {0.647059, 0.647059, 0.647059}};   // none, table, cube, finger
  for (int p = threadIdx.x; p < npix; p += blockDim.x) {
    const int r = p / width, c = p - r * width;
    const real dx = (c + 0.5 - 0.5 * width) * inv_f, dy = -(r + 0.5 - 0.5 * height) * inv_f;
    real d[3] = {sc.cam_x[0] * dx + sc.cam_y[0] * dy - sc.cam_z[0], sc.cam_x[1] * dx + sc.cam_y[1] * dy - sc.cam_z[1],
                 sc.cam_x[2] * dx + sc.cam_y[2] * dy - sc.cam_z[2]};
    real nrm[3];
    int mat;
    const real best = cast_ray(m, sc, d, zfar, nrm, mat);
    if constexpr (!RGB) {
      depth[(size_t)env * npix + p] = (float)fmin(fmax(best, znear), zfar);
    } else {
      real I = 0;
      if (mat != 0) {
        const real dn = 1.0 / sqrt(dot3(d, d));
        const real head = fmax(0.0, -(nrm[0] * d[0] + nrm[1] * d[1] + nrm[2] * d[2]) * dn);      // headlight at the camera
        I = 0.4 + 0.4 * head;
#pragma unroll
        for (int l = 0; l < 3; l++) I += 0.3 * fmax(0.0, nrm[0] * L[l][0] + nrm[1] * L[l][1] + nrm[2] * L[l][2]);
        I = fmin(I, 1.0);
      }
      uint8_t* o = rgb + ((size_t)env * npix + p) * 3;
      o[0] = (uint8_t)(255.0 * col[mat][0] * I + 0.5); o[1] = (uint8_t)(255.0 * col[mat][1] * I + 0.5); o[2] = (uint8_t)(255.0 * col[mat][2] * I + 0.5);
    }
  }
}

void kmanip_launch_render_depth(const KDeviceModel* dm, const KDeviceState& st, int cam, int height, int width, float* depth,
                                hipStream_t stream) {
  hipLaunchKernelGGL(k_render<false>, dim3(st.num_envs), dim3(256), 0, stream, dm, st, cam, height, width, depth, (uint8_t*)nullptr);
}
void kmanip_launch_render_rgb(const KDeviceModel* dm, const KDeviceState& st, int cam, int height, int width, uint8_t* rgb,
                              hipStream_t stream) {
  hipLaunchKernelGGL(k_render<true>, dim3(st.num_envs), dim3(256), 0, stream, dm, st, cam, height, width, (float*)nullptr, rgb);
}
