// kmanip_render.hip -- gripper-camera depth render of every env's current state (BASELINE.json config 5).
//
// Replaces the camera branch of KManipTask.get_observation / KManipEnvSim.k_render (reference
// gym_kmanip/env_sim.py:140-145,187-188; cameras grip_r / grip_l, mode="targetbody", fovy 20:
// arm_r_body.xml:68, arm_l_body.xml:68, torso_body.xml:104,173) with the output BASELINE config 5 names:
// a float32 depth image (metres along the optical axis).  The rendered scene is the build's surrogate
// geometry = its collision primitives (cube box, table plane, finger spheres); the reference's robot meshes
// are absent from the checkout.
//
// One workgroup (256 lanes) per env: lane 0 runs the forward kinematics (rotation-matrix propagation) and
// the MuJoCo targetbody camera frame into LDS; then every lane ray-casts pixels p = lane, lane + 256, ...
// so each wave writes 64 consecutive floats (coalesced 256-B stores).  The kernel is HBM-write bound:
// 4 B per pixel against ~60 FLOP of ray/primitive tests.
#include "kmanip_device.hpp"

struct RenderScene {
  real cam_o[3], cam_x[3], cam_y[3], cam_z[3];
  real cube_p[3], cube_R[9];
  real sph[KM_MAX_SPHERES][3];
  real focal;
};

__device__ void render_setup(const KDeviceModel* dm, const KDeviceState& st, int env, int cam, int height, RenderScene* sc) {
  const KModelDesc* m = &dm->d;
  const int nl = m->nlink, NE = st.num_envs;
  real xpos[KM_MAX_LINKS][3], xmat[KM_MAX_LINKS][9];
  for (int i = 0; i < nl; i++) {
    const int p = m->link_parent[i];
    real Rl[9], q4[4] = {m->link_quat[i][0], m->link_quat[i][1], m->link_quat[i][2], m->link_quat[i][3]};
    normalize4(q4);
    quat2mat(Rl, q4);
    real pos[3] = {m->link_pos[i][0], m->link_pos[i][1], m->link_pos[i][2]}, R1[9];
    if (p < 0) { for (int c = 0; c < 9; c++) R1[c] = Rl[c]; }
    else {
      real t[3];
      mat_vec3(t, xmat[p], pos);
      pos[0] = xpos[p][0] + t[0]; pos[1] = xpos[p][1] + t[1]; pos[2] = xpos[p][2] + t[2];
      for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++)
        R1[3 * a + b] = xmat[p][3 * a] * Rl[b] + xmat[p][3 * a + 1] * Rl[3 + b] + xmat[p][3 * a + 2] * Rl[6 + b];
    }
    const real q = st.qpos[(size_t)i * NE + env];
    if (m->jnt_type[i] == KM_JNT_SLIDE) {
      for (int c = 0; c < 9; c++) xmat[i][c] = R1[c];
      pos[0] += R1[2] * q; pos[1] += R1[5] * q; pos[2] += R1[8] * q;
    } else {
      real sn, cs;
      sincos(q, &sn, &cs);
      for (int a = 0; a < 3; a++) {
        xmat[i][3 * a] = cs * R1[3 * a] + sn * R1[3 * a + 1];
        xmat[i][3 * a + 1] = cs * R1[3 * a + 1] - sn * R1[3 * a];
        xmat[i][3 * a + 2] = R1[3 * a + 2];
      }
    }
    xpos[i][0] = pos[0]; xpos[i][1] = pos[1]; xpos[i][2] = pos[2];
  }
  // camera frame (mj_camlight, targetbody): z = (cam - target)/|.|, x = (0,0,1) x z, y = z x x
  const int cl = m->cam_link[cam], tl = m->cam_target_link[cam];
  real co[3], to[3], t[3];
  real cp[3] = {m->cam_pos[cam][0], m->cam_pos[cam][1], m->cam_pos[cam][2]};
  real tp[3] = {m->cam_target_pos[cam][0], m->cam_target_pos[cam][1], m->cam_target_pos[cam][2]};
  // (a link of -1 = world frame: the fixed `top` / `head` cameras and their target, the table body)
  if (cl < 0) { co[0] = cp[0]; co[1] = cp[1]; co[2] = cp[2]; }
  else { mat_vec3(t, xmat[cl], cp); co[0] = xpos[cl][0] + t[0]; co[1] = xpos[cl][1] + t[1]; co[2] = xpos[cl][2] + t[2]; }
  if (tl < 0) { to[0] = tp[0]; to[1] = tp[1]; to[2] = tp[2]; }
  else { mat_vec3(t, xmat[tl], tp); to[0] = xpos[tl][0] + t[0]; to[1] = xpos[tl][1] + t[1]; to[2] = xpos[tl][2] + t[2]; }
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
    mat_vec3(t, xmat[l], sl);
    sc->sph[s][0] = xpos[l][0] + t[0]; sc->sph[s][1] = xpos[l][1] + t[1]; sc->sph[s][2] = xpos[l][2] + t[2];
  }
}

__global__ __launch_bounds__(256) void k_render_depth(const KDeviceModel* __restrict__ dm, KDeviceState st, int cam, int height,
                                                      int width, float* __restrict__ depth) {
  __shared__ RenderScene sc;
  const KModelDesc* m = &dm->d;
  const int env = blockIdx.x;
  if (threadIdx.x == 0) render_setup(dm, st, env, cam, height, &sc);
  __syncthreads();
  const real zfar = m->cam_zfar, znear = m->cam_znear;
  const int npix = height * width;
  float* out = depth + (size_t)env * npix;
  const real inv_f = 1.0 / sc.focal;
  for (int p = threadIdx.x; p < npix; p += blockDim.x) {
    const int r = p / width, c = p - r * width;
    const real dx = (c + 0.5 - 0.5 * width) * inv_f, dy = -(r + 0.5 - 0.5 * height) * inv_f;
    real d[3] = {sc.cam_x[0] * dx + sc.cam_y[0] * dy - sc.cam_z[0], sc.cam_x[1] * dx + sc.cam_y[1] * dy - sc.cam_z[1],
                 sc.cam_x[2] * dx + sc.cam_y[2] * dy - sc.cam_z[2]};
    real best = zfar;
    // table plane z = table_z
    if (d[2] != 0) { real t = (m->table_z - sc.cam_o[2]) / d[2]; if (t > 0 && t < best) best = t; }
    // cube box (slab test in the cube frame)
    {
      real rel[3] = {sc.cam_o[0] - sc.cube_p[0], sc.cam_o[1] - sc.cube_p[1], sc.cam_o[2] - sc.cube_p[2]}, ol[3], dl[3];
      matT_vec3(ol, sc.cube_R, rel);
      matT_vec3(dl, sc.cube_R, d);
      real t0 = -INFINITY, t1 = INFINITY;
      bool ok = true;
#pragma unroll
      for (int a = 0; a < 3; a++) {
        const real h = m->cube_half[a];
        if (dl[a] != 0) {
          real ta = (-h - ol[a]) / dl[a], tb = (h - ol[a]) / dl[a];
          if (ta > tb) { real s = ta; ta = tb; tb = s; }
          t0 = fmax(t0, ta); t1 = fmin(t1, tb);
        } else if (ol[a] < -h || ol[a] > h) ok = false;
      }
      if (ok && t0 <= t1 && t1 > 0) { real t = t0 > 0 ? t0 : t1; if (t < best) best = t; }
    }
    // finger spheres
    for (int s = 0; s < m->nsphere; s++) {
      real oc[3] = {sc.cam_o[0] - sc.sph[s][0], sc.cam_o[1] - sc.sph[s][1], sc.cam_o[2] - sc.sph[s][2]};
      const real a = dot3(d, d), b = dot3(d, oc), cc = dot3(oc, oc) - m->sphere_radius[s] * m->sphere_radius[s];
      const real disc = b * b - a * cc;
      if (disc >= 0) { real t = (-b - sqrt(disc)) / a; if (t > 0 && t < best) best = t; }
    }
    out[p] = (float)fmin(fmax(best, znear), zfar);
  }
}

void kmanip_launch_render_depth(const KDeviceModel* dm, const KDeviceState& st, int cam, int height, int width, float* depth,
                                hipStream_t stream) {
  hipLaunchKernelGGL(k_render_depth, dim3(st.num_envs), dim3(256), 0, stream, dm, st, cam, height, width, depth);
}
