// kmanip_math.hpp -- lean double-precision elementary functions for the kernels' bounded arguments (gfx950).
// The ROCm device library's sincos / atan2 / sqrt carry large-argument (Payne-Hanek) paths, inf / nan / signed-zero handling and
// IEEE-correct rounding sequences: 150-270 instructions per call, 10 % of k_step's dynamic instructions in round 2.  Joint angles,
// Euler angles and unit-quaternion components need none of that: these versions are branch-free, 30-45 instructions, and accurate
// to <= 2 ulp on their stated domains (tools/libm_check.hip compares them with the device library on the GPU).
#pragma once
#include <hip/hip_runtime.h>

// sin(x), cos(x) for |x| <~ 1e5 (joint and Euler angles are within a few pi): Cody-Waite reduction by pi/2 with two FMA steps
// (the product k * pi/2_hi is never rounded by itself), fdlibm's __kernel_sin / __kernel_cos polynomials on |r| <= pi/4, then the
// quadrant swap / signs as selects.
__device__ __forceinline__ void km_sincos(double x, double* sn, double* cs) {
  const double k = __builtin_rint(x * 6.36619772367581382433e-01);
  double r = __builtin_fma(-k, 1.57079632679489655800e+00, x);
  r = __builtin_fma(-k, 6.12323399573676603587e-17, r);
  const double z = r * r;
  double ps = __builtin_fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
  ps = __builtin_fma(z, ps, 2.75573137070700676789e-06);
  ps = __builtin_fma(z, ps, -1.98412698298579493134e-04);
  ps = __builtin_fma(z, ps, 8.33333333332248946124e-03);
  ps = __builtin_fma(z, ps, -1.66666666666666324348e-01);
  const double s = __builtin_fma(z * r, ps, r);
  double pc = __builtin_fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
  pc = __builtin_fma(z, pc, -2.75573143513906633035e-07);
  pc = __builtin_fma(z, pc, 2.48015872894767294178e-05);
  pc = __builtin_fma(z, pc, -1.38888888888741095749e-03);
  pc = __builtin_fma(z, pc, 4.16666666666666019037e-02);
  const double hz = 0.5 * z, w = 1.0 - hz;
  const double c = w + (((1.0 - w) - hz) + z * (z * pc));
  const int n = (int)k;
  const double s0 = (n & 1) ? c : s, c0 = (n & 1) ? s : c;
  *sn = (n & 2) ? -s0 : s0;
  *cs = ((n + 1) & 2) ? -c0 : c0;
}

// 1/x, 1/sqrt(x): hardware estimate + two Newton steps
__device__ __forceinline__ double km_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = __builtin_fma(r, __builtin_fma(-x, r, 1.0), r);
  r = __builtin_fma(r, __builtin_fma(-x, r, 1.0), r);
  return r;
}
// sqrt(x) for x >= 0 (0 -> 0), no denormal / inf handling: x * rsqrt(x) with one residual correction.
// DOMAIN: finite x.  +inf gives NaN (rsq(inf) = 0, 0 * inf), not inf.  Callers that can see a diverging env's numbers (the reward's
// |qvel|, the Newton / IK convergence norms) do not rely on it: k_step raises the divergence flag from isfinite(qacc) / isfinite(qpos)
// -- independent of these values --, zeroes reward and observation of such an env, and a NaN norm fails every `< tol` test, i.e. the
// loop runs to its iteration cap and the non-finite qacc is then flagged (tests: test_diverged_flag_and_recovery).
__device__ __forceinline__ double km_sqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = y * __builtin_fma(-0.5 * x * y, y, 1.5);
  double s = x * y;
  s = __builtin_fma(__builtin_fma(-s, s, x), 0.5 * y, s);
  s = __builtin_fma(__builtin_fma(-s, s, x), 0.5 * y, s);
  return x > 0 ? s : 0.0;
}

// atan2(y, x) in (-pi, pi], finite arguments, (0, 0) -> 0: a = min / max of the magnitudes, one division after folding the
// reduction atan(a) = pi/4 + atan((a - 1) / (a + 1)) for a > tan(pi/8) into its numerator and denominator, a degree-11 polynomial
// in t^2 (Chebyshev-node fit computed with 60-digit arithmetic: 2e-19 relative on |t| <= tan(pi/8)), octant fix-ups as selects.
__device__ __forceinline__ double km_atan2(double y, double x) {
  const double ax = __builtin_fabs(x), ay = __builtin_fabs(y);
  const double mx = __builtin_fmax(__builtin_fmax(ax, ay), 1e-300), mn = __builtin_fmin(ax, ay);
  const bool big = mn > 4.14213562373095048802e-01 * mx;
  const double num = big ? mn - mx : mn, den = big ? mn + mx : mx;
  const double t = num * km_rcp(den);
  const double z = t * t;
  double p = __builtin_fma(z, 0.016285756855221028291, -0.034570561981427746882);
  p = __builtin_fma(z, p, 0.045515932206265491693);
  p = __builtin_fma(z, p, -0.052304542706502445183);
  p = __builtin_fma(z, p, 0.058789289978347751327);
  p = __builtin_fma(z, p, -0.066664248857382553335);
  p = __builtin_fma(z, p, 0.076922963750321423991);
  p = __builtin_fma(z, p, -0.090909087535008768442);
  p = __builtin_fma(z, p, 0.11111111105155446565);
  p = __builtin_fma(z, p, -0.14285714285659827606);
  p = __builtin_fma(z, p, 0.19999999999999804526);
  p = __builtin_fma(z, p, -0.33333333333333333217);
  double a = __builtin_fma(t * z, p, t);                        // atan(t)
  a += big ? 7.85398163397448309616e-01 : 0.0;                   // + pi/4
  a = ay > ax ? 1.57079632679489661923 - a : a;                  // octant: atan(ay / ax) for ay > ax
  a = x < 0 ? 3.14159265358979323846 - a : a;
  return y < 0 ? -a : a;
}
