// tools/libm_check.hip -- accuracy of kmanip_math.hpp against the ROCm device library, on the GPU.
//   hipcc -O3 --offload-arch=gfx950 tools/libm_check.hip -o tools/_build/libm_check && tools/_build/libm_check
#include "../gym_kmanip_amd/csrc/kmanip_math.hpp"
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void k(const double* a, const double* b, double* out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s, c, s2, c2;
  km_sincos(a[i], &s, &c);
  sincos(a[i], &s2, &c2);
  out[8 * i + 0] = s; out[8 * i + 1] = s2; out[8 * i + 2] = c; out[8 * i + 3] = c2;
  out[8 * i + 4] = km_atan2(a[i], b[i]); out[8 * i + 5] = atan2(a[i], b[i]);
  out[8 * i + 6] = km_sqrt(fabs(b[i])); out[8 * i + 7] = sqrt(fabs(b[i]));
}
static double ulps(double x, double ref) {
  if (x == ref) return 0;
  double u = std::nextafter(std::fabs(ref), INFINITY) - std::fabs(ref);
  return std::fabs(x - ref) / u;
}
int main() {
  const int n = 1 << 22;
  std::vector<double> a(n), b(n), o(8 * (size_t)n);
  uint64_t s = 88172645463325252ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0; };
  for (int i = 0; i < n; i++) {
    double r = rnd();
    a[i] = (i & 1) ? (2 * rnd() - 1) * 7.0 : (2 * rnd() - 1) * std::pow(10.0, -8 * r);   // angles within ~2 pi, and tiny ones
    b[i] = (i & 2) ? (2 * rnd() - 1) * 7.0 : (2 * rnd() - 1) * std::pow(10.0, -8 * rnd());
    if (i < 8) { a[i] = (i & 1) ? 0.0 : 1.0; b[i] = (i & 2) ? 0.0 : ((i & 4) ? -1.0 : 1.0); }
  }
  double *da, *db, *dout;
  hipMalloc(&da, n * 8); hipMalloc(&db, n * 8); hipMalloc(&dout, 8 * (size_t)n * 8);
  hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, da, db, dout, n);
  hipMemcpy(o.data(), dout, 8 * (size_t)n * 8, hipMemcpyDeviceToHost);
  double us = 0, uc = 0, ua = 0, uq = 0, as_ = 0, ac = 0, aa = 0;
  for (int i = 0; i < n; i++) {
    const double* r = &o[8 * (size_t)i];
    us = std::fmax(us, ulps(r[0], r[1])); uc = std::fmax(uc, ulps(r[2], r[3])); ua = std::fmax(ua, ulps(r[4], r[5])); uq = std::fmax(uq, ulps(r[6], r[7]));
    as_ = std::fmax(as_, std::fabs(r[0] - r[1])); ac = std::fmax(ac, std::fabs(r[2] - r[3])); aa = std::fmax(aa, std::fabs(r[4] - r[5]));
  }
  printf("max ulp vs device library over %d samples: sin %.2f cos %.2f atan2 %.2f sqrt %.2f | max abs: sin %.3g cos %.3g atan2 %.3g\n", n, us, uc, ua, uq, as_, ac, aa);
  for (int i = 0; i < 8; i++) printf("  atan2(%g, %g) = %.17g (lib %.17g)\n", a[i], b[i], o[8 * (size_t)i + 4], o[8 * (size_t)i + 5]);
  return (us < 4 && uc < 4 && ua < 4 && uq < 2) ? 0 : 1;
}
