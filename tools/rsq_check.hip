// Accuracy of the hardware estimates v_rsq_f64 / v_rcp_f64 on gfx950 and of one / two Newton steps on top of them (rsqrt_nr, frcp in
// kmanip_device.hpp use two): max relative error over 2^24 samples spread over 40 binades, against host long double.
// hipcc -O3 --offload-arch=gfx950 -o tools/_build/rsq_check tools/rsq_check.hip && tools/_build/rsq_check
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double* x, double* out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double s = x[i];
  double y = __builtin_amdgcn_rsq(s);
  out[i] = y;
  y = y * (1.5 - 0.5 * s * y * y);
  out[n + i] = y;
  y = y * (1.5 - 0.5 * s * y * y);
  out[2 * n + i] = y;
  double r = __builtin_amdgcn_rcp(s);
  out[3 * n + i] = r;
  r = r + r * (1.0 - s * r);
  out[4 * n + i] = r;
  r = r + r * (1.0 - s * r);
  out[5 * n + i] = r;
}
int main() {
  const int n = 1 << 24;
  std::vector<double> x(n), o(6 * (size_t)n);
  unsigned long long st = 88172645463325252ull;
  for (int i = 0; i < n; i++) {
    st ^= st << 13; st ^= st >> 7; st ^= st << 17;
    const double u = (st >> 11) * (1.0 / 9007199254740992.0);
    x[i] = std::ldexp(1.0 + u, (int)(st % 40) - 20);
  }
  double *dx, *dout;
  hipMalloc(&dx, n * 8); hipMalloc(&dout, 6 * (size_t)n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
  hipMemcpy(o.data(), dout, 6 * (size_t)n * 8, hipMemcpyDeviceToHost);
  const char* nm[6] = {"v_rsq_f64", "  + 1 Newton step", "  + 2 Newton steps", "v_rcp_f64", "  + 1 Newton step", "  + 2 Newton steps"};
  for (int v = 0; v < 6; v++) {
    long double worst = 0;
    for (int i = 0; i < n; i++) {
      const long double ref = v < 3 ? 1.0L / sqrtl((long double)x[i]) : 1.0L / (long double)x[i];
      const long double e = fabsl(((long double)o[(size_t)v * n + i] - ref) / ref);
      if (e > worst) worst = e;
    }
    printf("%-20s max relative error %.3Le  (%.2Lf ulp of double)\n", nm[v], worst, worst / 1.1102230246251565e-16L);
  }
  return 0;
}
