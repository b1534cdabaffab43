// Can v_mfma_f64_4x4x4_4b_f64 (four independent 4x4x4 blocks = the four 16-lane env groups of a wave) do k_step's lane reductions?
// (1) operand / result lane layout, found empirically; (2) what it does under a partial EXEC mask; (3) cost next to the DPP butterfly.
// Build + run: hipcc -O3 --offload-arch=gfx950 tools/mfma_red_probe.hip -o tools/_build/mfma_red_probe && tools/_build/mfma_red_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#define REP 256
#define X4(s) s s s s
#define X8(s) X4(s) X4(s)
#define X16(s) X8(s) X8(s)

__device__ __forceinline__ double mfma4(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }

template <int CTRL> __device__ __forceinline__ double dpp_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double gsum_dpp(double v) {
#pragma clang fp contract(off)
  v += dpp_f64<0x140>(v); v += dpp_f64<0x141>(v); v += dpp_f64<0xB1>(v); v += dpp_f64<0x4E>(v);
  return v;
}

__global__ void probe(double* out, long long* cyc) {
  const int l = threadIdx.x;
  // ---- (1) layout: A = 2^(lane % 16) (exact), B = 1: which lanes does each result sum?  Then roles swapped.
  const double p2 = (double)(1u << (l & 15));
  out[l] = mfma4(p2, 1.0, 0.0);            // D = A x ones
  out[64 + l] = mfma4(1.0, p2, 0.0);       // D = ones x B
  // two-stage all-reduce candidates
  const double x = 1.0 + l;                // group g sum: 16 + sum(16g .. 16g+15)
  const double s1 = mfma4(x, 1.0, 0.0);
  out[128 + l] = mfma4(1.0, s1, 0.0);      // candidate A: ones x (A x ones)
  out[192 + l] = mfma4(s1, 1.0, 0.0);      // candidate B: (A x ones) x ones
  const double s2 = mfma4(1.0, x, 0.0);
  out[256 + l] = mfma4(s2, 1.0, 0.0);      // candidate C
  out[320 + l] = mfma4(1.0, s2, 0.0);      // candidate D
  // ---- (2) partial EXEC: only group 1 (lanes 16..31) runs the MFMA; the others keep a sentinel in the destination register
  double d = -7.0;
  if ((l >> 4) == 1) asm volatile("s_nop 4\n\tv_mfma_f64_4x4x4_4b_f64 %0, %1, %2, 0\n\ts_nop 7\n\ts_nop 7" : "+v"(d) : "v"(x), "v"(1.0));
  out[384 + l] = d;
  // ---- (3) cost: one wave on its SIMD
  long long t0, t1; int k = 0;
  double r0 = x, r1 = x + 1, r2 = x + 2, r3 = x + 3;
#define BEGIN t0 = clock64();
#define END t1 = clock64(); if (l == 0) cyc[k] = t1 - t0; k++;
  BEGIN for (int i = 0; i < REP; i++) { r0 = gsum_dpp(r0) * 0.0625; asm volatile("" : "+v"(r0)); } END                                  // 0: DPP butterfly, one value, dependent
  BEGIN for (int i = 0; i < REP; i++) { r0 = gsum_dpp(r0) * 0.0625; r1 = gsum_dpp(r1) * 0.0625; r2 = gsum_dpp(r2) * 0.0625; r3 = gsum_dpp(r3) * 0.0625;
    asm volatile("" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3)); } END                                                                     // 1: four values
  BEGIN for (int i = 0; i < REP; i++) { r0 = mfma4(1.0, mfma4(r0, 1.0, 0.0), 0.0) * 0.0625; asm volatile("" : "+v"(r0)); } END            // 2: MFMA pair, one value, dependent
  BEGIN for (int i = 0; i < REP; i++) { double a0 = mfma4(r0, 1.0, 0.0), a1 = mfma4(r1, 1.0, 0.0), a2 = mfma4(r2, 1.0, 0.0), a3 = mfma4(r3, 1.0, 0.0);
    r0 = mfma4(1.0, a0, 0.0) * 0.0625; r1 = mfma4(1.0, a1, 0.0) * 0.0625; r2 = mfma4(1.0, a2, 0.0) * 0.0625; r3 = mfma4(1.0, a3, 0.0) * 0.0625;
    asm volatile("" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3)); } END                                                                     // 3: four values
  // 4: back-to-back independent MFMAs (issue rate)
  { double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    BEGIN for (int i = 0; i < REP; i++) asm volatile(X4("v_mfma_f64_4x4x4_4b_f64 %0, %4, %5, %0\n\tv_mfma_f64_4x4x4_4b_f64 %1, %4, %5, %1\n\tv_mfma_f64_4x4x4_4b_f64 %2, %4, %5, %2\n\tv_mfma_f64_4x4x4_4b_f64 %3, %4, %5, %3\n\t")
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(1.0)); END
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory"); r1 += a0 + a1 + a2 + a3; }
  // 5: 4 independent MFMAs + 16 independent v_fma_f64 interleaved (do the VALU ops hide behind the matrix pipe?)
  { double a0 = 0, a1 = 0, a2 = 0, a3 = 0, f0 = x, f1 = x, f2 = x, f3 = x;
    BEGIN for (int i = 0; i < REP; i++) asm volatile(X4("v_mfma_f64_4x4x4_4b_f64 %0, %8, %9, %0\n\tv_fma_f64 %4, %4, %9, %9\n\tv_fma_f64 %5, %5, %9, %9\n\tv_fma_f64 %6, %6, %9, %9\n\tv_fma_f64 %7, %7, %9, %9\n\t"
        "v_mfma_f64_4x4x4_4b_f64 %1, %8, %9, %1\n\tv_fma_f64 %4, %4, %9, %9\n\tv_fma_f64 %5, %5, %9, %9\n\tv_fma_f64 %6, %6, %9, %9\n\tv_fma_f64 %7, %7, %9, %9\n\t"
        "v_mfma_f64_4x4x4_4b_f64 %2, %8, %9, %2\n\tv_fma_f64 %4, %4, %9, %9\n\tv_fma_f64 %5, %5, %9, %9\n\tv_fma_f64 %6, %6, %9, %9\n\tv_fma_f64 %7, %7, %9, %9\n\t"
        "v_mfma_f64_4x4x4_4b_f64 %3, %8, %9, %3\n\tv_fma_f64 %4, %4, %9, %9\n\tv_fma_f64 %5, %5, %9, %9\n\tv_fma_f64 %6, %6, %9, %9\n\tv_fma_f64 %7, %7, %9, %9\n\t")
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(x), "v"(1.0)); END
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory"); r2 += a0 + a1 + a2 + a3 + f0 + f1 + f2 + f3; }
  // 6: the same 16 x 4 v_fma_f64 alone
  { double f0 = x, f1 = x, f2 = x, f3 = x;
    BEGIN for (int i = 0; i < REP; i++) asm volatile(X16("v_fma_f64 %0, %0, %4, %4\n\tv_fma_f64 %1, %1, %4, %4\n\tv_fma_f64 %2, %2, %4, %4\n\tv_fma_f64 %3, %3, %4, %4\n\t")
        : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(1.0)); END
    r3 += f0 + f1 + f2 + f3; }
  // 7: dependent MFMA chain (D of one is A of the next): latency
  { double a0 = x;
    BEGIN for (int i = 0; i < REP; i++) { a0 = mfma4(a0, 0.25, 0.0); a0 = mfma4(a0, 0.25, 0.0); a0 = mfma4(a0, 0.25, 0.0); a0 = mfma4(a0, 0.25, 0.0); asm volatile("" : "+v"(a0)); } END
    r0 += a0; }
  out[448 + l] = r0 + r1 + r2 + r3;
  if (l == 0) cyc[63] = k;
}

int main() {
  double* out; long long* cyc;
  CHECK(hipMalloc(&out, 512 * 8)); CHECK(hipMalloc(&cyc, 64 * 8));
  for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, out, cyc); CHECK(hipDeviceSynchronize()); }
  double h[512]; long long hc[64];
  CHECK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost)); CHECK(hipMemcpy(hc, cyc, sizeof(hc), hipMemcpyDeviceToHost));
  auto bits = [](double v) { static char s[64]; unsigned m = (unsigned)v; int n = 0; for (int i = 0; i < 16; i++) if (m >> i & 1) n += sprintf(s + n, "%d ", i); s[n] = 0; return s; };
  printf("# (1) D = A x ones with A = 2^(lane%%16): lanes of the group summed into each result lane (group 0 shown; all groups alike: %s)\n",
         [&] { for (int l = 16; l < 64; l++) if (h[l] != h[l & 15]) return "NO"; return "yes"; }());
  for (int l = 0; l < 16; l++) printf("  lane %2d <- { %s}\n", l, bits(h[l]));
  printf("# D = ones x B with B = 2^(lane%%16)\n");
  for (int l = 0; l < 16; l++) printf("  lane %2d <- { %s}\n", l, bits(h[64 + l]));
  const char* cand[] = {"ones x (x x ones)", "(x x ones) x ones", "(ones x x) x ones", "ones x (ones x x)"};
  for (int c = 0; c < 4; c++) {
    int ok = 1;
    for (int l = 0; l < 64; l++) { int g = l >> 4; double want = 16.0 + (16 * g) * 16 + 120; if (h[128 + 64 * c + l] != want) ok = 0; }
    printf("# two-stage candidate %-22s: %s (lane 0: %.1f, lane 17: %.1f, lane 63: %.1f; want 136 / 392 / 904)\n", cand[c], ok ? "ALL-REDUCE over each 16-lane group" : "no",
           h[128 + 64 * c], h[128 + 64 * c + 17], h[128 + 64 * c + 63]);
  }
  printf("# (2) MFMA under EXEC = lanes 16..31 only; destination preset to -7 everywhere:\n  ");
  for (int l = 0; l < 64; l += 5) printf("lane %d: %.1f  ", l, h[384 + l]);
  printf("\n");
  const char* name[] = {"DPP butterfly gsum, 1 value (+mul)", "DPP butterfly gsum, 4 values (+4 mul)", "MFMA pair, 1 value (+mul)", "MFMA pairs, 4 values (+4 mul)",
                        "16 independent MFMAs (per MFMA)", "4 MFMA + 16 v_fma_f64 interleaved (per group of 20)", "16 v_fma_f64 x 4 (per group of 16)", "dependent MFMA chain (per MFMA)"};
  const double per[] = {1, 1, 1, 1, 16, 4, 4, 4};
  printf("# (3) s_memtime ticks per unit\n");
  for (int i = 0; i < (int)hc[63]; i++) printf("%-56s %8.2f\n", name[i], (double)hc[i] / (REP * per[i]));
  return 0;
}
