// Instruction issue / dependent-latency probe for one wave per SIMD on gfx950 (what k_step's phases are made of).
// One workgroup of 64 lanes; every test runs REP x 32 instructions between two s_memtime reads; cycles per instruction printed.
// Build + run: hipcc -O3 --offload-arch=gfx950 tools/lat_probe.hip -o tools/_build/lat_probe && tools/_build/lat_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP 256
#define X4(s) s s s s
#define X8(s) X4(s) X4(s)
#define X32(s) X8(s) X8(s) X8(s) X8(s)
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void probe(double* out, long long* cyc, const double* in) {
  __shared__ double lds[1024];
  const int l = threadIdx.x;
  for (int i = l; i < 1024; i += 64) lds[i] = in[i & 63];
  __syncthreads();
  double a = in[l], b = in[l + 64] * 1e-9, c = 0.999999;
  double r0 = a, r1 = a + 1, r2 = a + 2, r3 = a + 3, r4 = a + 4, r5 = a + 5, r6 = a + 6, r7 = a + 7;
  long long t0, t1;
  int k = 0;
#define BEGIN t0 = clock64();
#define END t1 = clock64(); if (l == 0) cyc[k] = t1 - t0; k++;
  // 0: dependent v_fma_f64
  BEGIN for (int i = 0; i < REP; i++) asm volatile(X32("v_fma_f64 %0, %0, %1, %2\n\t") : "+v"(r0) : "v"(c), "v"(b)); END
  // 1: 8 independent v_fma_f64
  BEGIN for (int i = 0; i < REP; i++) asm volatile(X4("v_fma_f64 %0, %0, %8, %9\n\tv_fma_f64 %1, %1, %8, %9\n\tv_fma_f64 %2, %2, %8, %9\n\tv_fma_f64 %3, %3, %8, %9\n\t"
      "v_fma_f64 %4, %4, %8, %9\n\tv_fma_f64 %5, %5, %8, %9\n\tv_fma_f64 %6, %6, %8, %9\n\tv_fma_f64 %7, %7, %8, %9\n\t")
      : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(c), "v"(b)); END
  // 2: dependent v_mul_f64
  BEGIN for (int i = 0; i < REP; i++) asm volatile(X32("v_mul_f64 %0, %0, %1\n\t") : "+v"(r0) : "v"(c)); END
  // 3: dependent v_add_f64
  BEGIN for (int i = 0; i < REP; i++) asm volatile(X32("v_add_f64 %0, %0, %1\n\t") : "+v"(r0) : "v"(b)); END
  // 4: dependent v_fmac_f64_dpp (accumulator chain; DPP source constant)
  BEGIN for (int i = 0; i < REP; i++) asm volatile(X32("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t") : "+v"(r0) : "v"(b), "v"(c)); END
  // 5: 8 independent v_fmac_f64_dpp
  BEGIN for (int i = 0; i < REP; i++) asm volatile(X4("v_fmac_f64_dpp %0, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %2, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %4, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %5, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %6, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %7, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t")
      : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(b), "v"(c)); END
  // 6: dependent v_rsq_f64 (x -> rsq(x), stays near 1)
  r0 = 1.0 + 1e-3 * l;
  BEGIN for (int i = 0; i < REP; i++) asm volatile(X32("v_rsq_f64 %0, %0\n\t") : "+v"(r0)); END
  // 7: 8 independent v_rsq_f64
  BEGIN for (int i = 0; i < REP; i++) asm volatile(X4("v_rsq_f64 %0, %0\n\tv_rsq_f64 %1, %1\n\tv_rsq_f64 %2, %2\n\tv_rsq_f64 %3, %3\n\tv_rsq_f64 %4, %4\n\tv_rsq_f64 %5, %5\n\tv_rsq_f64 %6, %6\n\tv_rsq_f64 %7, %7\n\t")
      : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)); END
  // 8: dependent fma alternating with an independent dpp fmac (does the fmac fill the fma's latency?)
  BEGIN for (int i = 0; i < REP; i++) asm volatile(X8("v_fma_f64 %0, %0, %3, %2\n\tv_fmac_f64_dpp %1, %2, %3 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
      "v_fma_f64 %0, %0, %3, %2\n\tv_fmac_f64_dpp %4, %2, %3 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t") : "+v"(r0), "+v"(r1) : "v"(b), "v"(c), "v"(r2)); END
  // 9: v_mov_b64_dpp feeding a dependent add (the broadcast-then-use pattern: s_nop 1 for the DPP read hazard)
  BEGIN for (int i = 0; i < REP; i++) asm volatile(X32("s_nop 1\n\tv_mov_b64_dpp %1, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_add_f64 %0, %1, %2\n\t") : "+v"(r0), "+v"(r1) : "v"(b)); END
  // 10: 64-bit value moved by two v_mov_b32_dpp (row_shr:1) feeding a dependent add (the lane-reduction pattern)
  { int lo = l, hi = l + 1, tl = 0, th = 0;
  BEGIN for (int i = 0; i < REP; i++) asm volatile(X32("s_nop 1\n\tv_mov_b32_dpp %2, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_u32 %0, %2, %3\n\t")
      : "+v"(lo), "+v"(hi), "+v"(tl), "+v"(th)); END r1 += lo + hi; }
  // 11: dependent v_cndmask_b32 pair
  { int lo = l, hi = l + 1;
  BEGIN for (int i = 0; i < REP; i++) asm volatile(X32("v_cndmask_b32 %0, %0, %1, vcc\n\tv_cndmask_b32 %1, %1, %0, vcc\n\t") : "+v"(lo), "+v"(hi) : : "vcc"); END r1 += lo + hi; }
  // 12: LDS round trip: ds_read_b64 whose address depends on the previous value (pointer chase through zeros)
  {
    int addr = (l & 15) * 8;
    double v = 0;
    BEGIN for (int i = 0; i < REP; i++) asm volatile(X32("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)\n\t") : "+v"(v) : "v"(addr)); END
    r1 += v;
  }
  // 13: LDS: 8 independent ds_read_b128 then one wait (a batch)
  {
    int addr = (l & 15) * 16;
    double q0, q1, q2, q3, q4, q5, q6, q7, q8, q9, qa, qb, qc, qd, qe, qf;
    BEGIN for (int i = 0; i < REP * 4; i++) asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:256\n\tds_read_b128 %2, %8 offset:512\n\tds_read_b128 %3, %8 offset:768\n\t"
        "ds_read_b128 %4, %8 offset:1024\n\tds_read_b128 %5, %8 offset:1280\n\tds_read_b128 %6, %8 offset:1536\n\tds_read_b128 %7, %8 offset:1792\n\ts_waitcnt lgkmcnt(0)\n\t"
        : "=v"(*(double2*)&q0), "=v"(*(double2*)&q2), "=v"(*(double2*)&q4), "=v"(*(double2*)&q6), "=v"(*(double2*)&q8), "=v"(*(double2*)&qa), "=v"(*(double2*)&qc), "=v"(*(double2*)&qe) : "v"(addr)); END
    r2 += q0 + q2 + q4 + q6 + q8 + qa + qc + qe + q1 + q3 + q5 + q7 + q9 + qb + qd + qf;
  }
  // 14: LDS write -> read round trip (ds_write_b64, wait, ds_read_b64, wait)
  {
    int addr = l * 8;
    double v = r0;
    BEGIN for (int i = 0; i < REP; i++) asm volatile(X32("ds_write_b64 %1, %0\n\ts_waitcnt lgkmcnt(0)\n\tds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)\n\t") : "+v"(v) : "v"(addr)); END
    r3 += v;
  }
  // 15: global load round trip (same 64-byte line per lane group, L2 / TCP hit), dependent address
  {
    const double* p = in;
    double v = 0;
    BEGIN for (int i = 0; i < REP; i++) asm volatile(X8("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)\n\t") : "+v"(v) : "v"(p)); END
    r4 += v;
  }
  // 16: scalar load round trip
  {
    const double* p = in;
    double v;
    BEGIN for (int i = 0; i < REP; i++) asm volatile(X8("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)\n\t") : "=s"(v) : "s"(p)); END
    r5 += v;
  }
  // 17: ds_bpermute_b32 pair round trip (a 64-bit __shfl)
  {
    int idx = ((l + 1) & 63) * 4;
    BEGIN { int lo = l, hi = l + 1; for (int i = 0; i < REP; i++) asm volatile(X32("ds_bpermute_b32 %0, %2, %0\n\tds_bpermute_b32 %1, %2, %1\n\ts_waitcnt lgkmcnt(0)\n\t") : "+v"(lo), "+v"(hi) : "v"(idx)); r1 += lo + hi; } END
  }
  // 18: v_readlane + use as scalar operand in a dependent fma
  {
    int s;
    BEGIN { int lo = l; for (int i = 0; i < REP; i++) asm volatile(X32("v_readlane_b32 %1, %0, 3\n\tv_add_u32 %0, %1, %0\n\t") : "+v"(lo), "=s"(s)); r1 += lo; } END
  }
  // 19: v_sqrt_f64 independent x8 ; 20: v_rcp_f64 dependent
  BEGIN for (int i = 0; i < REP; i++) asm volatile(X4("v_sqrt_f64 %0, %0\n\tv_sqrt_f64 %1, %1\n\tv_sqrt_f64 %2, %2\n\tv_sqrt_f64 %3, %3\n\tv_sqrt_f64 %4, %4\n\tv_sqrt_f64 %5, %5\n\tv_sqrt_f64 %6, %6\n\tv_sqrt_f64 %7, %7\n\t")
      : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)); END
  r0 = 1.0 + 1e-3 * l;
  BEGIN for (int i = 0; i < REP; i++) asm volatile(X32("v_rcp_f64 %0, %0\n\t") : "+v"(r0)); END
  // 21: dependent fma chain with ONE independent fma between links (ILP 2), 22: ILP 4
  BEGIN for (int i = 0; i < REP; i++) asm volatile(X8("v_fma_f64 %0, %0, %2, %3\n\tv_fma_f64 %1, %1, %2, %3\n\tv_fma_f64 %0, %0, %2, %3\n\tv_fma_f64 %1, %1, %2, %3\n\t") : "+v"(r0), "+v"(r1) : "v"(c), "v"(b)); END
  BEGIN for (int i = 0; i < REP; i++) asm volatile(X8("v_fma_f64 %0, %0, %4, %5\n\tv_fma_f64 %1, %1, %4, %5\n\tv_fma_f64 %2, %2, %4, %5\n\tv_fma_f64 %3, %3, %4, %5\n\t") : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(c), "v"(b)); END
  // 23: v_fmac_f64_dpp chain ILP 2
  BEGIN for (int i = 0; i < REP; i++) asm volatile(X8("v_fmac_f64_dpp %0, %2, %3 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %2, %3 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %2, %3 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %2, %3 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t") : "+v"(r0), "+v"(r1) : "v"(b), "v"(c)); END
  // 24: dependent v_mul_f32 (reference: 32-bit VALU)
  { float f = (float)a; BEGIN for (int i = 0; i < REP; i++) asm volatile(X32("v_mul_f32 %0, %0, %1\n\t") : "+v"(f) : "v"(0.9999f)); END r7 += f; }
  out[l] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
  if (l == 0) cyc[63] = k;
}

int main() {
  double *in, *out; long long* cyc;
  CHECK(hipMalloc(&in, 4096 * 8)); CHECK(hipMalloc(&out, 64 * 8)); CHECK(hipMalloc(&cyc, 64 * 8));
  double h[4096]; for (int i = 0; i < 4096; i++) h[i] = 0.0; for (int i = 0; i < 128; i++) h[i] = 1.0 + i * 1e-3;
  CHECK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
  const char* name[] = {"v_fma_f64 dependent", "v_fma_f64 x8 independent", "v_mul_f64 dependent", "v_add_f64 dependent", "v_fmac_f64_dpp dependent (acc)",
    "v_fmac_f64_dpp x8 independent", "v_rsq_f64 dependent", "v_rsq_f64 x8 independent", "dep fma + indep dpp fmac (pair)", "s_nop1 + v_mov_b64_dpp + dep add (triple)",
    "s_nop1 + 2 v_mov_b32_dpp + dep add (quad)", "v_cndmask_b32 x2 dependent (pair)", "LDS read round trip (ds_read_b64 + wait + and)", "LDS 8 x ds_read_b128 + one wait (per batch)",
    "LDS write+wait+read+wait", "global_load round trip (L2/TCP hit) + 2 valu", "s_load round trip", "ds_bpermute x2 + wait", "v_readlane + dependent valu (pair)",
    "v_sqrt_f64 x8 independent", "v_rcp_f64 dependent", "v_fma_f64 ILP2", "v_fma_f64 ILP4", "v_fmac_f64_dpp ILP2", "v_mul_f32 dependent"};
  const double per[] = {32, 32, 32, 32, 32, 32, 32, 32, 16, 32, 32, 32, 32, 0.25, 32, 8, 8, 32, 32, 32, 32, 32, 32, 32, 32};
  for (int rep = 0; rep < 2; rep++) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, out, cyc, in);
    CHECK(hipDeviceSynchronize());
  }
  long long hc[64]; CHECK(hipMemcpy(hc, cyc, sizeof(hc), hipMemcpyDeviceToHost));
  printf("# s_memtime ticks per unit (one wave on its SIMD, gfx950); the unit is one instruction unless the name says pair / triple / batch\n");
  for (int i = 0; i < (int)hc[63]; i++) printf("%-52s %8.2f\n", name[i], (double)hc[i] / (REP * per[i]));
  return 0;
}
