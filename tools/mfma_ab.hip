// A/B microbenchmark (VERDICT r1 #8 / #4e): the Newton Hessian accumulate H = M + sum_r d_r J_r J_r^T for nv = 16,
// R active rows, four envs per wave in the product's layout (env e = lanes 16e..16e+15, lane j holds entry j of every row
// and, after the build, column j of H -- what the Cholesky that follows consumes).
//   A: what k_step does -- per row one scale and 16 v_fmac_f64_dpp (row_newbcast), all four envs per instruction.
//   B: v_mfma_f64_16x16x4_f64 -- one env per instruction, K = 4 rows per issue; operands staged through LDS into the
//      MFMA layout (lane = 16 k + i), the 16x16 result brought back to the column-per-lane layout through LDS.
// Both variants are checked against a host computation before they are timed.  One wave per workgroup, 1024 workgroups
// (one wave per SIMD, as k_step runs), ITER dependent builds per wave.
// Build + run (GPU box):  hipcc -O3 --offload-arch=gfx950 -I include -o /tmp/mfma_ab tools/mfma_ab.hip && /tmp/mfma_ab
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../gym_kmanip_amd/csrc/kmanip_device.hpp"

constexpr int NV = 16, R = 24, EPW = 4, ITER = 256;
typedef double v4d __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// inputs: J[env][r][j], d[env][r], M[env][i][j] (symmetric); output H[env][i][j]
__global__ __launch_bounds__(64) void build_dpp(const double* Jg, const double* dg, const double* Mg, double* Hg, int iters) {
  const int lane = threadIdx.x, sub = lane & 15, env = blockIdx.x * EPW + lane / 16;
  double J[R], d[R], Mc[NV], H[NV];
#pragma unroll
  for (int r = 0; r < R; r++) { J[r] = Jg[((size_t)env * R + r) * NV + sub]; d[r] = dg[(size_t)env * R + r]; }
#pragma unroll
  for (int i = 0; i < NV; i++) Mc[i] = Mg[((size_t)env * NV + i) * NV + sub];
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NV; i++) H[i] = Mc[i];
#pragma unroll
    for (int q = 0; q < R / 4; q++) {
      const double t0 = d[4 * q] * J[4 * q], t1 = d[4 * q + 1] * J[4 * q + 1], t2 = d[4 * q + 2] * J[4 * q + 2], t3 = d[4 * q + 3] * J[4 * q + 3];
      static_for<0, NV>([&](auto I) {
        constexpr int i = decltype(I)::value;
        dppfma_acc4<i>(H[i], J[4 * q], t0, J[4 * q + 1], t1, J[4 * q + 2], t2, J[4 * q + 3], t3);
      });
    }
    J[0] = fma(H[it & 15], 1e-300, J[0]);          // loop-carried dependency: the builds cannot overlap or be hoisted
  }
#pragma unroll
  for (int i = 0; i < NV; i++) Hg[((size_t)env * NV + i) * NV + sub] = H[i];
}

template <int LAYOUT>
__global__ __launch_bounds__(64) void build_mfma(const double* Jg, const double* dg, const double* Mg, double* Hg, int iters) {
  __shared__ double sJ[EPW][R][NV], sT[EPW][R][NV], sH[EPW][NV][NV];
  const int lane = threadIdx.x, sub = lane & 15, e_own = lane / 16, env = blockIdx.x * EPW + e_own;
  double J[R], d[R], Mc[NV], H[NV];
#pragma unroll
  for (int r = 0; r < R; r++) { J[r] = Jg[((size_t)env * R + r) * NV + sub]; d[r] = dg[(size_t)env * R + r]; }
#pragma unroll
  for (int i = 0; i < NV; i++) Mc[i] = Mg[((size_t)env * NV + i) * NV + sub];
  for (int it = 0; it < iters; it++) {
    // the rows live in registers in the product (built there): stage them and their scaled copies
#pragma unroll
    for (int r = 0; r < R; r++) { sJ[e_own][r][sub] = J[r]; sT[e_own][r][sub] = d[r] * J[r]; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int e = 0; e < EPW; e++) {
      v4d acc = {0, 0, 0, 0};
#pragma unroll
      for (int q = 0; q < R / 4; q++) {
        const double a = sT[e][4 * q + lane / 16][sub], b = sJ[e][4 * q + lane / 16][sub];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; r++) sH[e][LAYOUT == 0 ? 4 * (lane / 16) + r : 4 * r + lane / 16][sub] = acc[r];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < NV; i++) H[i] = Mc[i] + sH[e_own][i][sub];
    J[0] = fma(H[it & 15], 1e-300, J[0]);
    __builtin_amdgcn_wave_barrier();
  }
#pragma unroll
  for (int i = 0; i < NV; i++) Hg[((size_t)env * NV + i) * NV + sub] = H[i];
}

// ---------------------------------------------------------------------------------------------------------------------
// Second A/B: the one contraction the north star names for MFMA -- the IK's J^T J.  ik_jac's dense part is
// 6 x n (n = 7; the two regulariser blocks are 9e-3 I and add a constant to the diagonal), one problem per 16-lane DPP row,
// lane c < 8 of the row holds COLUMN c of J (6 doubles) and wants ROW c of A = J^T J (7 doubles) -- what the cooperative
// trust-region solve consumes (kmanip_ik_coop.hpp).
//   A: what the product does -- per entry six v_fmac_f64_dpp (row_newbcast:j folded into the FMA), four problems per instruction.
//   B: v_mfma_f64_16x16x4_f64 -- two problems per 16 x 16 tile (columns 0-6 and 8-14; the off-diagonal blocks are waste), K = 6
//      padded to 8 = two issues per tile, two tiles per wave; operands to the MFMA layout (lane = 16 k + i) and the result back to
//      row-per-lane through LDS.
constexpr int JN = 7, JK = 6;
__global__ __launch_bounds__(64) void jtj_dpp(const double* Jg, double* Ag, int iters) {
  const int lane = threadIdx.x, c = lane & 15, prob = blockIdx.x * 4 + lane / 16;
  double Jc[JK], A[JN];
#pragma unroll
  for (int k = 0; k < JK; k++) Jc[k] = c < JN ? Jg[((size_t)prob * JK + k) * 8 + c] : 0.0;
  for (int it = 0; it < iters; it++) {
    static_for<0, JN>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      double s = 0;
      dppfma_acc3<j>(s, Jc[0], Jc[0], Jc[1], Jc[1], Jc[2], Jc[2]);
      dppfma_acc3<j>(s, Jc[3], Jc[3], Jc[4], Jc[4], Jc[5], Jc[5]);
      A[j] = s;
    });
    Jc[0] = fma(A[it % JN], 1e-300, Jc[0]);
  }
#pragma unroll
  for (int j = 0; j < JN; j++) if (c < JN) Ag[((size_t)prob * 8 + c) * 8 + j] = A[j];
}
template <int LAYOUT>
__global__ __launch_bounds__(64) void jtj_mfma(const double* Jg, double* Ag, int iters) {
  __shared__ double sJ[2][8][16];       // [tile][k (padded to 8)][column: problem 2t -> 0..7, problem 2t+1 -> 8..15]
  __shared__ double sA[2][16][16];
  const int lane = threadIdx.x, c = lane & 15, p_own = lane / 16, prob = blockIdx.x * 4 + p_own;
  double Jc[JK], A[JN];
#pragma unroll
  for (int k = 0; k < JK; k++) Jc[k] = c < JN ? Jg[((size_t)prob * JK + k) * 8 + c] : 0.0;
  if (lane < 32) { sJ[lane / 16][6][lane & 15] = 0; sJ[lane / 16][7][lane & 15] = 0; }
  for (int it = 0; it < iters; it++) {
    if (c < 8) {
#pragma unroll
      for (int k = 0; k < JK; k++) sJ[p_own / 2][k][(p_own & 1) * 8 + c] = Jc[k];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int t = 0; t < 2; t++) {
      v4d acc = {0, 0, 0, 0};
#pragma unroll
      for (int q = 0; q < 2; q++) {
        const double a = sJ[t][4 * q + lane / 16][c];          // A operand: (J^T)[i = c][k]; B operand: J[k][j = c] -- the same number
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; r++) sA[t][LAYOUT == 0 ? 4 * (lane / 16) + r : 4 * r + lane / 16][c] = acc[r];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < JN; j++) A[j] = sA[p_own / 2][(p_own & 1) * 8 + (c & 7)][(p_own & 1) * 8 + j];
    Jc[0] = fma(A[it % JN], 1e-300, Jc[0]);
    __builtin_amdgcn_wave_barrier();
  }
#pragma unroll
  for (int j = 0; j < JN; j++) if (c < JN) Ag[((size_t)prob * 8 + c) * 8 + j] = A[j];
}
static int jtj_ab() {
  const int nblk = 1024, nprob = nblk * 4;
  std::vector<double> J((size_t)nprob * JK * 8, 0.0), A((size_t)nprob * 64), Aref((size_t)64 * 64, 0.0);
  srand(2);
  for (int p = 0; p < nprob; p++) for (int k = 0; k < JK; k++) for (int c = 0; c < JN; c++) J[((size_t)p * JK + k) * 8 + c] = rand() / (double)RAND_MAX - 0.5;
  for (int p = 0; p < 64; p++) for (int i = 0; i < JN; i++) for (int j = 0; j < JN; j++) {
    double s = 0;
    for (int k = 0; k < JK; k++) s += J[((size_t)p * JK + k) * 8 + i] * J[((size_t)p * JK + k) * 8 + j];
    Aref[((size_t)p * 8 + i) * 8 + j] = s;
  }
  double *Jd, *Ad;
  CK(hipMalloc(&Jd, J.size() * 8)); CK(hipMalloc(&Ad, A.size() * 8));
  CK(hipMemcpy(Jd, J.data(), J.size() * 8, hipMemcpyHostToDevice));
  auto check = [&](const char* name) {
    CK(hipMemcpy(A.data(), Ad, A.size() * 8, hipMemcpyDeviceToHost));
    double err = 0;
    for (int p = 0; p < 64; p++) for (int i = 0; i < JN; i++) for (int j = 0; j < JN; j++)
      err = fmax(err, fabs(A[((size_t)p * 8 + i) * 8 + j] - Aref[((size_t)p * 8 + i) * 8 + j]));
    fprintf(stderr, "%s: max |A - host| on 64 problems = %.3e\n", name, err);
    return err;
  };
  CK(hipMemset(Ad, 0, A.size() * 8));
  hipLaunchKernelGGL(jtj_dpp, dim3(nblk), dim3(64), 0, 0, Jd, Ad, 1); CK(hipDeviceSynchronize());
  const double errA = check("jtj dpp");
  hipLaunchKernelGGL(jtj_mfma<0>, dim3(nblk), dim3(64), 0, 0, Jd, Ad, 1); CK(hipDeviceSynchronize());
  const double e0 = check("jtj mfma layout 0");
  hipLaunchKernelGGL(jtj_mfma<1>, dim3(nblk), dim3(64), 0, 0, Jd, Ad, 1); CK(hipDeviceSynchronize());
  const double e1 = check("jtj mfma layout 1");
  const int layout = e0 <= e1 ? 0 : 1;
  hipEvent_t ev0, ev1; CK(hipEventCreate(&ev0)); CK(hipEventCreate(&ev1));
  auto timeit = [&](int which) {
    float best = 1e30f;
    for (int rep = 0; rep < 5; rep++) {
      CK(hipEventRecord(ev0));
      if (which == 0) hipLaunchKernelGGL(jtj_dpp, dim3(nblk), dim3(64), 0, 0, Jd, Ad, ITER);
      else if (layout == 0) hipLaunchKernelGGL(jtj_mfma<0>, dim3(nblk), dim3(64), 0, 0, Jd, Ad, ITER);
      else hipLaunchKernelGGL(jtj_mfma<1>, dim3(nblk), dim3(64), 0, 0, Jd, Ad, ITER);
      CK(hipEventRecord(ev1)); CK(hipEventSynchronize(ev1));
      float ms; CK(hipEventElapsedTime(&ms, ev0, ev1)); best = fminf(best, ms);
    }
    return best;
  };
  const float msA = timeit(0), msB = timeit(1);
  printf("{\"what\": \"IK normal matrix A = J^T J, J 6 x 7 (ik_jac's dense rows), one problem per DPP row (4 per wave), 1024 waves (1 per SIMD), %d dependent builds per wave\", "
         "\"dpp_valu\": {\"ns_per_build\": %.1f, \"max_abs_err\": %.2e}, \"mfma_f64_16x16x4\": {\"ns_per_build\": %.1f, \"max_abs_err\": %.2e, \"tiles_per_wave\": 2, \"issues_per_tile\": 2}, "
         "\"mfma_over_dpp\": %.3f}\n", ITER, 1e6 * msA / ITER, errA, 1e6 * msB / ITER, fmin(e0, e1), msB / msA);
  return (errA < 1e-12 && fmin(e0, e1) < 1e-12) ? 0 : 2;
}

int main() {
  const int nblk = 1024, nenv = nblk * EPW;
  std::vector<double> J((size_t)nenv * R * NV), d((size_t)nenv * R), M((size_t)nenv * NV * NV), Href((size_t)nenv * NV * NV), H((size_t)nenv * NV * NV);
  srand(1);
  for (auto& x : J) x = rand() / (double)RAND_MAX - 0.5;
  for (auto& x : d) x = 0.5 + rand() / (double)RAND_MAX;
  for (int e = 0; e < nenv; e++)
    for (int i = 0; i < NV; i++) for (int j = 0; j <= i; j++) { double v = (i == j) ? 2.0 : 0.01 * (i + j); M[((size_t)e * NV + i) * NV + j] = v; M[((size_t)e * NV + j) * NV + i] = v; }
  for (int e = 0; e < 64; e++)                                  // host reference on the first 64 envs
    for (int i = 0; i < NV; i++) for (int j = 0; j < NV; j++) {
      double s = M[((size_t)e * NV + i) * NV + j];
      for (int r = 0; r < R; r++) s += d[(size_t)e * R + r] * J[((size_t)e * R + r) * NV + i] * J[((size_t)e * R + r) * NV + j];
      Href[((size_t)e * NV + i) * NV + j] = s;
    }
  double *Jd, *dd, *Md, *Hd;
  CK(hipMalloc(&Jd, J.size() * 8)); CK(hipMalloc(&dd, d.size() * 8)); CK(hipMalloc(&Md, M.size() * 8)); CK(hipMalloc(&Hd, H.size() * 8));
  CK(hipMemcpy(Jd, J.data(), J.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dd, d.data(), d.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(Md, M.data(), M.size() * 8, hipMemcpyHostToDevice));
  auto check = [&](const char* name) {
    CK(hipMemcpy(H.data(), Hd, H.size() * 8, hipMemcpyDeviceToHost));
    double err = 0;
    for (size_t k = 0; k < (size_t)64 * NV * NV; k++) err = fmax(err, fabs(H[k] - Href[k]));
    fprintf(stderr, "%s: max |H - host| on 64 envs = %.3e\n", name, err);
    return err;
  };
  hipLaunchKernelGGL(build_dpp, dim3(nblk), dim3(64), 0, 0, Jd, dd, Md, Hd, 1); CK(hipDeviceSynchronize());
  const double errA = check("dpp");
  hipLaunchKernelGGL(build_mfma<0>, dim3(nblk), dim3(64), 0, 0, Jd, dd, Md, Hd, 1); CK(hipDeviceSynchronize());
  const double errB0 = check("mfma layout row = 4*(lane/16)+r");
  hipLaunchKernelGGL(build_mfma<1>, dim3(nblk), dim3(64), 0, 0, Jd, dd, Md, Hd, 1); CK(hipDeviceSynchronize());
  const double errB1 = check("mfma layout row = 4*r+lane/16");
  const int layout = errB0 <= errB1 ? 0 : 1;
  const double errB = layout == 0 ? errB0 : errB1;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](int which) {
    float best = 1e30f;
    for (int rep = 0; rep < 5; rep++) {
      CK(hipEventRecord(e0));
      if (which == 0) hipLaunchKernelGGL(build_dpp, dim3(nblk), dim3(64), 0, 0, Jd, dd, Md, Hd, ITER);
      else if (layout == 0) hipLaunchKernelGGL(build_mfma<0>, dim3(nblk), dim3(64), 0, 0, Jd, dd, Md, Hd, ITER);
      else hipLaunchKernelGGL(build_mfma<1>, dim3(nblk), dim3(64), 0, 0, Jd, dd, Md, Hd, ITER);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = fminf(best, ms);
    }
    return best;
  };
  const float msA = timeit(0), msB = timeit(1);
  // every wave does ITER dependent builds; all 1024 waves run concurrently (one per SIMD), so ms / ITER = one build's latency
  printf("{\"what\": \"H = M + sum_r d_r J_r J_r^T, nv 16, %d rows, 4 envs per wave, 1024 waves (1 per SIMD), %d dependent builds per wave\", "
         "\"dpp_valu\": {\"ns_per_build\": %.1f, \"max_abs_err\": %.2e}, \"mfma_f64_16x16x4\": {\"ns_per_build\": %.1f, \"max_abs_err\": %.2e, \"d_layout\": \"%s\"}, "
         "\"mfma_over_dpp\": %.3f}\n",
         R, ITER, 1e6 * msA / ITER, errA, 1e6 * msB / ITER, errB, layout == 0 ? "row = 4*(lane/16)+r" : "row = 4*r+lane/16", msB / msA);
  const int rc2 = jtj_ab();
  return (errA < 1e-9 && errB < 1e-9) ? rc2 : 2;
}
