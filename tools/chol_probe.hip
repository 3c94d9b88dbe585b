// chol_probe.hip -- development probe (not part of the product): times the phases of chol_panel with s_memtime stamps
// and the three Cholesky kernels with HIP events on a random SPD matrix.  Build: see tools/run_chol_probe.sh
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#define PPO_CHOL_STAMPS 1
__device__ long long g_stamps[48];
#define STAMP(i) do { if (blockIdx.x == 1 && threadIdx.x == 0) g_stamps[i] = wall_clock64(); } while (0)
#define STAMPB(i) do { if (blockIdx.x + 16 == gridDim.x && threadIdx.x == 0) g_stamps[i] = wall_clock64(); } while (0)  // block 1 of the backward chain (the head is the first workgroup)
#define STAMP1(i) do { if (blockIdx.x == 1 && threadIdx.x == 64) g_stamps[i] = wall_clock64(); } while (0)
#include "../point-plane-object-slam_amd/csrc/ppo_chol.hip"

__global__ void rsq_acc(double* out) {
  const double d = 1.0 + 1e-3 * (threadIdx.x + 256.0 * blockIdx.x) * (1.0 + 0.37 * threadIdx.x);
  const double h = 0.5 * d;
  double y = __builtin_amdgcn_rsq(d);
  {
    const double e1_ = __builtin_fma(-(d * y), y, 1.0);
    const double y3 = __builtin_fma(y, e1_ * __builtin_fma(0.375, e1_, 0.5), y);
    atomicMax((unsigned long long*)&out[3], (unsigned long long)__double_as_longlong(fabs(y3 * sqrt(d) - 1.0)));
  }
  const double e0 = fabs(y * sqrt(d) - 1.0);
  y = y * __builtin_fma(-h, y * y, 1.5);
  const double e1 = fabs(y * sqrt(d) - 1.0);
  y = y * __builtin_fma(-h, y * y, 1.5);
  const double e2 = fabs(y * sqrt(d) - 1.0);
  atomicMax((unsigned long long*)&out[0], (unsigned long long)__double_as_longlong(e0));
  atomicMax((unsigned long long*)&out[1], (unsigned long long)__double_as_longlong(e1));
  atomicMax((unsigned long long*)&out[2], (unsigned long long)__double_as_longlong(e2));
}

// ---- micro-benchmarks of the 16x16 register factorisation: where do the ~330 cycles per column go? -------------------
template <int MODE>  // 0 = full potf2_16, 1 = pivot chains only (no trailing updates), 2 = trailing updates only (fixed y)
__global__ void potf2_bench(double* out, long long* cyc) {
  __shared__ double sL[ppo::NB][ppo::LP];
  __shared__ double sT[16][17];
  const int lane = threadIdx.x & 63, r = lane & 15;
  for (int i = threadIdx.x; i < 64 * 65; i += 64) (&sL[0][0])[i] = 0.0;
  __syncthreads();
  for (int c = 0; c < 64; ++c) sL[lane][c] = (lane == c) ? 70.0 : 0.01 * ((lane * 7 + c * 3) % 11);
  __syncthreads();
  double a[16], e[16];
  for (int c = 0; c < 16; ++c) { a[c] = sL[r][c]; e[c] = sL[16 + lane % 48][c]; }
  long long t0 = clock64();
  if (MODE == 3) {  // the full sweep three times in a loop: is the first (cold instruction cache) pass slower?
    for (int rep = 0; rep < 3; ++rep) {
      t0 = clock64();
      int bad = 0;
      ppo::ColSweep<0>::run(a, e, bad);
      e[0] += bad;
      const long long t1 = clock64();
      if (threadIdx.x == 0) cyc[3 + rep] = t1 - t0;
      for (int c = 0; c < 16; ++c) { a[c] = sL[r][c] + 1e-9 * e[c]; e[c] = sL[16 + lane % 48][c]; }
      __builtin_amdgcn_s_sleep(1);
    }
  } else if (MODE == 0) {
    int bad = 0;
    ppo::ColSweep<0>::run(a, e, bad);
    e[0] += bad;
  } else if (MODE == 1) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const double d = ppo::row_bcast<0>(a[j]);
      const double h = 0.5 * d;
      double y = __builtin_amdgcn_rsq(d);
      y = y * __builtin_fma(-h, y * y, 1.5);
      y = y * __builtin_fma(-h, y * y, 1.5);
      a[j] = a[j] * y;
      if (j + 1 < 16) ppo::fmac_bcast<0, true>(a[j + 1], a[j], a[j]);
    }
  } else {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const double l = a[j], le = e[j];
#pragma unroll
      for (int c = j + 1; c < 16; ++c) { ppo::fmac_bcast<0, false>(a[c], l, l); ppo::fmac_bcast<0, false>(e[c], l, le); }
    }
  }
  const long long t1 = clock64();
  double sum = 0;
  for (int c = 0; c < 16; ++c) sum += a[c] + e[c];
  out[threadIdx.x] = sum;
  if (threadIdx.x == 0) cyc[MODE] = t1 - t0;
}

// ---- FP64 MFMA issue rate: NACC independent accumulator chains, 256 instructions per chain, one wave ----------------------
template <int NACC>
__global__ void mfma_rate(double* out, long long* cyc) {
  typedef double v4d __attribute__((ext_vector_type(4)));
  v4d acc[NACC];
  for (int q = 0; q < NACC; ++q) acc[q] = {0.0, 0.0, 0.0, 0.0};
  const double a = 1.0 + threadIdx.x * 1e-3, b = 0.5 - threadIdx.x * 1e-3;
  const long long t0 = clock64();
#pragma unroll 1
  for (int it = 0; it < 64; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int q = 0; q < NACC; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[q], 0, 0, 0);
  }
  const long long t1 = clock64();
  double sum = 0;
  for (int q = 0; q < NACC; ++q) sum += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
  out[threadIdx.x] = sum;
  if (threadIdx.x == 0) cyc[NACC] = (t1 - t0) / (256 * NACC);
}
int main() {
  { double* o; hipMalloc((void**)&o, 32); hipMemset(o, 0, 32); hipLaunchKernelGGL(rsq_acc, dim3(64), dim3(256), 0, 0, o); double h[4]; hipMemcpy(h, o, 32, hipMemcpyDeviceToHost);
    printf("v_rsq_f64 max rel err: raw %.3e, 1 Newton %.3e, 2 Newton %.3e, one third-order step %.3e\n", h[0], h[1], h[2], h[3]); }
  {
    double* o; long long* c; hipMalloc((void**)&o, 64 * 8); hipMalloc((void**)&c, 8 * 8);
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(potf2_bench<0>, dim3(1), dim3(64), 0, 0, o, c);
      hipLaunchKernelGGL(potf2_bench<1>, dim3(1), dim3(64), 0, 0, o, c);
      hipLaunchKernelGGL(potf2_bench<2>, dim3(1), dim3(64), 0, 0, o, c);
    }
    hipLaunchKernelGGL(potf2_bench<3>, dim3(1), dim3(64), 0, 0, o, c);
    long long h[6]; hipMemcpy(h, c, 48, hipMemcpyDeviceToHost);
    printf("potf2_16 sweep looped in one kernel: pass 1 %lld, pass 2 %lld, pass 3 %lld ticks\n", h[3], h[4], h[5]);
    printf("potf2_16 register sweep (s_memtime ticks): full %lld, pivot chains only %lld, 240 DPP fmacs only %lld\n", h[0], h[1], h[2]);
  }
  {
    double* o; long long* c; hipMalloc((void**)&o, 64 * 8); hipMalloc((void**)&c, 16 * 8);
    hipLaunchKernelGGL(mfma_rate<1>, dim3(1), dim3(64), 0, 0, o, c);
    hipLaunchKernelGGL(mfma_rate<2>, dim3(1), dim3(64), 0, 0, o, c);
    hipLaunchKernelGGL(mfma_rate<4>, dim3(1), dim3(64), 0, 0, o, c);
    hipLaunchKernelGGL(mfma_rate<8>, dim3(1), dim3(64), 0, 0, o, c);
    long long h[16]; hipMemcpy(h, c, 128, hipMemcpyDeviceToHost);
    printf("v_mfma_f64_16x16x4 cycles per instruction, one wave: 1 chain %lld, 2 chains %lld, 4 chains %lld, 8 chains %lld\n", h[1], h[2], h[4], h[8]);
  }
  const int n = 1664;
  std::vector<double> A((size_t)n * n, 0.0), b(n, 1.0);
  srand(1);
  // SPD: diagonally dominant symmetric
  for (int c = 0; c < n; ++c)
    for (int r = c; r < n; ++r) {
      double v = (r == c) ? n * 1.0 : ((rand() % 2001) - 1000) * 1e-3;
      A[(size_t)c * n + r] = v;
      A[(size_t)r * n + c] = v;
    }
  double *dA, *dA0, *dx, *db, *dws; int* dinfo;
  hipMalloc((void**)&dA, A.size() * 8); hipMalloc((void**)&dA0, A.size() * 8); hipMalloc((void**)&dx, n * 8); hipMalloc((void**)&db, n * 8); hipMalloc((void**)&dws, ppo::chol_workspace_doubles(n) * 8); hipMemset(dws, 0, ppo::chol_workspace_doubles(n) * 8); hipMalloc((void**)&dinfo, 4);
  hipMemcpy(dA0, A.data(), A.size() * 8, hipMemcpyHostToDevice);
  hipStream_t s; hipStreamCreate(&s);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 5; ++rep) {
    hipMemcpyAsync(dA, dA0, A.size() * 8, hipMemcpyDeviceToDevice, s);
    hipMemcpyAsync(db, b.data(), n * 8, hipMemcpyHostToDevice, s);
    hipEventRecord(e0, s);
    ppo::chol_factor_solve(dA, n, db, dx, dinfo, dws, s);
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("rep %d: factor+solve %.1f us\n", rep, ms * 1e3);
  }
  // residual check
  std::vector<double> x(n);
  hipMemcpy(x.data(), dx, n * 8, hipMemcpyDeviceToHost);
  double rmax = 0;
  for (int r = 0; r < n; ++r) { double sres = -1.0; for (int c = 0; c < n; ++c) sres += A[(size_t)c * n + r] * x[c]; rmax = fmax(rmax, fabs(sres)); }
  int info; hipMemcpy(&info, dinfo, 4, hipMemcpyDeviceToHost);
  printf("info %d max residual %.3e\n", info, rmax);
  // phase stamps of column workgroup 1 of the second step (update of panel 0 fused with the factorisation of panel 1)
  hipMemcpyAsync(dA, dA0, A.size() * 8, hipMemcpyDeviceToDevice, s);
  hipLaunchKernelGGL(ppo::chol_step, dim3(1 + 2 * (n / 64 - 1)), dim3(ppo::kStepThreads), 0, s, (const int*)nullptr, 0, dA, n, 0, db, dws, dx, dinfo, 0, (double*)nullptr, ppo::step_no_overlap(n / 64), ppo::StepFix{nullptr, nullptr, nullptr, 0, 0});
  { const int m = n / 64 - 1; hipLaunchKernelGGL(ppo::chol_step, dim3(1 + 2 * (m - 1) + m * (m - 1) / 2), dim3(ppo::kStepThreads), 0, s, (const int*)nullptr, 0, dA, n, 64, db, dws, dx, dinfo, 0, (double*)nullptr, ppo::step_no_overlap(n / 64), ppo::StepFix{nullptr, nullptr, nullptr, 0, 0}); }
  hipStreamSynchronize(s);
  long long st[48];
  hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamps), sizeof(st));
  printf("  entry -> column path %.2f us -> addresses ready %.2f us; first tile issued %.2f us, second %.2f us, halves + rhs %.2f us\n", (st[7] - st[0]) * 0.01,
         (st[15] - st[7]) * 0.01, (st[16] - st[15]) * 0.01, (st[17] - st[16]) * 0.01, (st[6] - st[17]) * 0.01);
  printf("  loads issued        +%.2f us\n", (st[6] - st[0]) * 0.01);
  printf("  tiles in LDS        +%.2f us\n", (st[5] - st[6]) * 0.01);
  printf("  column-0 update     +%.2f us\n", (st[1] - st[5]) * 0.01);
  printf("  factor + side jobs  +%.2f us\n", (st[2] - st[1]) * 0.01);
  printf("  last solve step     +%.2f us\n", (st[3] - st[2]) * 0.01);
  printf("  stored              +%.2f us\n", (st[4] - st[3]) * 0.01);
  for (int i = 8; i < 15; ++i) printf("  factor stamp %d: +%.2f us\n", i, (st[i] - (i == 8 ? st[1] : st[i - 1])) * 0.01);
  for (int kk = 0; kk < 4; ++kk) printf("  wave 1 side jobs of phase %d: %.2f us\n", kk, (st[28 + kk] - st[24 + kk]) * 0.01);
  if (1) printf("  potf2(kk=1): LDS loads %.2f us, sweep %.2f, stores %.2f, then to the barrier stamp %.2f\n", (st[40] - st[9]) * 0.01, (st[41] - st[40]) * 0.01, (st[42] - st[41]) * 0.01, (st[10] - st[42]) * 0.01);
  // backward pipeline: block 1's view (stamps 22..29 = moment x_i was seen, i = 2..9)
  hipMemcpyAsync(dA, dA0, A.size() * 8, hipMemcpyDeviceToDevice, s);
  ppo::chol_factor_solve(dA, n, db, dx, dinfo, dws, s);
  hipStreamSynchronize(s);
  hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamps), sizeof(st));
  for (int i = 9; i >= 3; --i) printf("  hop: x_%d seen -> x_%d seen  %.2f us\n", i, i - 1, (st[20 + i - 1] - st[20 + i]) * 0.01);
  printf("  block 1: x_3 seen -> loop left %.2f us -> partial c_1 written %.2f us -> x_2 seen %.2f us -> all partial products in %.2f us -> x_1 stored %.2f us\n", (st[35] - st[23]) * 0.01,
         (st[36] - st[35]) * 0.01, (st[33] - st[36]) * 0.01, (st[30] - st[33]) * 0.01, (st[31] - st[30]) * 0.01);
  return 0;
}
