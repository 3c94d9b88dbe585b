// chol_probe.hip -- development probe (not part of the product): times the phases of chol_panel with s_memtime stamps
// and the three Cholesky kernels with HIP events on a random SPD matrix.  Build: see tools/run_chol_probe.sh
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#define PPO_CHOL_STAMPS 1
__device__ long long g_stamps[32];
#define STAMP(i) do { if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) g_stamps[i] = wall_clock64(); } while (0)
#include "../point-plane-object-slam_amd/csrc/ppo_chol.hip"

__global__ void rsq_acc(double* out) {
  const double d = 1.0 + 1e-3 * (threadIdx.x + 256.0 * blockIdx.x) * (1.0 + 0.37 * threadIdx.x);
  const double h = 0.5 * d;
  double y = __builtin_amdgcn_rsq(d);
  const double e0 = fabs(y * sqrt(d) - 1.0);
  y = y * __builtin_fma(-h, y * y, 1.5);
  const double e1 = fabs(y * sqrt(d) - 1.0);
  y = y * __builtin_fma(-h, y * y, 1.5);
  const double e2 = fabs(y * sqrt(d) - 1.0);
  atomicMax((unsigned long long*)&out[0], (unsigned long long)__double_as_longlong(e0));
  atomicMax((unsigned long long*)&out[1], (unsigned long long)__double_as_longlong(e1));
  atomicMax((unsigned long long*)&out[2], (unsigned long long)__double_as_longlong(e2));
}
int main() {
  { double* o; hipMalloc((void**)&o, 32); hipMemset(o, 0, 32); hipLaunchKernelGGL(rsq_acc, dim3(64), dim3(256), 0, 0, o); double h[3]; hipMemcpy(h, o, 24, hipMemcpyDeviceToHost);
    printf("v_rsq_f64 max rel err: raw %.3e, 1 Newton %.3e, 2 Newton %.3e\n", h[0], h[1], h[2]); }
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
  double *dA, *dA0, *dx; int* dinfo;
  hipMalloc((void**)&dA, A.size() * 8); hipMalloc((void**)&dA0, A.size() * 8); hipMalloc((void**)&dx, n * 8); hipMalloc((void**)&dinfo, 4);
  hipMemcpy(dA0, A.data(), A.size() * 8, hipMemcpyHostToDevice);
  hipStream_t s; hipStreamCreate(&s);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 5; ++rep) {
    hipMemcpyAsync(dA, dA0, A.size() * 8, hipMemcpyDeviceToDevice, s);
    hipMemcpyAsync(dx, b.data(), n * 8, hipMemcpyHostToDevice, s);
    hipEventRecord(e0, s);
    ppo::chol_factor_solve(dA, n, dx, dinfo, nullptr, s);
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
  // phase stamps of the LAST block of the first panel launch: rerun one panel
  hipMemcpyAsync(dA, dA0, A.size() * 8, hipMemcpyDeviceToDevice, s);
  hipLaunchKernelGGL(ppo::chol_panel, dim3(n / 64), dim3(256), 0, s, dA, n, 0, dx, dinfo);
  hipStreamSynchronize(s);
  long long st[32];
  hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamps), sizeof(st));
  const char* names[] = {"start", "loaded->LDS", "factor_64 done", "R subst done", "stored"};
  for (int i = 1; i < 5; ++i) printf("  %-16s +%lld ticks (100 MHz wall clock => %.2f us)\n", names[i], st[i] - st[i - 1], (st[i] - st[i - 1]) * 0.01);
  for (int i = 8; i < 20; ++i) if (st[i]) printf("  factor stamp %d: +%.2f us\n", i, (st[i] - st[i - 1 < 8 ? 1 : i - 1]) * 0.01);
  return 0;
}
