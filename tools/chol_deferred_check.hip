// chol_deferred_check.hip -- development probe (not part of the product): the schedule logic of the overlapped reduction in csrc/ppo_chol.hip
// WITHOUT pair workgroups.  With an empty chunk list a schedule only decides which trailing tiles take their update as atomic adds and makes
// the right-hand side an in-place, atomically updated vector; the solution must agree with the plain per-panel solve to rounding.
// Build + run: tools/run_chol_deferred_check.sh
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../point-plane-object-slam_amd/csrc/ppo_chol.hip"

int main() {
  const int sizes[] = {100, 128, 256, 320, 594, 1194, 1644, 2048, 2600, 4096};  // (2600, 4096: more blocks than the ring of the backward substitution holds)
  int bad = 0;
  for (int n : sizes) {
    const int ld = ppo::chol_padded_ld(n), nb = ld / 64;
    std::vector<double> A((size_t)ld * ld, 0.0), b(ld, 0.0);
    srand(n);
    for (int c = 0; c < n; ++c)
      for (int r = c; r < n; ++r) A[(size_t)c * ld + r] = (r == c) ? 40.0 + (rand() % 100) * 0.1 : ((rand() % 2001) - 1000) * 1e-3 * ((r - c) < 200 ? 1.0 : 0.05);
    for (int c = n; c < ld; ++c) A[(size_t)c * ld + c] = 1.0;
    for (int r = 0; r < n; ++r) b[r] = 1.0 + 0.01 * (r % 17);
    double *dA, *dA0, *dx, *db, *db0, *dws; int* dinfo;
    const size_t wsn = ppo::chol_workspace_doubles(n);
    hipMalloc((void**)&dA, A.size() * 8); hipMalloc((void**)&dA0, A.size() * 8); hipMalloc((void**)&dx, ld * 8); hipMalloc((void**)&db, ld * 8); hipMalloc((void**)&db0, ld * 8);
    hipMalloc((void**)&dws, wsn * 8); hipMalloc((void**)&dinfo, 4);
    hipMemset(dws, 0, wsn * 8);
    hipMemcpy(dA0, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(db0, b.data(), ld * 8, hipMemcpyHostToDevice);
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<int32_t> cb(nb + 1, 0);  // no chunks
    std::vector<double> xref(ld), x(ld);
    for (int variant = 0; variant < 4; ++variant) {
      std::vector<int> fin(nb + 1, nb);
      if (variant == 1) for (int k = 0; k <= nb; ++k) fin[k] = std::min(nb, k + 1);                       // as late as allowed: two column blocks of atomic tiles per launch
      if (variant == 2) { int f = 1; for (int k = 0; k <= nb; ++k) { fin[k] = std::min(nb, f); f += 2 + k; } }  // 1 3 6 10 15 ...
      if (variant == 3) { int f = 2; for (int k = 0; k <= nb; ++k) { fin[k] = std::min(nb, f); f += 2; } }       // 2 4 6 ...
      double best = 1e30;
      for (int rep = 0; rep < 5; ++rep) {
        hipMemcpyAsync(dA, dA0, A.size() * 8, hipMemcpyDeviceToDevice, s);
        hipMemcpyAsync(db, db0, ld * 8, hipMemcpyDeviceToDevice, s);
        hipEventRecord(e0, s);
        if (variant == 0) ppo::chol_factor_solve(dA, ld, db, dx, dinfo, dws, s);
        else {
          ppo::CholOverlap ov{};
          ov.ctx.S = dA; ov.ctx.bs = db; ov.ctx.ld = ld;
          ov.cb_start = cb.data(); ov.fin = fin.data();
          ppo::chol_factor_solve_overlapped(dA, ld, dx, dinfo, dws, ov, s);
        }
        hipEventRecord(e1, s);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms * 1e3 < best) best = ms * 1e3;
      }
      int info;
      hipMemcpy(x.data(), dx, ld * 8, hipMemcpyDeviceToHost);
      hipMemcpy(&info, dinfo, 4, hipMemcpyDeviceToHost);
      if (variant == 0) xref = x;
      double dmax = 0, xmax = 0;
      for (int i = 0; i < n; ++i) { dmax = fmax(dmax, fabs(x[i] - xref[i])); xmax = fmax(xmax, fabs(xref[i])); }
      printf("n %4d nb %2d variant %d: %.1f us, info %d, max |x - x_plain| %.2e (max |x| %.2e)\n", n, nb, variant, best, info, dmax, xmax);
      if (!(dmax <= 1e-12 * xmax) || info != 0) ++bad;
    }
    hipFree(dA); hipFree(dA0); hipFree(dx); hipFree(db); hipFree(db0); hipFree(dws); hipFree(dinfo);
  }
  printf(bad ? "FAILED (%d)\n" : "ok\n", bad);
  return bad != 0;
}
