// chol_check.hip -- development probe (not part of the product): the native dense solve of csrc/ppo_chol.hip on random SPD
// matrices of several sizes: residual, agreement of the one-launch factorisation with the per-panel one, time per solve, and
// the timeline of one block row's head inside chol_fwd_all.  Build + run: tools/run_chol_check.sh
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
__device__ long long g_stamps[64 * 16];
#define STAMPF(i) do { if (head && threadIdx.x == 0) g_stamps[bi * 16 + (i)] = wall_clock64(); } while (0)  // every head's timeline
__device__ long long g_wg[512 * 4];
#define STAMPG(i) do { if ((threadIdx.x & 63) == 0 && ((i) == 3 || threadIdx.x == 0) && blockIdx.x < 512) g_wg[blockIdx.x * 4 + (i)] = wall_clock64(); } while (0)
#include "../point-plane-object-slam_amd/csrc/ppo_chol.hip"

static double run(int n, bool panels, int reps, std::vector<double>* xout, int* info_out) {
  const int ld = ppo::chol_padded_ld(n);
  std::vector<double> A((size_t)ld * ld, 0.0), b(ld, 0.0);
  srand(n);
  for (int c = 0; c < n; ++c)
    for (int r = c; r < n; ++r) {
      const double v = (r == c) ? 40.0 + (rand() % 100) * 0.1 : ((rand() % 2001) - 1000) * 1e-3 * ((r - c) < 200 ? 1.0 : 0.05);
      A[(size_t)c * ld + r] = v;
    }
  for (int c = n; c < ld; ++c) A[(size_t)c * ld + c] = 1.0;
  for (int r = 0; r < n; ++r) b[r] = 1.0 + 0.01 * (r % 17);
  double *dA, *dA0, *dx, *db, *dws; int* dinfo;
  const size_t wsn = ppo::chol_workspace_doubles(n);
  hipMalloc((void**)&dA, A.size() * 8); hipMalloc((void**)&dA0, A.size() * 8); hipMalloc((void**)&dx, ld * 8); hipMalloc((void**)&db, ld * 8);
  hipMalloc((void**)&dws, wsn * 8); hipMalloc((void**)&dinfo, 4);
  hipMemset(dws, 0, wsn * 8);
  hipMemcpy(dA0, A.data(), A.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(db, b.data(), ld * 8, hipMemcpyHostToDevice);
  hipStream_t s; hipStreamCreate(&s);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  double best = 1e30;
  for (int rep = 0; rep < reps; ++rep) {
    hipMemcpyAsync(dA, dA0, A.size() * 8, hipMemcpyDeviceToDevice, s);
    hipEventRecord(e0, s);
    ppo::chol_factor_solve(dA, ld, db, dx, dinfo, dws, s);
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms * 1e3 < best) best = ms * 1e3;
  }
  xout->resize(ld);
  hipMemcpy(xout->data(), dx, ld * 8, hipMemcpyDeviceToHost);
  hipMemcpy(info_out, dinfo, 4, hipMemcpyDeviceToHost);
  // residual of the symmetric system (lower triangle stored)
  double rmax = 0;
  for (int r = 0; r < n; ++r) {
    double sres = -b[r];
    for (int c = 0; c < n; ++c) sres += (r >= c ? A[(size_t)c * ld + r] : A[(size_t)r * ld + c]) * (*xout)[c];
    rmax = fmax(rmax, fabs(sres));
  }
  printf("n %4d (ld %4d) %-10s: best of %d %.1f us, info %d, max residual %.2e\n", n, ld, panels ? "per panel" : "one launch", reps, best, *info_out, rmax);
  hipFree(dA); hipFree(dA0); hipFree(dx); hipFree(db); hipFree(dws); hipFree(dinfo);
  return rmax;
}

int main(int argc, char** argv) {
  // the environment switch is read once per process (static): the per-panel arm runs in a separate process (argv[1] = "panels")
  const bool panels = argc > 1 && argv[1][0] == 'p';
  setenv("PPO_BA_CHOL_ONELAUNCH", panels ? "0" : "1", 1);
  const int sizes[] = {64, 100, 128, 192, 320, 594, 1194, 1216, 1644, 2048, 2112, 2600};
  for (int n : sizes) {
    std::vector<double> x; int info;
    run(n, panels, 6, &x, &info);
    double cs = 0; for (double v : x) cs += v;
    printf("     checksum of x %.15e\n", cs);
    fflush(stdout);
  }
  if (!panels) {
    std::vector<double> x; int info;
    run(1194, false, 3, &x, &info);
    static long long st[64 * 16];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamps), sizeof(st));
    const long long t0 = st[0];
    printf("heads of n = 1194 (us since head 0 entered): entry | updates done | piece 0..3 seen | subst done | potf2(0..3) done | end\n");
    for (int i = 0; i < 19; ++i) {
      printf("  head %2d:", i);
      for (int k = 0; k < 12; ++k) printf(" %7.2f", (st[i * 16 + k] - t0) * 0.01);
      printf("\n");
    }
    static long long wg[512 * 4];
    hipMemcpyFromSymbol(wg, HIP_SYMBOL(g_wg), sizeof(wg));
    printf("workgroups of rows 2..6: (row, column | head): updates done, substitution done, tile flagged (us since head 0 entered)\n");
    for (int i = 2; i <= 6; ++i)
      for (int sl = 0; sl < i; ++sl) {
        const int b = 1 + i * (i - 1) / 2 + sl;
        printf("  (%d,%d)%s: %7.2f %7.2f %7.2f\n", i, sl == i - 1 ? i - 1 : sl, sl == i - 1 ? " head" : "     ", (wg[b * 4 + 1] - t0) * 0.01, (wg[b * 4 + 2] - t0) * 0.01, (wg[b * 4 + 3] - t0) * 0.01);
      }
  }
  return 0;
}
