// exec_mask_probe.hip -- development probe (not part of the product): does a wave64 FP64 vector instruction get cheaper when only one, two or three of its four
// 16-lane rows are active?  (The register factorisation of the dense solve computes the 16 x 16 tile's own rows redundantly in all four rows of lanes -- csrc/ppo_chol.hip,
// ColSweep; if the hardware skipped the passes of rows whose EXEC bits are all zero, computing them in ONE row would cut the sweep's issue time.)
// 512 v_fma_f64 on 8 independent accumulators, and 512 v_fmac_f64_dpp row_newbcast, timed with s_memtime on one wave, for 64 / 48 / 32 / 16 / 1 active lanes.
// Build + run: hipcc -O3 --offload-arch=gfx950 -o build/exec_mask_probe tools/r06/exec_mask_probe.hip && build/exec_mask_probe
#include <hip/hip_runtime.h>
#include <cstdio>

template <int DPP>
__global__ __launch_bounds__(64) void k_probe(int active, double* out, long long* ticks) {
  const int lane = threadIdx.x;
  double acc[8], m = 1.0 + 1e-9 * lane, b = 1e-3 * lane;
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 1.0 + k + lane;
  long long t0 = 0, t1 = 0;
  if (lane < active) {
    t0 = clock64();
#pragma unroll 1
    for (int it = 0; it < 8; ++it) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          if (DPP) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(acc[k]) : "v"(m), "v"(b));
          else asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(acc[k]) : "v"(m), "v"(b));
        }
      }
    }
    t1 = clock64();
  }
  double s = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) s += acc[k];
  out[lane] = s;
  if (lane == 0) *ticks = t1 - t0;
}

int main() {
  double* out;
  long long* ticks;
  hipMalloc((void**)&out, 64 * 8);
  hipMalloc((void**)&ticks, 8);
  printf("# ticks of clock64() (s_memtime; profiles/r06_potf2_anatomy.txt calibrates one tick) for 512 instructions on one wave; per instruction in brackets\n");
  for (int dpp = 0; dpp < 2; ++dpp)
    for (int active : {64, 48, 32, 16, 1}) {
      long long best = 1LL << 60;
      for (int rep = 0; rep < 5; ++rep) {
        if (dpp) hipLaunchKernelGGL(k_probe<1>, dim3(1), dim3(64), 0, 0, active, out, ticks);
        else hipLaunchKernelGGL(k_probe<0>, dim3(1), dim3(64), 0, 0, active, out, ticks);
        long long t;
        hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
        if (t < best) best = t;
      }
      printf("%-34s %2d active lanes: %6lld ticks (%.2f per instruction)\n", dpp ? "v_fmac_f64_dpp row_newbcast" : "v_fma_f64", active, best, best / 512.0);
    }
  return 0;
}
