// boundary_vs_barrier.hip -- development probe (not part of the product): what ONE persistent launch per LM pass (review item 4) has to beat.
// A solve attempt on an ICL-sized window is five dependent launches of 8..100 workgroups; a persistent kernel replaces the four boundaries between
// them by grid barriers.  This measures both prices on the device itself, on the same dependent toy phases:
//   chain:      P phases as P launches on one stream (what the product does; the queue is kept full: 5 x 40 launches between two events)
//   persistent: the same P phases inside ONE launch, a grid barrier (one agent-scope atomic counter, bounded spin) between them
// for G = 8, 24, 48, 96 workgroups of 256 threads, with an EMPTY phase (pure boundary / pure barrier) and with a phase that does ~2 us of
// dependent work (reads what another workgroup wrote in the phase before: a boundary or a barrier is REQUIRED between them).
// Build + run: tools/r06/boundary_vs_barrier.sh -> gpurun_out/r06_boundary_vs_barrier.txt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kThreads = 256;
constexpr int kSpinCap = 4000000;  // bounded: a barrier that never opens ends the kernel with the flag raised instead of hanging the device

__device__ __forceinline__ void phase_work(double* __restrict__ buf, int g, int G, int phase, int work) {
  // every workgroup reads the block its LEFT neighbour wrote in the phase before and writes its own: `work` dependent rounds of 256 x 8 bytes
  const int src = (g + G - 1) % G;
  double v = 0.0;
  for (int w = 0; w < work; ++w) {
    const double x = __builtin_nontemporal_load(&buf[((size_t)((phase + 1) & 1) * G + src) * kThreads + (threadIdx.x + 17 * w) % kThreads]);
    v = __builtin_fma(v, 0.999, x + 1.0);
  }
  buf[((size_t)(phase & 1) * G + g) * kThreads + threadIdx.x] = v;
}

__global__ __launch_bounds__(kThreads) void k_phase(double* buf, int phase, int work) { phase_work(buf, blockIdx.x, gridDim.x, phase, work); }

__global__ __launch_bounds__(kThreads) void k_persistent(double* buf, int phases, int work, unsigned* counter, int* dead) {
  const int G = gridDim.x;
  __shared__ int s_dead;
  for (int p = 0; p < phases; ++p) {
    phase_work(buf, blockIdx.x, G, p, work);
    // grid barrier: release what this workgroup wrote, count in, wait until everybody of this phase has
    __syncthreads();
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned want = (unsigned)(p + 1) * (unsigned)G;
      int spins = 0, d = 0;
      while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        if (++spins > kSpinCap || __hip_atomic_load(dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
          __hip_atomic_store(dead, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          d = 1;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      s_dead = d;
    }
    __syncthreads();
    if (s_dead) return;
  }
}

int main() {
  hipStream_t s;
  hipStreamCreate(&s);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  double* buf;
  unsigned* counter;
  int* dead;
  hipMalloc((void**)&buf, 2 * 256 * kThreads * sizeof(double));
  hipMemset(buf, 0, 2 * 256 * kThreads * sizeof(double));
  hipMalloc((void**)&counter, 4);
  hipMalloc((void**)&dead, 4);
  const int P = 5, R = 40;  // 5 phases per "solve attempt", 40 attempts queued back to back
  printf("# per PHASE, microseconds (best of 7 timed blocks of %d x %d phases); work = dependent rounds inside a phase\n", R, P);
  printf("# %4s %5s | %10s %10s | %s\n", "G", "work", "chain", "persistent", "a boundary costs this much more than a barrier");
  for (int work : {0, 24, 96}) {
    for (int G : {8, 24, 48, 96, 192}) {
      double best_chain = 1e30, best_pers = 1e30;
      int any_dead = 0;
      for (int rep = 0; rep < 8; ++rep) {
        hipEventRecord(e0, s);
        for (int r = 0; r < R; ++r)
          for (int p = 0; p < P; ++p) hipLaunchKernelGGL(k_phase, dim3(G), dim3(kThreads), 0, s, buf, p, work);
        hipEventRecord(e1, s);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms * 1e3 / (R * P) < best_chain) best_chain = ms * 1e3 / (R * P);
        hipMemsetAsync(counter, 0, 4, s);
        hipMemsetAsync(dead, 0, 4, s);
        hipEventRecord(e0, s);
        hipLaunchKernelGGL(k_persistent, dim3(G), dim3(kThreads), 0, s, buf, R * P, work, counter, dead);
        hipEventRecord(e1, s);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms * 1e3 / (R * P) < best_pers) best_pers = ms * 1e3 / (R * P);
        int d = 0;
        hipMemcpy(&d, dead, 4, hipMemcpyDeviceToHost);
        any_dead |= d;
      }
      printf("  %4d %5d | %10.2f %10.2f | %+.2f%s\n", G, work, best_chain, best_pers, best_chain - best_pers, any_dead ? "   (A BARRIER TIMED OUT)" : "");
    }
  }
  return 0;
}
