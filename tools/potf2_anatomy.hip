// potf2_anatomy.hip -- development probe (not part of the product), round 6: where do the cycles of the 16 x 16 register factorisation go?
//
// csrc/ppo_chol.hip: potf2_16 / ColSweep<0> is the serial core of the dense solve (76 of them at the BASELINE size, 137 us of a 358 us LM iteration).
// One pivot = v_mov_b64_dpp -> v_rsq_f64 -> 6 dependent f64 operations -> 2 x (15 - J) v_fmac_f64_dpp; 384 f64 instructions per tile.  The ledger of
// DESIGN.md 4.1 held three statements that do not fit together (row 9: not issue-bound; row 11: not latency-bound; e884af2: issue-bound).  This probe
// settles it with ONE wave, registers only (no LDS, no side waves, no loader waves), s_memtime around `reps` back-to-back sweeps:
//   (1) the sweep as the product runs it, without the extra rows (E), without the sweep multiply-adds (chain only), without the chain (multiply-adds only);
//   (2) the cost of every instruction class involved, as a DEPENDENT chain and as an INDEPENDENT stream of 8 accumulators: if the independent stream is
//       as slow as the dependent one, FP64 instructions of one wave do not overlap and "issue-bound" and "latency-bound" are the same statement;
//   (3) the same sweep with 1, 2, 4 waves of the workgroup running it side by side (different SIMDs) and with two waves on ONE SIMD;
//   (4) the pieces a 4-wide blocked tile (one v_mfma_f64_16x16x4 per panel for the trailing columns) would add: the selects that build the matrix
//       operands out of the row-per-lane layout, the matrix instruction, and the way back (v_permlane16/32_swap all-gather of the four 16-lane rows).
// Build + run: tools/r06/run_potf2_anatomy.sh (hipcc --offload-arch=gfx950).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../point-plane-object-slam_amd/csrc/ppo_chol.hip"

namespace {
using ppo::SB;
typedef double v4d __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void init_rows(double (&a)[SB], double (&e)[SB], int lane) {
  const int r = lane & 15;
#pragma unroll
  for (int c = 0; c < SB; ++c) {
    a[c] = (r == c) ? 70.0 + r : 0.01 * ((r * 7 + c * 3) % 11 + 1);
    a[c] = (c > r) ? 0.01 * ((c * 7 + r * 3) % 11 + 1) : a[c];  // symmetric
    e[c] = 0.02 * (((lane >> 4) * 5 + r * 3 + c) % 13);
  }
}

// MODE 0 full sweep, 1 without E, 2 chain only (pivot + the one multiply-add the next pivot waits for), 3 multiply-adds only
template <int J, int MODE>
struct Sweep {
  static __device__ __forceinline__ void run(double (&a)[SB], double (&e)[SB]) {
    double l = a[J], le = e[J];
    if (MODE != 3) {
      const double d = ppo::row_bcast<J>(a[J]);
      const double y0 = __builtin_amdgcn_rsq(d);
      const double e1 = __builtin_fma(-(d * y0), y0, 1.0);
      const double y = __builtin_fma(y0, e1 * __builtin_fma(0.375, e1, 0.5), y0);
      l = a[J] * y;
      le = e[J] * y;
      a[J] = l;
      e[J] = le;
    }
    if constexpr (J + 1 < SB) {
      if (MODE == 2) {
        ppo::fmac_bcast<J + 1, true>(a[J + 1], l, l);
      } else {
#pragma unroll
        for (int c = J + 1; c < SB; ++c) {
          if (c == J + 1 && MODE != 3) ppo::fmac_bcast<J + 1, true>(a[c], l, l);
          else asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:1 row_mask:0xf bank_mask:0xf" : "+v"(a[c]) : "v"(l), "v"(l));
          if (MODE != 1) asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:1 row_mask:0xf bank_mask:0xf" : "+v"(e[c]) : "v"(l), "v"(le));
        }
      }
      Sweep<J + 1, MODE>::run(a, e);
    }
  }
};

// waves: the first `active` waves of the workgroup run the sweep; wave 0's ticks are reported
template <int MODE>
__global__ __launch_bounds__(512) void k_sweep(int reps, unsigned active_mask, double* out, long long* ticks, long long* wall) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (!((active_mask >> wave) & 1u)) return;
  double a[SB], e[SB];
  init_rows(a, e, lane);
  double sink = 0.0;
  const long long w0 = wall_clock64();
  const long long t0 = clock64();
#pragma unroll 1
  for (int r = 0; r < reps; ++r) {
    Sweep<0, MODE>::run(a, e);
    sink += a[15] + e[15];
    init_rows(a, e, lane);
    a[0] += 1e-9 * sink;
  }
  const long long t1 = clock64();
  const long long w1 = wall_clock64();
  out[threadIdx.x] = sink;
  if (threadIdx.x == 0) { *ticks = t1 - t0; *wall = w1 - w0; }
}
// the loop overhead of k_sweep alone (re-initialisation of 32 registers per repetition)
__global__ __launch_bounds__(512) void k_sweep_empty(int reps, double* out, long long* ticks) {
  const int lane = threadIdx.x & 63;
  double a[SB], e[SB];
  init_rows(a, e, lane);
  double sink = 0.0;
  const long long t0 = clock64();
#pragma unroll 1
  for (int r = 0; r < reps; ++r) {
    asm volatile("" : "+v"(a[0]), "+v"(a[15]), "+v"(e[15]));
    sink += a[15] + e[15];
    init_rows(a, e, lane);
    a[0] += 1e-9 * sink;
  }
  const long long t1 = clock64();
  out[threadIdx.x] = sink;
  if (threadIdx.x == 0) *ticks = t1 - t0;
}

// ---- instruction classes: 512 instructions, dependent chain (NACC = 1) or NACC independent accumulators -----------------------------------
enum { I_FMA = 0, I_FMAC_DPP, I_MUL, I_RSQ, I_MOV_DPP, I_MFMA, I_CNDMASK, I_BPERMUTE, I_PERM16, I_PERM32, I_N };
template <int KIND, int NACC>
__global__ __launch_bounds__(64) void k_instr(double* out, long long* ticks) {
  double acc[8];
  v4d macc[8];
  int iacc[8];
  for (int q = 0; q < 8; ++q) { acc[q] = 1.0 + 1e-3 * (threadIdx.x + q); macc[q] = v4d{0.0, 0.0, 0.0, 0.0}; iacc[q] = threadIdx.x + q; }
  const double m = 1.0 - 1e-9 * threadIdx.x, b = 1e-9;
  const int lane4 = (threadIdx.x * 4) & 255;
  const long long t0 = clock64();
#pragma unroll 1
  for (int it = 0; it < 512 / (8 * NACC) * 1; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int q = 0; q < NACC; ++q) {
        if (KIND == I_FMA) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(acc[q]) : "v"(m), "v"(b));
        if (KIND == I_FMAC_DPP) asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(acc[q]) : "v"(b), "v"(m));
        if (KIND == I_MUL) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(acc[q]) : "v"(m));
        if (KIND == I_RSQ) asm volatile("v_rsq_f64 %0, %0" : "+v"(acc[q]));
        if (KIND == I_MOV_DPP) asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %0 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(acc[q]));
        if (KIND == I_MFMA) macc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(m, b, macc[q], 0, 0, 0);
        if (KIND == I_CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(iacc[q]) : "v"(lane4) : );
        if (KIND == I_BPERMUTE) iacc[q] = __builtin_amdgcn_ds_bpermute(lane4, iacc[q]);
        if (KIND == I_PERM16) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(iacc[q]), "+v"(iacc[(q + 1) & 7]));
        if (KIND == I_PERM32) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(iacc[q]), "+v"(iacc[(q + 1) & 7]));
      }
  }
  const long long t1 = clock64();
  double s = 0.0;
  for (int q = 0; q < 8; ++q) s += acc[q] + macc[q][0] + macc[q][1] + macc[q][2] + macc[q][3] + iacc[q];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) *ticks = t1 - t0;
}

// a chain operation followed by K independent multiply-adds, 64 times: does the independent work hide behind the dependent one?
template <int K>
__global__ __launch_bounds__(64) void k_mix(double* out, long long* ticks) {
  double chain = 1.0 + 1e-3 * threadIdx.x, acc[8];
  for (int q = 0; q < 8; ++q) acc[q] = 1.0 + 1e-3 * q;
  const double m = 1.0 - 1e-9 * threadIdx.x, b = 1e-9;
  const long long t0 = clock64();
#pragma unroll 1
  for (int it = 0; it < 64; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(chain) : "v"(m), "v"(b));
#pragma unroll
      for (int q = 0; q < K; ++q) asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(acc[q]) : "v"(b), "v"(m));
    }
  }
  const long long t1 = clock64();
  double s = chain;
  for (int q = 0; q < 8; ++q) s += acc[q];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) *ticks = t1 - t0;
}

// ---- the pieces of one 4-wide blocked panel step (timing only): operand selects + one matrix instruction + the way back -----------------------
// Row-per-lane layout: lane (r, g) = row r of the tile in every 16-lane row g.  v_mfma_f64_16x16x4 wants A[i][k] in lane (i, k): the panel entry
// a[4 p + g] -- a select among four registers by the lane's row group (3 x 2 v_cndmask_b32); C / D: four entries of row r, columns g + 4 reg -- a select
// among four registers for each of them; afterwards every lane needs the other twelve columns of its row back: an all-gather over the four rows,
// v_permlane16_swap + 2 x v_permlane32_swap per 32-bit half and value.
__global__ __launch_bounds__(64) void k_blocked_panel(int reps, double* out, long long* ticks) {
  const int lane = threadIdx.x & 63, g = lane >> 4;
  double a[SB], e[SB];
  init_rows(a, e, lane);
  double sink = 0.0;
  const long long t0 = clock64();
#pragma unroll 1
  for (int r = 0; r < reps; ++r) {
    // operand of the matrix instruction: a[g] of the panel's four columns
    const double p = g == 0 ? a[0] : (g == 1 ? a[1] : (g == 2 ? a[2] : a[3]));
    v4d c;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) c[reg] = g == 0 ? a[4 * reg] : (g == 1 ? a[4 * reg + 1] : (g == 2 ? a[4 * reg + 2] : a[4 * reg + 3]));
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(-p, p, c, 0, 0, 0);
    // all-gather of the four row groups, per value: x = [v0 v1 v2 v3] -> four registers holding v0, v1, v2, v3 in every row
#pragma unroll
    for (int reg = 1; reg < 4; ++reg) {  // (columns 4 .. 15: three of the four values)
      int lo = __double2loint(c[reg]), hi = __double2hiint(c[reg]);
      int lo1 = lo, hi1 = hi;
      asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(lo), "+v"(lo1));
      asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(hi), "+v"(hi1));
      int lo2 = lo, hi2 = hi, lo3 = lo1, hi3 = hi1;
      asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(lo), "+v"(lo2));
      asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(hi), "+v"(hi2));
      asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(lo1), "+v"(lo3));
      asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(hi1), "+v"(hi3));
      a[4 * reg] = __hiloint2double(hi, lo);
      a[4 * reg + 1] = __hiloint2double(hi1, lo1);
      a[4 * reg + 2] = __hiloint2double(hi2, lo2);
      a[4 * reg + 3] = __hiloint2double(hi3, lo3);
    }
    sink += a[15];
    a[0] += 1e-9 * sink;
  }
  const long long t1 = clock64();
  out[threadIdx.x] = sink;
  if (threadIdx.x == 0) *ticks = t1 - t0;
}
}  // namespace

int main() {
  double* o;
  long long *t, *w;
  hipMalloc((void**)&o, 512 * 8); hipMalloc((void**)&t, 8); hipMalloc((void**)&w, 8);
  auto ticks = [&]() { long long h; hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost); return h; };
  auto wallt = [&]() { long long h; hipMemcpy(&h, w, 8, hipMemcpyDeviceToHost); return h; };
  const int reps = 200;
  // ---- calibration: s_memtime ticks against the 100 MHz wall clock
  hipLaunchKernelGGL(k_sweep<0>, dim3(1), dim3(64), 0, 0, reps, 1u, o, t, w);
  hipLaunchKernelGGL(k_sweep<0>, dim3(1), dim3(64), 0, 0, reps, 1u, o, t, w);
  const double ns_per_tick = 10.0 * (double)wallt() / (double)ticks();
  printf("calibration: 1 s_memtime tick = %.3f ns (%lld ticks in %lld x 10 ns)\n", ns_per_tick, ticks(), wallt());
  hipLaunchKernelGGL(k_sweep_empty, dim3(1), dim3(64), 0, 0, reps, o, t);
  hipLaunchKernelGGL(k_sweep_empty, dim3(1), dim3(64), 0, 0, reps, o, t);
  const double empty = (double)ticks() / reps;
  printf("loop overhead (re-initialisation of the 32 row registers): %.0f ticks per repetition, subtracted below\n", empty);
  // ---- (1) the four anatomy numbers
  const char* names[4] = {"full sweep (as the product runs it)", "without the extra rows (E)", "chain only (16 pivots + the multiply-add each next pivot waits for)",
                          "multiply-adds only (240 v_fmac_f64_dpp, no chain)"};
  double sweep[4];
  for (int mode = 0; mode < 4; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      if (mode == 0) hipLaunchKernelGGL(k_sweep<0>, dim3(1), dim3(64), 0, 0, reps, 1u, o, t, w);
      if (mode == 1) hipLaunchKernelGGL(k_sweep<1>, dim3(1), dim3(64), 0, 0, reps, 1u, o, t, w);
      if (mode == 2) hipLaunchKernelGGL(k_sweep<2>, dim3(1), dim3(64), 0, 0, reps, 1u, o, t, w);
      if (mode == 3) hipLaunchKernelGGL(k_sweep<3>, dim3(1), dim3(64), 0, 0, reps, 1u, o, t, w);
    }
    sweep[mode] = (double)ticks() / reps - empty;
    printf("(1) %-74s %7.0f ticks = %6.3f us per tile, %5.1f ticks per pivot\n", names[mode], sweep[mode], sweep[mode] * ns_per_tick * 1e-3, sweep[mode] / 16.0);
  }
  printf("    chain only + multiply-adds only = %.0f against the full sweep's %.0f: %s\n", sweep[2] + sweep[3], sweep[0],
         sweep[2] + sweep[3] < 1.15 * sweep[0] ? "they ADD (nothing overlaps inside the wave)" : "they overlap");
  // ---- (2) instruction classes
  const char* inames[I_N] = {"v_fma_f64", "v_fmac_f64_dpp row_newbcast", "v_mul_f64", "v_rsq_f64", "s_nop 1 + v_mov_b64_dpp row_newbcast", "v_mfma_f64_16x16x4",
                             "v_cndmask_b32", "ds_bpermute_b32", "v_permlane16_swap_b32", "v_permlane32_swap_b32"};
  double dep[I_N], ind[I_N];
#define RUN_I(K)                                                                                  \
  hipLaunchKernelGGL((k_instr<K, 1>), dim3(1), dim3(64), 0, 0, o, t); hipLaunchKernelGGL((k_instr<K, 1>), dim3(1), dim3(64), 0, 0, o, t); \
  dep[K] = (double)ticks() / 512.0;                                                               \
  hipLaunchKernelGGL((k_instr<K, 8>), dim3(1), dim3(64), 0, 0, o, t); hipLaunchKernelGGL((k_instr<K, 8>), dim3(1), dim3(64), 0, 0, o, t); \
  ind[K] = (double)ticks() / 512.0;
  RUN_I(I_FMA) RUN_I(I_FMAC_DPP) RUN_I(I_MUL) RUN_I(I_RSQ) RUN_I(I_MOV_DPP) RUN_I(I_MFMA) RUN_I(I_CNDMASK) RUN_I(I_BPERMUTE) RUN_I(I_PERM16) RUN_I(I_PERM32)
  for (int k = 0; k < I_N; ++k) printf("(2) %-40s dependent chain %6.1f ticks per instruction, 8 independent accumulators %6.1f\n", inames[k], dep[k], ind[k]);
  {
    double mix[4];
    hipLaunchKernelGGL(k_mix<0>, dim3(1), dim3(64), 0, 0, o, t); hipLaunchKernelGGL(k_mix<0>, dim3(1), dim3(64), 0, 0, o, t); mix[0] = (double)ticks() / 256.0;
    hipLaunchKernelGGL(k_mix<1>, dim3(1), dim3(64), 0, 0, o, t); hipLaunchKernelGGL(k_mix<1>, dim3(1), dim3(64), 0, 0, o, t); mix[1] = (double)ticks() / 256.0;
    hipLaunchKernelGGL(k_mix<2>, dim3(1), dim3(64), 0, 0, o, t); hipLaunchKernelGGL(k_mix<2>, dim3(1), dim3(64), 0, 0, o, t); mix[2] = (double)ticks() / 256.0;
    hipLaunchKernelGGL(k_mix<4>, dim3(1), dim3(64), 0, 0, o, t); hipLaunchKernelGGL(k_mix<4>, dim3(1), dim3(64), 0, 0, o, t); mix[3] = (double)ticks() / 256.0;
    printf("(2) one dependent v_fma_f64 followed by K independent v_fmac_f64_dpp, ticks per group: K = 0: %.1f, 1: %.1f, 2: %.1f, 4: %.1f\n", mix[0], mix[1], mix[2], mix[3]);
  }
  // ---- (3) co-resident waves: waves w and w + 4 share SIMD w % 4
  struct { unsigned mask; const char* what; } co[] = {{0x01u, "wave 0 alone"}, {0x03u, "waves 0, 1 (two SIMDs)"}, {0x0fu, "waves 0-3 (four SIMDs)"},
                                                      {0x11u, "waves 0, 4 (both on SIMD 0)"}, {0xffu, "waves 0-7 (two per SIMD)"}};
  for (auto& c : co) {
    hipLaunchKernelGGL(k_sweep<0>, dim3(1), dim3(512), 0, 0, reps, c.mask, o, t, w);
    hipLaunchKernelGGL(k_sweep<0>, dim3(1), dim3(512), 0, 0, reps, c.mask, o, t, w);
    printf("(3) full sweep, %-30s %7.0f ticks per tile on wave 0\n", c.what, (double)ticks() / reps - empty);
  }
  // ---- (4) one panel step of a 4-wide blocked tile
  hipLaunchKernelGGL(k_blocked_panel, dim3(1), dim3(64), 0, 0, reps, o, t);
  hipLaunchKernelGGL(k_blocked_panel, dim3(1), dim3(64), 0, 0, reps, o, t);
  const double panel = (double)ticks() / reps;
  printf("(4) blocked panel step: operand selects + ONE v_mfma_f64_16x16x4 + all-gather of twelve columns over the four row groups: %.0f ticks\n", panel);
  printf("    it replaces the 4 x (12 - 4 p) sweep multiply-adds of panel p on the A rows: 48 / 32 / 16 v_fmac_f64_dpp = %.0f / %.0f / %.0f ticks at %.1f ticks each\n",
         48 * ind[I_FMAC_DPP], 32 * ind[I_FMAC_DPP], 16 * ind[I_FMAC_DPP], ind[I_FMAC_DPP]);
  printf("    bound of the blocked tile: 3 panel steps cost %.0f ticks and save %.0f -> %+.0f ticks = %+.3f us per tile (the E rows keep their 120 multiply-adds:\n"
         "    their operands live one extra row per LANE, four different tiles per wave -- four more matrix instructions per panel and a transpose each)\n",
         3 * panel, 96 * ind[I_FMAC_DPP], 3 * panel - 96 * ind[I_FMAC_DPP], (3 * panel - 96 * ind[I_FMAC_DPP]) * ns_per_tick * 1e-3);
  return 0;
}
