// ppo_chol.hip -- native FP64 Cholesky solve of the dense reduced system (replaces LinearSolverDense /
// Eigen::LDLT of Thirdparty/g2o/g2o/solvers/linear_solver_dense.h:65-113 on the device).
//
// Why not rocSOLVER: at n ~ 1.6k dpotrf + dpotrs spend 5 ms in latency-bound single-workgroup kernels
// (profiles/r01_v1_kernel_stats.txt).  This solver is built for that regime:
//   * right-looking, 64-wide panels, the matrix padded to a multiple of 64 (pad diagonal = 1);
//   * chol_step(k), ONE launch per panel: the "column" workgroups first apply the rank-64 update of panel k-1 to the diagonal
//     block and to their own block (look-ahead by one), then every one of them factors the 64x64 diagonal block redundantly in
//     LDS -- four 16x16 wave-level register factorisations whose broadcast-multiply-adds are single f64 DPP instructions
//     (v_fmac_f64 row_newbcast), the rows below riding along in the other lanes, and v_mfma_f64_16x16x4 for every 16x16x16
//     block update -- and solves its own 64-row block against it; the right-hand side rides along as a 65th row, so the
//     forward substitution costs nothing extra.  The remaining workgroups of the launch apply panel k-1's update to one
//     64x64 tile of the trailing matrix each (64 FP64 MFMAs per wave);
//   * chol_back_all: the backward substitution in one launch, one workgroup per block, pipelined through data-tagged
//     8-byte granules (no fences, no grid barrier); a link of the chain is a matrix-vector product with explicit inverses
//     (T_k = L_kk^-1, M_k = L_{k+1,k} T_k) that one extra workgroup per panel launch prepares (inv_block_wg);
//   * overlapped reduction (chol_factor_solve_overlapped, StepOverlap): the panel launches also carry the pose-pair chunks of
//     the Schur complement for the column blocks they have not reached yet (ppo_pairs.hpp) -- DESIGN.md 4.6;
//   * deterministic mode (chol_factor_solve_fixed, StepFix; the library's default since round 6): the system arrives as 2^-24 fixed-point integers in a second
//     array; open column blocks collect the factorisation's own trailing updates there too, the launch that finds a block final converts it with one load,
//     the factor's array is never cleared -- DESIGN.md 4.10;
//   * n/64 + 1 launches per solve (optionally captured once into a hipGraph and replayed, PPO_BA_CHOL_GRAPH=1).
// FMAs are written explicitly (__builtin_fma): the library is built with -ffp-contract=off for the numeric Jacobians.
// info follows LAPACK potrf: 0 = positive definite, j > 0 = leading minor j is not (the LM loop treats that as
// "solver failed", levenberg.cpp:126-127).
#include "ppo_chol.hpp"
#include "ppo_dev.hpp"
#include "ppo_pairs.hpp"
#include "ppo_dpp.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>

#include "../../include/ppo_ba.h"

namespace ppo {
namespace {

constexpr int NB = 64;  // panel width
constexpr int LP = 65;  // padded LDS row length (doubles): column walks are bank-conflict free
constexpr int SB = 16;  // sub-block of the diagonal factorization = MFMA tile edge

typedef double v4d __attribute__((ext_vector_type(4)));
// 16-byte pairs as a native vector type: copies of HIP's double2 STRUCT are memcpys to the optimiser, and register arrays of them that
// are filled from memory and emptied into LDS were left in scratch memory once chol_step grew
typedef double d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ d2 d2v(double x, double y) { d2 v; v.x = x; v.y = y; return v; }
constexpr unsigned long long kSentinel = 0xFFF8C0DEC0DE0001ull;  // "not written yet": a NaN payload no arithmetic produces
constexpr int kSpinCap = 400000;

#ifndef STAMP1
#define STAMP1(i)  // same, taken by wave 1
#endif
#ifndef STAMPB
#define STAMPB(i)  // same, inside chol_back_all (workgroup of block 1)
#endif
#ifndef STAMP
#define STAMP(i)  // tools/chol_probe.hip defines it to record wall-clock stamps
#endif
#ifndef STAMPF
#define STAMPF(i)  // same, inside chol_fwd_all (the head of one block row)
#endif
#ifndef STAMPG
#define STAMPG(i)  // same, every workgroup of chol_fwd_all (slot i of its row in the probe's table)
#endif

__device__ __forceinline__ double bcast(double v, int lane) {  // lane is wave-uniform (a constant after unrolling)
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, lane);
  hi = __builtin_amdgcn_readlane(hi, lane);
  return __hiloint2double(hi, lo);
}

// 64x64 tile at `base` (column-major, leading dimension ld) -> registers: thread t holds 8 d2, element pair
// p = t + 256 i covers column p >> 5, rows 2 (p & 31) and +1.  All 8 loads are issued before any use.
__device__ __forceinline__ void load_tile(const double* __restrict__ base, int ld, d2 (&t)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int p = threadIdx.x + 256 * i;
    t[i] = *reinterpret_cast<const d2*>(&base[(size_t)(p >> 5) * ld + 2 * (p & 31)]);
  }
}
// the same tile spread over NT threads (chol_step: the four compute waves plus four loader waves)
template <int NT>
__device__ __forceinline__ void load_tile_nt(const double* __restrict__ base, int ld, d2 (&t)[2048 / NT]) {
#pragma unroll
  for (int i = 0; i < 2048 / NT; ++i) {
    const int p = threadIdx.x + NT * i;
    t[i] = *reinterpret_cast<const d2*>(&base[(size_t)(p >> 5) * ld + 2 * (p & 31)]);
  }
}

// All MFMA helpers below fetch EVERY LDS operand of a job (and the old values of the output tile) into registers before
// the first MFMA: left to itself the compiler emitted ds_read -> s_waitcnt lgkmcnt(0) -> 2 MFMAs per k-step, i.e. one full
// LDS round trip in front of every pair of matrix instructions, and a read-wait-subtract-write chain per output element.
// Output tiles are written whole, with a select where only the lower triangle may change (no exec-mask branches).

// C[16x16] (rows r0.., cols c0.. of M) -= X[16x16] * Y[16x16]^T where X = rows xr.. / Y = rows yr.. of columns [o, o+16)
// in the padded LDS arrays MX / MY.  One wave, 4 MFMAs.  D layout of v_mfma_f64_16x16x4: col = lane & 15, row = (lane >> 4) + 4 reg.
__device__ __forceinline__ void mfma_sub_16(double (*M)[LP], int r0, int c0, const double (*MX)[LP], int xr, const double (*MY)[LP], int yr, int o) {
  const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
  double a[4], b[4], c[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    a[m] = MX[xr + li][o + 4 * m + lk];  // A[i = li][k = lk]
    b[m] = MY[yr + li][o + 4 * m + lk];  // B[k = lk][j = li]
    c[m] = M[r0 + lk + 4 * m][c0 + li];
  }
  v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int m = 0; m < 4; ++m) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[m], acc, 0, 0, 0);
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) M[r0 + lk + 4 * reg][c0 + li] = c[reg] - acc[reg];
}

// C[16x16] = X[16x16] * T[16x16] written to rows r0.., cols c0.. of M, where X = rows xr.., columns [xo, xo+16) of MX and
// T is a 16x16 block stored [k][j] with row length 17.  In-place use (M == MX, same tile) is safe: the wave reads all of X
// before it writes.  With T = L^-T this is the triangular solve X L^T = B.
__device__ __forceinline__ void mfma_mul_16(double (*M)[LP], int r0, int c0, const double (*MX)[LP], int xr, int xo, const double (*T)[17]) {
  const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
  double a[4], b[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    a[m] = MX[xr + li][xo + 4 * m + lk];  // A[i = li][k = lk]
    b[m] = T[4 * m + lk][li];             // B[k = lk][j = li]
  }
  v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int m = 0; m < 4; ++m) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[m], acc, 0, 0, 0);
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) M[r0 + lk + 4 * reg][c0 + li] = acc[reg];
}

// (f64 DPP helpers fmac_bcast / row_bcast: ppo_dpp.hpp)

template <int J, int C>
struct ColUpdate {
  static __device__ __forceinline__ void run(double (&a)[SB], double (&e)[SB], double l, double le) {
    fmac_bcast<C, C == J + 1>(a[C], l, l);
#if !(defined(PPO_BA_TEST_HOOKS) && defined(PPO_CHOL_EXPERIMENT_NO_E))  // (timing experiment of tools/run_chol_experiments.sh: wrong results, the sweep without its extra rows)
    fmac_bcast<C, false>(e[C], l, le);
#endif
    if constexpr (C + 1 < SB) ColUpdate<J, C + 1>::run(a, e, l, le);
  }
};

template <int J>
struct ColSweep {
  static __device__ __forceinline__ void run(double (&a)[SB], double (&e)[SB], int& bad) {
    const double d = row_bcast<J>(a[J]);
    if (!(d > 0.0) && bad == 0) bad = J + 1;
    // 1/sqrt(d): v_rsq_f64 (measured 5e-8 relative) + ONE third-order correction: with e = 1 - d y0^2 (|e| ~ 1e-7),
    // 1/sqrt(d) = y0 (1 - e)^-1/2 = y0 (1 + e (1/2 + 3/8 e)) + O(e^3): five dependent operations instead of the six of two
    // Newton steps (this chain is the serial core of the whole solve), full double accuracy (2.2e-16 measured)
    const double y0 = __builtin_amdgcn_rsq(d);
    const double e1 = __builtin_fma(-(d * y0), y0, 1.0);
    const double y = __builtin_fma(y0, e1 * __builtin_fma(0.375, e1, 0.5), y0);
    const double l = a[J] * y, le = e[J] * y;
    a[J] = l;
    e[J] = le;
    if constexpr (J + 1 < SB) {
      ColUpdate<J, J + 1>::run(a, e, l, le);
      ColSweep<J + 1>::run(a, e, bad);
    }
  }
};

// One wave factors the 16x16 block sL[o..o+16)[o..o+16) in place (lower) AND applies the same column operations to 48 more
// rows.  Every group of 16 lanes (a DPP row) holds a copy of the 16 block rows, one row per lane, in registers (a[]); next
// to it each lane carries one extra row (e[]): group 0 the rows of an identity matrix -- the Cholesky elimination is a
// right-multiplication by L^-T, so they end up as L^-T: sT[i][c] = (L^-T)[i][c] -- and groups 1..3 the rows of the 16x16
// tiles below the block inside the 64x64 panel block, which come out as X = B L^-T (the triangular solve, for free).
// Returns the 1-based local index of the first non-positive pivot, 0 if none.
__device__ __forceinline__ int potf2_16(double (*sL)[LP], double (*sT)[17], const double (*sId)[SB], int o, bool store) {
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  const int xrow = o + SB * g + r;  // extra row of groups 1..3
  const bool has_x = g > 0 && xrow < NB;
  // Every lane reads its two rows through one base address each -- the extra row of group 0 comes from an identity block
  // kept in LDS, a lane without a row reads a zero row -- and writes them back through one base address each, whole rows,
  // in two predicated regions.  (Per-element selects and `if (c <= r)` stores made this part, not the sweep, the larger
  // half of the routine: 1.0 us of loads + stores around a 1.0 us sweep.  The upper triangle of the diagonal block is
  // written with whatever the sweep left there: nothing reads it.)
  double* pa = &sL[o + r][o];
  const double* pe = g == 0 ? &sId[r][0] : (has_x ? &sL[xrow][o] : &sId[SB][0]);
  double a[SB], e[SB];
#pragma unroll
  for (int c = 0; c < SB; ++c) {
    a[c] = pa[c];
    e[c] = pe[c];
  }
  int bad = 0;
  if (o == SB) STAMP(40);  // (tools/chol_probe.hip: the second of the four register factorisations, taken apart)
  ColSweep<0>::run(a, e, bad);
  if (o == SB) STAMP(41);
  if (!store) return bad;
  if (g == 0) {
#pragma unroll
    for (int c = 0; c < SB; ++c) pa[c] = a[c];
  }
  if (g == 0 || has_x) {
    double* qe = g == 0 ? &sT[r][0] : &sL[xrow][o];
#pragma unroll
    for (int c = 0; c < SB; ++c) qe[c] = e[c];
  }
  if (o == SB) STAMP(42);
  return bad;
}

// x <- L^-T v for a factored 64x64 block, executed by one wave.  Lane i works on the pre-scaled unknown w_i = v_i / L_ii
// and holds, in registers, column i of L scaled the same way (lcol[c] = L[c][i] / L_ii for c > i, 0 otherwise -- loaded
// before the data arrives).  The sweep from the last unknown to the first is then one broadcast + one FMA per step with
// no memory access, multiply or select inside the dependency chain (with LDS reads inside it the sweep took 3.8 us).
__device__ __forceinline__ void bwd_load_64(const double (*sL)[LP], double (&lcol)[NB], double& inv) {
  const int lane = threadIdx.x & 63;
  inv = 1.0 / sL[lane][lane];
#pragma unroll
  for (int c = 0; c < NB; ++c) lcol[c] = (c > lane) ? sL[c][lane] * inv : 0.0;
}
__device__ __forceinline__ double bwd_vec_64(double v, const double (&lcol)[NB], double inv) {
  double w = v * inv;
#pragma unroll
  for (int c = NB - 1; c > 0; --c) w = __builtin_fma(-lcol[c], bcast(w, c), w);
  return w;
}

constexpr int UP = NB + 4;  // row length of the update operand tiles in LDS

// The wave's 32x32 quadrant of X Y^T for two 64x64 operand tiles held k-major in LDS (sX[m][row]); computed transposed
// (A operand = Y rows = columns of the result, B operand = X rows) so that in the accumulator 16 lanes hold 16 consecutive
// rows: acc[jc][ir][reg] = element (row wr + 16 ir + li, column wc + 16 jc + lk + 4 reg).
__device__ __forceinline__ void quadrant_product(const double (*sX)[UP], const double (*sY)[UP], int wr, int wc, int li, int lk, v4d (&acc)[2][2]) {
#pragma unroll 4
  for (int m0 = 0; m0 < NB; m0 += 4) {
    const double q0 = sY[m0 + lk][wc + li], q1 = sY[m0 + lk][wc + 16 + li];
    const double p0 = sX[m0 + lk][wr + li], p1 = sX[m0 + lk][wr + 16 + li];
    acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(q0, p0, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(q0, p1, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(q1, p0, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(q1, p1, acc[1][1], 0, 0, 0);
  }
}

// One wave, two 16x16 tiles at once: M[16 t.., 16 u..] -= X_t Y_u^T over the whole previous panel (K = 64), X / Y k-major
// in LDS; 32 MFMAs (v_mfma_f64_16x16x4 issues every 64 cycles whatever the chaining).  lo1/lo2: only the lower triangle of
// that tile may change.
__device__ __forceinline__ void tile_update2(double (*M)[LP], int t1, int u1, bool lo1, int t2, int u2, bool lo2, const double (*sX)[UP],
                                             const double (*sY)[UP]) {
  const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
  double a1[16], b1[16], a2[16], b2[16], c1[4], c2[4];
#pragma unroll
  for (int m = 0; m < 16; ++m) {
    a1[m] = sX[4 * m + lk][t1 * SB + li];
    b1[m] = sY[4 * m + lk][u1 * SB + li];
    a2[m] = sX[4 * m + lk][t2 * SB + li];
    b2[m] = sY[4 * m + lk][u2 * SB + li];
  }
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    c1[reg] = M[t1 * SB + lk + 4 * reg][u1 * SB + li];
    c2[reg] = M[t2 * SB + lk + 4 * reg][u2 * SB + li];
  }
  v4d acc1 = {0.0, 0.0, 0.0, 0.0}, acc2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int m = 0; m < 16; ++m) {
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[m], b1[m], acc1, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2[m], b2[m], acc2, 0, 0, 0);
  }
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    const int r1 = t1 * SB + lk + 4 * reg, q1 = u1 * SB + li, r2 = t2 * SB + lk + 4 * reg, q2 = u2 * SB + li;
    M[r1][q1] = (!lo1 || r1 >= q1) ? c1[reg] - acc1[reg] : c1[reg];
    M[r2][q2] = (!lo2 || r2 >= q2) ? c2[reg] - acc2[reg] : c2[reg];
  }
}

// One tile (16 MFMAs), used where a wave has a single tile to do.
__device__ __forceinline__ void tile_update1(double (*M)[LP], int t, int u, bool lo, const double (*sX)[UP], const double (*sY)[UP]) {
  const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
  double a[16], b[16], c[4];
#pragma unroll
  for (int m = 0; m < 16; ++m) {
    a[m] = sX[4 * m + lk][t * SB + li];
    b[m] = sY[4 * m + lk][u * SB + li];
  }
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) c[reg] = M[t * SB + lk + 4 * reg][u * SB + li];
  v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int m = 0; m < 16; ++m) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[m], acc, 0, 0, 0);
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    const int r = t * SB + lk + 4 * reg, q = u * SB + li;
    M[r][q] = (!lo || r >= q) ? c[reg] - acc[reg] : c[reg];
  }
}

__device__ __forceinline__ void wave_lds_fence() {  // make this wave's LDS writes visible to its own other lanes
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Block substitution step KK for the 16 rows rt of B (X = B L^-T): tile (rt, KK) <- tile * L_KK^-T, then the tiles to its
// right take their update.  One wave; the halves meet through LDS, hence the wave-level fences.
template <int KK>
__device__ __forceinline__ void solve_rows(double (*sB)[LP], int rt, const double (*sL)[LP], const double (*sT)[SB][17]) {
  constexpr int o = KK * SB, NR = NB / SB - 1 - KK;
  const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
  wave_lds_fence();
  // operands that do not depend on the multiplication: L tiles below the diagonal block and the old values of the targets
  double lb[NR > 0 ? NR : 1][4], cold[NR > 0 ? NR : 1][4];
#pragma unroll
  for (int q = 0; q < NR; ++q)
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      lb[q][m] = sL[(KK + 1 + q) * SB + li][o + 4 * m + lk];
      cold[q][m] = sB[rt * SB + lk + 4 * m][(KK + 1 + q) * SB + li];
    }
  mfma_mul_16(sB, rt * SB, o, sB, rt * SB, o, sT[KK]);
  wave_lds_fence();
  if constexpr (NR > 0) {
    double a[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) a[m] = sB[rt * SB + li][o + 4 * m + lk];
    v4d acc[NR] = {};
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int q = 0; q < NR; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], lb[q][m], acc[q], 0, 0, 0);
#pragma unroll
    for (int q = 0; q < NR; ++q)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) sB[rt * SB + lk + 4 * reg][(KK + 1 + q) * SB + li] = cold[q][reg] - acc[q][reg];
  }
}

// The right-hand-side row (row NB of sB) through substitution step kk: x = b T (16 lanes), then b[rest] -= x L[rest]^T.
__device__ __forceinline__ void solve_rhs_row(double (*sB)[LP], int kk, const double (*sL)[LP], const double (*sT)[SB][17]) {
  const int lane = threadIdx.x & 63, o = kk * SB;
  double xv = 0.0;
  wave_lds_fence();
  if (lane < SB) {
#pragma unroll
    for (int mm = 0; mm < SB; ++mm) xv = __builtin_fma(sB[NB][o + mm], sT[kk][mm][lane], xv);
  }
  __builtin_amdgcn_wave_barrier();
  if (lane < SB) sB[NB][o + lane] = xv;
  wave_lds_fence();
  if (lane < NB - o - SB) {
    const int c = o + SB + lane;
    double sacc = 0.0;
#pragma unroll
    for (int mm = 0; mm < SB; ++mm) sacc = __builtin_fma(sB[NB][o + mm], sL[c][o + mm], sacc);
    sB[NB][c] -= sacc;
  }
  wave_lds_fence();
}

// ------------------------------------------------------------------------------------------------------------------
// chol_step(k): ONE launch per 64-wide panel.  With m = (ld - k0) / 64 block rows left, the grid is
//   * 1 + 2 (m - 1) "column" workgroups: workgroup 0 owns the diagonal block (k,k); each block row below is split into two
//     32-row halves, one workgroup each.  They apply the rank-64 update of the PREVIOUS panel (k0 - 64) to the diagonal block
//     and to their own half block (look-ahead by one), factor the diagonal block (every workgroup redundantly, in LDS), solve
//     their half against it, and carry the right-hand side along as a 65th row;
//   * m (m - 1) / 2 "trailing" workgroups: apply the previous panel's update to one 64x64 tile (i >= j > k) each.
// The serial part is the chain of four 16x16 register factorisations on wave 0 (~1.8 us each).  Everything else a column
// workgroup has to do -- the rest of the diagonal block's update, the update of its own rows and the block substitution
// -- is MFMA work cut into 16-MFMA jobs and run by waves 1..3 in the shadow of those factorisations (r0, r1 = the two
// 16-row tiles of the own half):
//                 wave 1                           wave 2                           wave 3
//   before    D(w,0) on wave w (all four): block column 0 of the diagonal block, what potf2(0) needs
//   potf2(0)  D(1,1)  U(r0,0)                  D(2,1)  U(r0,1)                  D(3,1)  U(r0,2)
//   potf2(1)  U(r0,3)  S0(r0)                  D(2,2)  U(r1,0)                  D(3,2)  U(r1,1)
//   potf2(2)  U(r1,2)  S1(r0)                  D(3,3)  U(r1,3)                  R0  R1
//   potf2(3)  S0(r1) S1(r1) S2(r1)             S2(r0)                           R2
//   after     S3(r0) on wave 0, S3(r1) on wave 1, R3 on wave 3
// D = panel update of a diagonal-block tile, U = of an own-block tile, Ss = substitution step s of an own 16-row tile,
// Rs = step s of the right-hand-side row.  v_mfma_f64_16x16x4 issues every 64 cycles whatever the chaining, and a side
// wave reaches ~60 % of that with its LDS traffic: with whole 64-row blocks per workgroup (<= 48 MFMAs per wave and phase)
// the phases were bound by the side waves (2.3 us); with halves (<= 36) they are bound by the factorisation again
// (1.8 us), and the kernel went from 16 to 13.5 us per panel at n = 1216.
// ------------------------------------------------------------------------------------------------------------------
// Launched with kStepThreads = 512 threads: waves 4..7 are LOADER waves.  A workgroup pulls 64-96 KB through one CU before it can
// start, and that prologue is bound by the requests a wave may have in flight, not by bandwidth (4.6 us of a 13.5 us panel step
// with four waves issuing 24 requests each); the loader waves issue half of the requests, write their share into LDS, meet the
// compute waves at the first barrier and end (an ended wave no longer counts at later barriers).
#ifndef PPO_CHOL_STEP_THREADS
#define PPO_CHOL_STEP_THREADS 512
#endif
constexpr int kStepThreads = PPO_CHOL_STEP_THREADS;  // (256: no loader waves -- A/B builds)
// The LDS of one workgroup of chol_step: the column / trailing workgroups' tiles, or -- the pair workgroups of an overlapped reduction -- the
// transpose-sum buffers of the pair body (one allocation per launch: 147 KB, i.e. one workgroup per CU whatever its part)
struct StepLds {
  double sL[NB][LP];
  double sB[NB + 1][LP];  // row NB carries r_k^T: the forward substitution is one more row of the block solve
  double sT[NB / SB][SB][17];
  __attribute__((aligned(16))) double sP[NB][UP];  // [m][row of block i] of the previous panel
  __attribute__((aligned(16))) double sQ[NB][UP];  // [m][row of block j]
  double sId[SB + 1][SB];  // identity rows + one zero row: the "extra rows" of lanes that carry none
  int sBad;
};
union StepLdsU {
  StepLds a;
  double pairs[kStepThreads / 64][21][65];
};
// ---- overlapped reduction ---------------------------------------------------------------------------------------------------------
// The Schur pair reduction (ppo_pairs.hpp, 42 us at the BASELINE size) fills the lower triangle of S before the factorisation can start --
// but panel k only READS column block k, and while the column workgroups walk their serial chain (13 us per panel) 200 CUs have nothing
// to do.  With the pair chunks ordered by column block, only the chunks of the first fin[0] column blocks are reduced ahead of the
// factorisation; launch k of chol_step carries, as extra workgroups right behind its column workgroups, the chunks of the column blocks
// [fin[k], fin[k + 1]).  fin[k] = number of column blocks that are FINAL (hold all their pair contributions) when launch k starts; the
// schedule keeps fin[k] >= k + 1, so the column workgroups of launch k only ever read final blocks.  The trailing update of a tile and
// the pair contributions to it are both sums, so their order does not matter -- but a tile the pair workgroups of THIS launch may be
// adding to must take its update as atomic adds too (column blocks fin[k] .. fin[k + 1]: the last one because a 6-column pose block may
// straddle a block boundary); every other tile keeps the plain read-modify-write.  The right-hand side receives pair contributions as
// well: it is solved in place, its row updates are atomic adds.
struct StepOverlap {
  PairCtx ctx;
  PairChunks pc;
  int chunk_lo, chunk_hi;    // chunks the pair workgroups of this launch reduce
  int npair_wg;              // their number (kStepThreads / 64 chunks each)
  int atomic_lo, atomic_hi;  // trailing tiles of the column blocks [atomic_lo, atomic_hi] add their update atomically (none: lo > hi)
};
__host__ __device__ inline StepOverlap step_no_overlap(int nb) {
  StepOverlap o{};
  o.atomic_lo = 1;
  o.atomic_hi = 0;
  return o;
}

// ---- deterministic mode (options.deterministic): the reduction's sums are INTEGERS ---------------------------------------------------------
// Every sum of the reduction that crosses a wave (assembly of H_pp + lambda I, the pair chunks, the cuboid terms) is an integer atomic on a 2^-24
// fixed-point image of the addend (ppo_kernels.hpp: fixed_add), so its result does not depend on the order the atomics arrive in.  Those sums live
// in a SECOND array of the shape of A (Afix, with bfix behind it).  A column block is OPEN for launch k while pair workgroups may still add to it
// (block >= fin[k]): the trailing update of an open tile is added to Afix as well -- the factorisation's own terms rounded to the same 2^-24 grid,
// integer atomics, still exact and order-free -- so that everything an open block has received lies in ONE place.  The first launch that finds the
// block final FOLDS it: the workgroup that owns a tile takes conv(Afix) as the tile's value (it never reads A: A needs no clearing between solves),
// applies its update and stores doubles into A; from then on the block lives in A like in the floating-point mode.  One load per tile in every launch,
// whatever the schedule.  (A first version kept the trailing updates of open tiles in A and added conv(Afix) on top at the fold: a launch that folded
// its own column block then requested both arrays in the prologue of the chain -- every launch of a schedule that runs at its minimum fin[k] = k + 1,
// 7 of the 10 launches at the 100-keyframe BASELINE configuration: 4 526 against 4 750 LM iterations/s in the floating-point mode.)
struct StepFix {
  const long long* Afix;  // [ld][ld] column-major, lower triangle; NULL: floating-point accumulation (nothing below applies).  (The kernel takes both
  const long long* bfix;  // [ld]  arrays through its preloaded `rhs_in` argument -- bfix == Afix + ld * ld --, these two members document the launch.)
  double* detbad;         // raised by a pair workgroup whose addend does not fit the fixed-point range
  int fold_lo, fold_hi;   // column blocks [fold_lo, fold_hi) are folded by this launch; blocks >= fold_hi are open (updates go to Afix / bfix), blocks < fold_lo live in A
};
constexpr double kStepFixScale = 16777216.0, kStepFixInv = 1.0 / 16777216.0;  // (= kFixScale / kFixInv of ppo_kernels.hpp)
// (v * 2^-24 rounded once, as (double)v * 2^-24 is -- written as hi * 2^8 + lo * 2^-24 in ONE fused multiply-add: four instructions instead of the five
//  the compiler makes of the cast, in the prologue of the chain)
__device__ __forceinline__ double fix2d(long long v) { return __builtin_fma((double)(int)(v >> 32), 256.0, (double)(unsigned)v * kStepFixInv); }
__device__ __forceinline__ d2 fix2d2(const long long* p) {
  const longlong2 v = *reinterpret_cast<const longlong2*>(p);
  return d2v(fix2d(v.x), fix2d(v.y));
}

// (not inlined: as part of chol_step's body the pair code -- 234 VGPRs, 60 scalar registers of keyframe data -- pushed the column
//  workgroups' prologue into scratch memory)
__device__ __forceinline__ void step_pairs(const StepOverlap& ov, int bid, double (*s_buf)[21][65]) {
  schur_pairs_body<kStepThreads / 64, false>(ov.ctx, ov.pc, ov.chunk_lo, ov.chunk_hi, bid, ov.npair_wg, s_buf, [](double* p, double v) { atomicAdd(p, v); });
}
// deterministic mode: ov.ctx.S / ov.ctx.bs point at the fixed-point arrays
__device__ __forceinline__ void step_pairs_fixed(const StepOverlap& ov, double* detbad, int bid, double (*s_buf)[21][65]) {
  schur_pairs_body<kStepThreads / 64, false>(ov.ctx, ov.pc, ov.chunk_lo, ov.chunk_hi, bid, ov.npair_wg, s_buf, [detbad](double* p, double v) {
    if (!(fabs(v) < 4.0e11)) { *detbad = 1.0; return; }  // (ppo_kernels.hpp: fixed_add -- an entry of S or b that does not fit: ACC_DETRANGE, the system is not usable)
    atomicAdd(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double2ll_rn(v * kStepFixScale));
  });
}

// ---- explicit inverses for the backward substitution ---------------------------------------------------------------------------
// The backward substitution is a chain of n / 64 hand-offs (chol_back_all); with the diagonal block's factor L_kk alone every link is
// "subtract L_{k+1,k}^T x_{k+1}, then a 64-step triangular sweep" (1.0 us of arithmetic behind a 0.5 us hand-off).  Launch k + 1 of the
// forward phase therefore carries ONE extra workgroup that has the whole launch (13 us, 200 CUs idle) to form
//   T_k = L_kk^-1   and   M_k = L_{k+1,k} T_k ,
// after which a link of the chain is x_k = T_k^T (y_k - sum_{i > k+1} L_ik^T x_i) - M_k^T x_{k+1}: the first term is ready before x_{k+1}
// arrives, the second is ONE 64 x 64 matrix-vector product out of registers.  (Block n/64 - 1, the head of the chain, keeps the sweep.)
// T and M are stored row-major ([row of T / of block k+1][column = row of x_k]) so that lane r reads its column with coalesced loads.
constexpr size_t kInvDoubles = 2 * (size_t)NB * NB;  // per block: T then M
__device__ __forceinline__ void inv_block_wg(const double* __restrict__ A, int ld, int kp, double* __restrict__ inv_out, StepLds& L) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
  double (*const sL)[LP] = L.sL;   // L_kk (lower), row-major
  double (*const sTm)[UP] = L.sP;  // T, row-major ( = k-major operand of the product below)
  double (*const sL1)[UP] = L.sQ;  // L_{k+1,k}, k-major: sL1[m][row]
  double (*const sX)[LP] = L.sB;   // scratch of the block recursion
  double* const dinv = &L.sId[0][0];
  {
    d2 ta[4], tb[4];  // element pair p = tid + 512 i: column p >> 5, rows 2 (p & 31), +1
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int p = tid + 512 * i, c = p >> 5, r = 2 * (p & 31);
      ta[i] = *reinterpret_cast<const d2*>(&A[(size_t)(kp + c) * ld + kp + r]);
      tb[i] = *reinterpret_cast<const d2*>(&A[(size_t)(kp + c) * ld + kp + NB + r]);
    }
    for (int i = tid; i < NB * UP; i += 512) (&sTm[0][0])[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int p = tid + 512 * i, c = p >> 5, r = 2 * (p & 31);
      sL[r][c] = (r >= c) ? ta[i].x : 0.0;
      sL[r + 1][c] = (r + 1 >= c) ? ta[i].y : 0.0;
      *reinterpret_cast<d2*>(&sL1[c][r]) = tb[i];
    }
  }
  __syncthreads();
  if (tid < NB) dinv[tid] = 1.0 / sL[tid][tid];
  __syncthreads();
  // T = L_kk^-1 by block recursion (a lane-per-column substitution over all 64 rows is 2 016 dependent multiply-adds behind LDS reads:
  // 25 us, longer than the launch it is meant to hide in).
  // level 0: the four 16 x 16 diagonal blocks at once on one wave -- lane (b, c) solves column c of block b, 120 multiply-adds
  if (wave == 0) {
    const int o = 16 * (lane >> 4), c = lane & 15;
    double t[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      double a0 = (r == c) ? 1.0 : 0.0, a1 = 0.0;
#pragma unroll
      for (int m = 0; m + 1 < r; m += 2) {
        a0 = __builtin_fma(-sL[o + r][o + m], t[m], a0);
        a1 = __builtin_fma(-sL[o + r][o + m + 1], t[m + 1], a1);
      }
      if (r & 1) a0 = __builtin_fma(-sL[o + r][o + r - 1], t[r - 1], a0);
      t[r] = (a0 + a1) * dinv[o + r];
      sTm[o + r][o + c] = t[r];
    }
  }
  __syncthreads();
  // level 1: T(hi, lo) = -T(hi, hi) (L(hi, lo) T(lo, lo)) for the 16-blocks (1, 0) and (3, 2): one output per thread
  {
    const int hi = 16 * (2 * (tid >> 8) + 1), lo = hi - 16, rr = (tid >> 4) & 15, cc = tid & 15;
    double v = 0.0;
#pragma unroll
    for (int m = 0; m < 16; ++m) v = __builtin_fma(sL[hi + rr][lo + m], sTm[lo + m][lo + cc], v);
    sX[hi + rr][lo + cc] = v;
    __syncthreads();
    double w = 0.0;
#pragma unroll
    for (int m = 0; m < 16; ++m) w = __builtin_fma(sTm[hi + rr][hi + m], sX[hi + m][lo + cc], w);
    __syncthreads();
    sTm[hi + rr][lo + cc] = -w;
  }
  __syncthreads();
  // level 2: the 32 x 32 block T(1, 0) = -T(1, 1) (L(1, 0) T(0, 0)): two outputs per thread
  {
    const int rr = tid >> 5, cc = tid & 31;
    double v0 = 0.0, v1 = 0.0;
#pragma unroll 8
    for (int m = 0; m < 32; ++m) {
      const double tv = sTm[m][cc];
      v0 = __builtin_fma(sL[32 + rr][m], tv, v0);
      v1 = __builtin_fma(sL[48 + rr][m], tv, v1);
    }
    sX[32 + rr][cc] = v0;
    sX[48 + rr][cc] = v1;
    __syncthreads();
    double w0 = 0.0, w1 = 0.0;
#pragma unroll 8
    for (int m = 0; m < 32; ++m) {
      const double xv = sX[32 + m][cc];
      w0 = __builtin_fma(sTm[32 + rr][32 + m], xv, w0);
      w1 = __builtin_fma(sTm[48 + rr][32 + m], xv, w1);
    }
    __syncthreads();
    sTm[32 + rr][cc] = -w0;
    sTm[48 + rr][cc] = -w1;
  }
  __syncthreads();
  double* const To = inv_out;
  double* const Mo = inv_out + (size_t)NB * NB;
  if (wave >= 4) {  // T out (row-major), while waves 0..3 form M
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int e = (tid - 256) + 256 * i;
      To[e] = sTm[e >> 6][e & 63];
    }
    return;
  }
  // M = L_{k+1,k} T on the matrix cores, as M^T = T^T L_{k+1,k}^T: both operands are k-major as they lie ([m][r'] and [m][c']), and the
  // 16 consecutive lanes of an accumulator hold 16 consecutive r' of one row c' -- the row-major layout the backward substitution reads
  const int wr = (wave >> 1) * 32, wc = (wave & 1) * 32;
  v4d acc[2][2] = {};
  quadrant_product(sTm, sL1, wr, wc, li, lk, acc);
#pragma unroll
  for (int jc = 0; jc < 2; ++jc)
#pragma unroll
    for (int ir = 0; ir < 2; ++ir)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) Mo[(size_t)(wc + 16 * jc + lk + 4 * reg) * NB + wr + 16 * ir + li] = acc[jc][ir][reg];
}

// OV = false: the launch carries no pair workgroups and no tile takes atomic updates (ov is not read) -- the plain factorisation and the
// later launches of an overlapped one; as ONE kernel the pair code's registers (240 against 182) and arguments cost every launch 0.4 us.
// FX (deterministic mode, StepFix): 0 floating-point accumulation; 1 trailing tiles and right-hand-side rows are open / folded / in A by their column
// block (runtime ranges), pair workgroups add integers, the column workgroups' own block (k0 / 64) lives in A already; 3 the own column block is
// folded by THIS launch (launches 0 and 1 always; later ones when the schedule runs at its minimum fin[k - 1] == k): the diagonal tile, the own half
// and the block's right-hand-side rows are requested from the fixed-point arrays INSTEAD of A -- the same number of requests in the prologue of the
// chain as a plain step.
template <bool OV, int FX = 0>
__global__ __launch_bounds__(kStepThreads) void chol_step_t(const int* gate_word, int gate_expect, double* __restrict__ A, int ld, int k0, const double* rhs_in, double* rhs, double* xout,
                                                            int* __restrict__ info, int inplace_rhs, double* __restrict__ inv, StepOverlap ov, StepFix fx) {
  constexpr int NT = kStepThreads, NI = 2048 / NT;
  __shared__ StepLdsU lds;
  // deterministic mode: the fixed-point arrays arrive through `rhs_in` (unused there: the right-hand side starts from bfix) -- one of the kernel arguments
  // that are PRELOADED into scalar registers; as members of the StepFix aggregate at the end of the list they cost the column workgroups of a folding
  // launch a scalar-load round trip in front of their first tile request (measured +1.4 us on a launch whose chain is the launch: 14.5 against 13.2 us).
  // bfix lies right behind Afix (ld * ld entries), as b lies behind S.
  [[maybe_unused]] const long long* const fxA = FX != 0 ? reinterpret_cast<const long long*>(rhs_in) : nullptr;
  [[maybe_unused]] const long long* const fxb = FX != 0 ? fxA + (size_t)ld * ld : nullptr;
  // LM control on the device: the epoch word is REQUESTED here and looked at only where a workgroup is about to write -- for the column
  // workgroups (the chain) behind their tile requests, so that its latency hides behind theirs.  (Not gated: the word read is `info`, ignored.)
  // (two scalars in front of the argument list, not an LmGate: an aggregate argument ends the kernel-argument preload for everything behind it,
  //  which put a scalar-load round trip, 0.25 us, at the head of every panel launch)
  const int gate_seen = *(gate_word ? gate_word : const_cast<const int*>(info));
#define STEP_DEAD() (gate_word != nullptr && gate_seen != gate_expect)
  double (*const sL)[LP] = lds.a.sL;
  double (*const sB)[LP] = lds.a.sB;
  double (*const sT)[SB][17] = lds.a.sT;
  double (*const sP)[UP] = lds.a.sP;
  double (*const sQ)[UP] = lds.a.sQ;
  double (*const sId)[SB] = lds.a.sId;
  int& sBad = lds.a.sBad;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  const int m = (ld - k0) / NB;
  const int kp = k0 - NB;  // previous panel's column
  const bool upd = k0 > 0;
  STAMP(0);
  const int ncol = 1 + 2 * (m - 1);  // column workgroups of this launch
  const int ninv = (inv != nullptr && upd) ? 1 : 0;
  if (ninv && (int)blockIdx.x == ncol) {
    if (STEP_DEAD()) return;
    // ---- T and M of the PREVIOUS panel's block for the backward substitution (first behind the column workgroups: the longest of the rest)
    inv_block_wg(A, ld, kp, inv + (size_t)(kp / NB) * kInvDoubles, lds.a);
    return;
  }
  int npair_wg = 0;
  if constexpr (OV) npair_wg = ov.npair_wg;
  if (OV && (int)blockIdx.x >= ncol + ninv && (int)blockIdx.x < ncol + ninv + npair_wg) {
    // ---- pair workgroup of an overlapped reduction: S -= W_j (D_l + lambda I)^-1 W_i' for chunks of column blocks that are not final yet
    // (right behind the column workgroups in the grid: they are the long ones of the launch's shadow, the trailing tiles fill in)
    if (STEP_DEAD()) return;
    if constexpr (FX != 0) step_pairs_fixed(ov, fx.detbad, (int)blockIdx.x - ncol - ninv, lds.pairs);
    else step_pairs(ov, (int)blockIdx.x - ncol - ninv, lds.pairs);
    return;
  }
  if ((int)blockIdx.x >= ncol + ninv) {
    // ---- trailing tile (bi >= bj >= 1 relative to block k) --------------------------------------------------------------
    if (STEP_DEAD()) return;
    const int wr = (wave >> 1) * 32, wc = (wave & 1) * 32;
    int t = blockIdx.x - ncol - ninv - npair_wg, bi = 0;
    while ((bi + 1) * (bi + 2) / 2 <= t) ++bi;
    const int bj = t - bi * (bi + 1) / 2;
    const int r0 = k0 + NB + bi * NB, c0 = k0 + NB + bj * NB;
    const int cblk = c0 / NB;  // column block of the tile
    bool atomic_out = false;  // pair workgroups of this launch may be adding to it
    if constexpr (OV && FX == 0) atomic_out = cblk >= ov.atomic_lo && cblk <= ov.atomic_hi;  // (FX: they add to Afix, A is this workgroup's alone)
    d2 tp[NI], tq[NI];
    load_tile_nt<NT>(&A[(size_t)kp * ld + r0], ld, tp);
    load_tile_nt<NT>(&A[(size_t)kp * ld + c0], ld, tq);
    double cold[2][2][4] = {};  // old values of the output quadrant, fetched while the panel tiles land (compute waves)
    [[maybe_unused]] bool open_out = false;  // deterministic mode: the tile's column block is still open -- its update goes to Afix as integers
    if constexpr (FX != 0) {
      open_out = cblk >= fx.fold_hi;
      const bool fold = !open_out && cblk >= fx.fold_lo;
      if (tid < 256 && !open_out) {  // a folded tile starts from conv(Afix) (A is not read: nothing of the previous solve survives), a tile in A from A
        // (ONE stream of sixteen requests from a base chosen once -- written as `fold ? fix2d(Afix[at]) : A[at]` the compiler made it sixteen branches, each
        //  with its own load and its own wait: sixteen memory round trips in a row, 14.4 us for a launch whose chain takes 13.2)
        const long long* const src = fold ? fxA : reinterpret_cast<const long long*>(A);
        long long raw[2][2][4];
#pragma unroll
        for (int jc = 0; jc < 2; ++jc)
#pragma unroll
          for (int ir = 0; ir < 2; ++ir)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) raw[jc][ir][reg] = src[(size_t)(c0 + wc + 16 * jc + lk + 4 * reg) * ld + r0 + wr + 16 * ir + li];
#pragma unroll
        for (int jc = 0; jc < 2; ++jc)
#pragma unroll
          for (int ir = 0; ir < 2; ++ir)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) cold[jc][ir][reg] = fold ? fix2d(raw[jc][ir][reg]) : __longlong_as_double(raw[jc][ir][reg]);
      }
    } else
    if (tid < 256 && !atomic_out) {
#pragma unroll
      for (int jc = 0; jc < 2; ++jc)
#pragma unroll
        for (int ir = 0; ir < 2; ++ir)
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) cold[jc][ir][reg] = A[(size_t)(c0 + wc + 16 * jc + lk + 4 * reg) * ld + r0 + wr + 16 * ir + li];
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int p = tid + NT * i, mm = p >> 5, r = 2 * (p & 31);
      *reinterpret_cast<d2*>(&sP[mm][r]) = tp[i];
      *reinterpret_cast<d2*>(&sQ[mm][r]) = tq[i];
    }
    __syncthreads();
    if (tid >= 256) return;  // loader waves
    v4d acc[2][2] = {};
    quadrant_product(sP, sQ, wr, wc, li, lk, acc);
#if defined(PPO_BA_TEST_HOOKS) && defined(PPO_CHOL_EXPERIMENT_PAIR_BOUND)
    quadrant_product(sP, sQ, wr, wc, li, lk, acc);  // (a pair launch's trailing tiles take a rank-128 update)
#endif
    if constexpr (FX != 0) {
      if (open_out) {  // (the same 2^-24 grid as every other term of the open block: exact integer adds, any order)
        unsigned long long* const Ai = reinterpret_cast<unsigned long long*>(const_cast<long long*>(fxA));
#pragma unroll
        for (int jc = 0; jc < 2; ++jc)
#pragma unroll
          for (int ir = 0; ir < 2; ++ir)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
              const int c = wc + 16 * jc + lk + 4 * reg, r = wr + 16 * ir + li;
              const double v = -acc[jc][ir][reg];
              if (bi != bj || r >= c) {
                if (!(fabs(v) < 4.0e11)) *fx.detbad = 1.0;
                else atomicAdd(&Ai[(size_t)(c0 + c) * ld + r0 + r], (unsigned long long)__double2ll_rn(v * kStepFixScale));
              }
            }
        return;
      }
    }
    if (atomic_out) {  // (16 lanes -> 16 consecutive rows of one column: one cache line per quarter wave)
#pragma unroll
      for (int jc = 0; jc < 2; ++jc)
#pragma unroll
        for (int ir = 0; ir < 2; ++ir)
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            const int c = wc + 16 * jc + lk + 4 * reg, r = wr + 16 * ir + li;
            if (bi != bj || r >= c) atomicAdd(&A[(size_t)(c0 + c) * ld + r0 + r], -acc[jc][ir][reg]);
          }
      return;
    }
#pragma unroll
    for (int jc = 0; jc < 2; ++jc)
#pragma unroll
      for (int ir = 0; ir < 2; ++ir)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int c = wc + 16 * jc + lk + 4 * reg, r = wr + 16 * ir + li;
          if (bi != bj || r >= c) A[(size_t)(c0 + c) * ld + r0 + r] = cold[jc][ir][reg] - acc[jc][ir][reg];
        }
    return;
  }
  STAMP(7);
#if defined(PPO_BA_TEST_HOOKS) && defined(PPO_CHOL_EXPERIMENT_NO_LOOKAHEAD)
  const bool updc = false;  // (timing experiment, WRONG results: a step that finds final tiles -- no previous-panel rows to load, no look-ahead update)
#else
  const bool updc = upd;
#endif
  // ---- column workgroup: workgroup 0 owns the diagonal block, workgroups 1 + 2 (b - 1) + h the row halves h = 0, 1 of block b ----
  const int cw = blockIdx.x;
  const int b = cw == 0 ? 0 : 1 + ((cw - 1) >> 1);
  const int half = cw == 0 ? 0 : ((cw - 1) & 1);
  const int r0t = 2 * half, r1t = 2 * half + 1;  // the two 16-row tiles of the own half
  const int row0 = k0 + b * NB;
  const bool own = b > 0;
  const bool uo = own && updc;
  // step 0 takes the right-hand side from rhs_in (the caller's b stays intact), later steps continue in rhs
  const double* rsrc = (k0 == 0 && FX == 0) ? rhs_in : rhs;  // (deterministic mode: launch 0 takes the right-hand side from bfix, see below)
  double rold = 0.0;  // the right-hand-side entry this thread updates at the end (threads 8 r of a row-half workgroup)
  [[maybe_unused]] bool rhs_open = false;  // (deterministic mode: its row block is still open -- the update goes to bfix as an integer)
  {
    // Everything is requested unconditionally and back to back (what is not needed aliases the diagonal tile): with the
    // loads under `if`s the compiler drained each group before issuing the next, four memory round trips in a row.
    // (round 3) The own half and its previous-panel rows (a third of the bytes) are requested LAST and stored into LDS only after the
    // column-0 update of the diagonal block: the critical tiles are waited for alone (memory operations of a wave complete in the order
    // they were issued), the others land while the matrix cores work on what the first factorisation needs.
    d2 td[NI], tq[NI], tb[NI / 2], tp[NI / 2];
    const double* pC = &A[(size_t)k0 * ld + k0];
    const double* pQ = updc ? &A[(size_t)kp * ld + k0] : pC;   // L[k][k-1]
    const double* pB = own ? &A[(size_t)k0 * ld + row0] : pC;
    const double* pP = uo ? &A[(size_t)kp * ld + row0] : pC;  // L[k+b][k-1]
    STAMP(15);
    [[maybe_unused]] longlong2 fd[NI], fb[NI / 2];  // (FX == 3: fixed-point images of the diagonal tile and of the own half, converted behind the requests)
    const long long* fC = FX == 3 ? &fxA[(size_t)k0 * ld + k0] : nullptr;
    const long long* fB = FX == 3 ? (own ? &fxA[(size_t)k0 * ld + row0] : fC) : nullptr;
    if constexpr (FX == 3) {
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int p = tid + NT * i;
        fd[i] = *reinterpret_cast<const longlong2*>(&fC[(size_t)(p >> 5) * ld + 2 * (p & 31)]);
      }
    } else {
      load_tile_nt<NT>(pC, ld, td);
    }
    STAMP(16);
    load_tile_nt<NT>(pQ, ld, tq);
    STAMP(17);
    double rv = (tid < NB) ? (FX == 3 ? fix2d(fxb[k0 + tid]) : rsrc[k0 + tid]) : 0.0;
#pragma unroll
    for (int i = 0; i < NI / 2; ++i) {  // the own half: 32 rows x 64 columns, 16 d2 per column
      const int p = tid + NT * i, c = p >> 4, r = 32 * half + 2 * (p & 15);
      if constexpr (FX == 3) fb[i] = *reinterpret_cast<const longlong2*>(&fB[(size_t)c * ld + r]);
      else tb[i] = *reinterpret_cast<const d2*>(&pB[(size_t)c * ld + r]);
      tp[i] = *reinterpret_cast<const d2*>(&pP[(size_t)c * ld + r]);
    }
    if constexpr (FX == 3) {  // the own column block is folded here: everything it has received lies in the fixed-point arrays
#pragma unroll
      for (int i = 0; i < NI; ++i) td[i] = d2v(fix2d(fd[i].x), fix2d(fd[i].y));
#pragma unroll
      for (int i = 0; i < NI / 2; ++i) tb[i] = d2v(fix2d(fb[i].x), fix2d(fb[i].y));
    }
#if defined(PPO_BA_TEST_HOOKS) && defined(PPO_CHOL_EXPERIMENT_PAIR_BOUND) && PPO_CHOL_EXPERIMENT_PAIR_BOUND == 3
    d2 tx[2 * NI];  // (timing experiment: what a real pair launch must also pull through this CU -- block row k+1's tile and its previous-panel rows, 64 KB)
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int p = tid + NT * i, c = p >> 5, r = 2 * (p & 31);
      tx[i] = *reinterpret_cast<const d2*>(&pB[(size_t)c * ld + r]);
      tx[NI + i] = *reinterpret_cast<const d2*>(&pP[(size_t)c * ld + r]);
    }
#endif
    rold = rsrc[own ? row0 + 32 * half + ((tid & 255) >> 3) : k0];  // (for the very end: requested here, not there; unconditional like the others)
    if constexpr (FX != 0) {
      // the right-hand-side rows of block k + b follow their column block: open -> the update is added to bfix as an integer; folded by this launch ->
      // they start from conv(bfix); else they live in `rhs`.  (The bfix request is unconditional like every other of this prologue: a load under an `if`
      // makes the compiler drain the requests in front of it first.)
      const int rblk = row0 / NB;
      const double rf = fix2d(fxb[own ? row0 + 32 * half + ((tid & 255) >> 3) : k0]);
      rhs_open = own && rblk >= fx.fold_hi;
      if (own && !rhs_open && rblk >= fx.fold_lo) rold = rf;
    }
    if (STEP_DEAD()) return;  // (everything above is a request; nothing has been written)
    STAMP(6);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int p = tid + NT * i, c = p >> 5, r = 2 * (p & 31);
      sL[r][c] = (r >= c) ? td[i].x : 0.0;
      sL[r + 1][c] = (r + 1 >= c) ? td[i].y : 0.0;
      *reinterpret_cast<d2*>(&sQ[c][r]) = tq[i];
    }
    if (tid < NB) sB[NB][tid] = rv;
    if (tid == 0) sBad = 0;
    for (int i = tid; i < (SB + 1) * SB; i += NT) sId[i / SB][i % SB] = (i / SB == i % SB) ? 1.0 : 0.0;
    __syncthreads();
    STAMP(5);
    if (updc && tid < 256) tile_update1(sL, wave, 0, wave == 0, sQ, sQ);  // block column 0 of the diagonal block: what potf2(0) needs
#pragma unroll
    for (int i = 0; i < NI / 2; ++i) {
      const int p = tid + NT * i, c = p >> 4, r = 32 * half + 2 * (p & 15);
      sB[r][c] = tb[i].x;
      sB[r + 1][c] = tb[i].y;
      *reinterpret_cast<d2*>(&sP[c][r]) = tp[i];
    }
#if defined(PPO_BA_TEST_HOOKS) && defined(PPO_CHOL_EXPERIMENT_PAIR_BOUND) && PPO_CHOL_EXPERIMENT_PAIR_BOUND == 3
#pragma unroll
    for (int i = 0; i < NI; ++i) {  // (they have to land somewhere: over the own half, wrong results)
      const int p = tid + NT * i, c = p >> 5, r = 2 * (p & 31);
      *reinterpret_cast<d2*>(&sP[c][r]) = tx[i] + tx[NI + i];
    }
#endif
  }
  __syncthreads();
  if (tid >= 256) return;  // loader waves
  STAMP(1);
#if defined(PPO_BA_TEST_HOOKS) && defined(PPO_CHOL_EXPERIMENT_PAIR_BOUND)
  // (timing experiment of tools/run_chol_experiments.sh, WRONG results: two panels per launch with the second panel's inputs "faked local" -- the
  //  workgroup walks the whole chain of a panel a second time on what lies in LDS, nothing more is loaded or stored: the UPPER BOUND of what a
  //  pair launch can save, i.e. one launch boundary + prologue + epilogue per pair.  PPO_CHOL_EXPERIMENT_PAIR_BOUND=2 also puts the redundant
  //  work of a real pair design into the shadow of the first chain: block row k+1's update and substitution, here as the own-half jobs run
  //  three times over.)
#pragma unroll 1
  for (int pair_rep = 0; pair_rep < 2; ++pair_rep) {
  if (pair_rep == 1) {
    if (updc) tile_update1(sL, wave, 0, wave == 0, sQ, sQ);
    __syncthreads();
  }
#define PPO_PAIR_SIDE_LOOP for (int side_rep = 0; side_rep < ((PPO_CHOL_EXPERIMENT_PAIR_BOUND >= 2 && pair_rep == 0) ? 3 : 1); ++side_rep)
#define PPO_PAIR_D (PPO_CHOL_EXPERIMENT_PAIR_BOUND != 3 || side_rep == 0)  // (=3: only the own-row jobs are redundant, the diagonal block's are not)
#else
#define PPO_PAIR_SIDE_LOOP
#define PPO_PAIR_D true
#endif
#pragma unroll
  for (int kk = 0; kk < NB / SB; ++kk) {
    const int o = kk * SB;
    STAMP1(24 + kk);
    if (wave == 0) {  // diagonal 16x16 block + the sub-diagonal tiles of this block column
      const int bad = potf2_16(sL, sT[kk], sId, o, true);
      if (lane == 0 && bad && sBad == 0) sBad = o + bad;
#if defined(PPO_BA_TEST_HOOKS) && defined(PPO_CHOL_EXPERIMENT_NO_SIDE)  // (timing experiment: wrong results, the register factorisations without the MFMA jobs in their shadow)
    } else if (true) {
#endif
    } else PPO_PAIR_SIDE_LOOP if (kk == 0) {
      if (updc && PPO_PAIR_D) tile_update1(sL, wave, 1, wave == 1, sQ, sQ);  // D(1,1) D(2,1) D(3,1): what the next factorisation needs
      if (uo) tile_update1(sB, r0t, wave - 1, false, sP, sQ);
    } else if (kk == 1) {
      // (first: the three tiles of the diagonal block that block column 0 still owes an update and that the factorisation running now does
      //  not read -- (2,2) (3,2) (3,3); as part of the update round behind potf2(0) they made it two rounds of matrix instructions, 0.64 us
      //  of the chain instead of 0.28)
      if (wave == 1) {
        if (PPO_PAIR_D) mfma_sub_16(sL, 3 * SB, 3 * SB, sL, 3 * SB, sL, 3 * SB, 0);
        if (uo) tile_update1(sB, r0t, 3, false, sP, sQ);
        if (own) solve_rows<0>(sB, r0t, sL, sT);
      } else if (wave == 2) {
        if (PPO_PAIR_D) mfma_sub_16(sL, 2 * SB, 2 * SB, sL, 2 * SB, sL, 2 * SB, 0);
        if (updc && PPO_PAIR_D) tile_update1(sL, 2, 2, true, sQ, sQ);
        if (uo) tile_update1(sB, r1t, 0, false, sP, sQ);
      } else {
        if (PPO_PAIR_D) mfma_sub_16(sL, 3 * SB, 2 * SB, sL, 3 * SB, sL, 2 * SB, 0);
        if (updc && PPO_PAIR_D) tile_update1(sL, 3, 2, false, sQ, sQ);
        if (uo) tile_update1(sB, r1t, 1, false, sP, sQ);
      }
    } else if (kk == 2) {
      if (wave == 1) {
        if (uo) tile_update1(sB, r1t, 2, false, sP, sQ);
        if (own) solve_rows<1>(sB, r0t, sL, sT);
      } else if (wave == 2) {
        if (updc && PPO_PAIR_D) tile_update1(sL, 3, 3, true, sQ, sQ);
        if (uo) tile_update1(sB, r1t, 3, false, sP, sQ);
      } else {
        if (PPO_PAIR_D) { solve_rhs_row(sB, 0, sL, sT); solve_rhs_row(sB, 1, sL, sT); }
      }
    } else {
      if (wave == 1) {
        if (own) { solve_rows<0>(sB, r1t, sL, sT); solve_rows<1>(sB, r1t, sL, sT); solve_rows<2>(sB, r1t, sL, sT); }
      } else if (wave == 2) {
        if (own) solve_rows<2>(sB, r0t, sL, sT);
      } else {
        if (PPO_PAIR_D) solve_rhs_row(sB, 2, sL, sT);
      }
    }
    STAMP1(28 + kk);
    __syncthreads();
    STAMP(8 + 2 * kk);
    if (kk == NB / SB - 1) break;
    // trailing 16x16 tiles of the diagonal block (bi >= bj > kk): A[bi][bj] -= X_bi X_bj^T, dealt round-robin to the 4 waves
    if (kk == 0) {  // behind potf2(0) only block column 1 -- what potf2(1) reads; the other three tiles ride in the shadow of potf2(1)
      if (wave < 3) mfma_sub_16(sL, (wave + 1) * SB, SB, sL, (wave + 1) * SB, sL, SB, o);
    } else {
      int t = 0;
#pragma unroll
      for (int bi = kk + 1; bi < NB / SB; ++bi)
#pragma unroll
        for (int bj = kk + 1; bj <= bi; ++bj, ++t)
          if ((t & 3) == wave) mfma_sub_16(sL, bi * SB, bj * SB, sL, bi * SB, sL, bj * SB, o);
    }
    __syncthreads();
    STAMP(9 + 2 * kk);
  }
#if defined(PPO_BA_TEST_HOOKS) && defined(PPO_CHOL_EXPERIMENT_PAIR_BOUND)
  if (pair_rep == 0) {  // the first panel's last substitution steps, before the second chain starts
    if (own && wave < 2) solve_rows<3>(sB, wave == 0 ? r0t : r1t, sL, sT);
    if (wave == 3) solve_rhs_row(sB, NB / SB - 1, sL, sT);
    __syncthreads();
  }
  }
#endif
  const int bad = sBad;
  STAMP(2);
  if (own && wave < 2) solve_rows<3>(sB, wave == 0 ? r0t : r1t, sL, sT);
  if (wave == 3) solve_rhs_row(sB, NB / SB - 1, sL, sT);
  __syncthreads();
  STAMP(3);
  if (b == 0) {
    if (k0 == 0)  // arm the hand-off granules of the backward substitution (chol_back_all)
      for (int i = tid; i < ld; i += 256) xout[i] = __longlong_as_double((long long)kSentinel);
    if (tid == 0) {
      if (k0 == 0) *info = bad;  // also resets the flag of the previous solve
      else if (bad) atomicCAS(info, 0, k0 + bad);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int p = tid + 256 * i, c = p >> 5, r = 2 * (p & 31);
      if (r + 1 >= c) *reinterpret_cast<d2*>(&A[(size_t)(k0 + c) * ld + k0 + r]) = d2v(sL[r][c], sL[r + 1][c]);
    }
    if (tid < NB) rhs[k0 + tid] = sB[NB][tid];  // y_k
    return;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int p = tid + 256 * i, c = p >> 4, r = 32 * half + 2 * (p & 15);
    *reinterpret_cast<d2*>(&A[(size_t)(k0 + c) * ld + row0 + r]) = d2v(sB[r][c], sB[r + 1][c]);
  }
  {  // r_b -= L_bk y_k: eight threads per row (as 32 threads x 64 dependent multiply-adds this was 0.5 us at the very end of the launch)
    const int myrow = 32 * half + (tid >> 3), m0 = 8 * (tid & 7);
    double sacc = 0.0;
#pragma unroll
    for (int mm = 0; mm < 8; ++mm) sacc = __builtin_fma(sB[myrow][m0 + mm], sB[NB][m0 + mm], sacc);
    sacc += __shfl_xor(sacc, 1, 64);
    sacc += __shfl_xor(sacc, 2, 64);
    sacc += __shfl_xor(sacc, 4, 64);
    if ((tid & 7) == 0) {
      if constexpr (FX != 0) {
        if (rhs_open) {
          if (!(fabs(sacc) < 4.0e11)) *fx.detbad = 1.0;
          else atomicAdd(reinterpret_cast<unsigned long long*>(const_cast<long long*>(fxb)) + row0 + myrow, (unsigned long long)__double2ll_rn(-sacc * kStepFixScale));
        } else {
          rhs[row0 + myrow] = rold - sacc;
        }
      } else if (inplace_rhs) atomicAdd(&rhs[row0 + myrow], -sacc);  // (overlapped reduction: the pair workgroups may be adding to this row too)
      else rhs[row0 + myrow] = rold - sacc;
    }
  }
  STAMP(4);
}
#undef STEP_DEAD

[[maybe_unused]] constexpr auto chol_step = chol_step_t<false>;  // (tools/chol_probe.hip launches single steps)

#ifdef PPO_BA_TEST_HOOKS
#include "ppo_chol_fwd.inc"  // chol_fwd_all: the one-launch forward factorisation (measured, not adopted)
#endif

// ------------------------------------------------------------------------------------------------------------------
// chol_back_all: the whole backward substitution L^T x = y in ONE launch, one workgroup per 64-block, pipelined through
// memory.  Workgroup j starts from y_j, subtracts L_ij^T x_i for i = nb-1 .. j+1 as the x_i appear, then solves its own
// diagonal block and publishes x_j.  Hand-off is by data-tagged granules (MI355X_MICROARCH.md, handoff-1to1, ~1 us per
// hop and no fences): x was pre-filled with a sentinel bit pattern by chol_step(0); the producer writes each value
// with ONE 8-byte write-through (sc1) store, consumers poll the values they need with sc1 loads until none of them is the
// sentinel.  Progress: block nb-1 waits for nobody and is dispatched first, every other block waits only for blocks dispatched before it; every spin is
// bounded (a timeout marks the solve as failed instead of hanging the device).
// The per-panel launches this replaces cost ~7.5 us each (19 at n = 1216), this pipeline ~2 us per block.
// ------------------------------------------------------------------------------------------------------------------

__device__ __forceinline__ double load_sc1(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void store_sc1(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

constexpr int kBackThreads = 320;  // four compute waves + the polling wave
constexpr int kRing = 32;          // blocks of x the polling wave may run ahead of the slowest compute wave
__global__ __launch_bounds__(kBackThreads) void chol_back_all(const int* gate_word, int gate_expect, const double* __restrict__ A, int ld, const double* __restrict__ y, double* x,
                                                              int* __restrict__ info, int* flags, int nflags, const double* __restrict__ inv) {
  __shared__ double sL[NB][LP];
  __shared__ double sV[NB];
  if (gate_word != nullptr && *gate_word != gate_expect) return;
  __shared__ double sXr[kRing][NB];  // x_i as the polling wave saw them, slot i % kRing
  __shared__ int s_seen;             // smallest i with x_i in the ring
  __shared__ int s_prog[4];          // smallest i compute wave w has finished reading
  __shared__ int s_red, s_dead;
  // every eighth workgroup works: workgroups are dealt to the 8 XCDs round-robin, so the whole pipeline sits behind ONE L2
  if (blockIdx.x & 7) {
    // the others have nothing to do in the product build (nflags = 0); in the test-hooks build they re-arm the hand-off flags of
    // the one-launch experiment for the next solve
    const int w = (blockIdx.x >> 3) * 7 + (blockIdx.x & 7) - 1, nw = (gridDim.x >> 3) * 7;
    for (int i = w * kBackThreads + threadIdx.x; i < nflags; i += nw * kBackThreads) flags[i] = 0;
    return;
  }
  // The head of the chain (block nb - 1) is the FIRST working workgroup of the grid and every workgroup waits only for workgroups dispatched
  // before it: the chain makes progress however few of them are resident at once.  (All of them land on one XCD -- 32 CUs, one workgroup of
  // five waves per CU: numbered the other way round the 33rd block could never start and the 32 in front of it waited for it for ever.)
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, nb = ld / NB, j = nb - 1 - (int)(blockIdx.x >> 3), k0 = j * NB;
  const int c = (tid & 255) >> 2, part = tid & 3;  // compute waves: 4 threads per column of block j, 16 rows of block i each
  // With T_j = L_jj^-1 and M_j = L_{j+1,j} T_j from the forward phase (inv_block_wg) the link of the chain is
  //   x_j = T_j^T (y_j - sum_{i > j+1} L_ij^T x_i) - M_j^T x_{j+1};
  // the head of the chain (block nb - 1, nothing to wait for) and a solve without the inverses keep the triangular sweep.
  const bool use_inv = inv != nullptr && j < nb - 1;
  const int i_last = use_inv ? j + 2 : j + 1;  // blocks i >= i_last are subtracted as L_ij^T x_i
  volatile int* const v_seen = &s_seen;
  volatile int* const v_prog = s_prog;
  volatile int* const v_red = &s_red;
  if (tid == 0) { s_seen = nb; s_red = 0; s_dead = 0; }
  if (tid < 4) s_prog[tid] = nb;
  if (wave == 4) {
    // ---- the polling wave.  Hand-off is by data-tagged granules; a poll is ONE request of 512 bytes by the whole wave (the first lane-0-then-
    // reload scheme paid two memory round trips, 0.4 us each, per block); it backs off with its distance from the pipeline front (25
    // workgroups polling flat out made a hop 5 us) and runs ahead of the compute waves through a ring in LDS, so that what a compute wave
    // does with x_i never delays the moment x_{i-1} is seen.
    __syncthreads();
    bool dead = false;
#pragma unroll 1
    for (int i = nb - 1; i > j; --i) {
      if (i + kRing <= nb - 1) {  // the slot still holds block i + kRing: wait until every compute wave is through with it
        while (max(max(v_prog[0], v_prog[1]), max(v_prog[2], v_prog[3])) > i + kRing) __builtin_amdgcn_s_sleep(2);
      }
      const double* xi = &x[(size_t)i * NB];
      const int nap = i - j <= 2 ? 0 : min(i - j - 1, 24);  // (the two workgroups next in line poll flat out)
      double v = load_sc1(&xi[lane]);
      int spins = 0;
      while (!dead && __any((unsigned long long)__double_as_longlong(v) == kSentinel)) {
        if (++spins > kSpinCap) { dead = true; break; }
        for (int q = 0; q < nap; ++q) __builtin_amdgcn_s_sleep(4);
        v = load_sc1(&xi[lane]);
      }
      sXr[i & (kRing - 1)][lane] = v;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) {
        if (dead) s_dead = 1;
        *v_seen = i;
      }
    }
    return;
  }
  // ---- everything that does not depend on other workgroups is fetched first
  d2 td[8];
  double trow[16];  // use_inv: T_j[16 wave + q][lane] -- this wave's share of c_j = T_j^T v (its 16 columns of block j are the 16 rows of T_j)
  if (use_inv) {
#pragma unroll
    for (int q = 0; q < 16; ++q) trow[q] = inv[(size_t)j * kInvDoubles + (size_t)(16 * wave + q) * NB + lane];
#pragma unroll
    for (int q = 0; q < 16; ++q) asm volatile("" : "+v"(trow[q]));  // (requested HERE: the compiler had moved these loads behind the loop, 1 - 3 us onto the chain)
  } else {
    load_tile(&A[(size_t)k0 * ld + k0], ld, td);
  }
  double acc = (part == 0) ? y[k0 + c] : 0.0;
  d2 lnext[8];
  if (nb - 1 >= i_last) {
    const d2* col = reinterpret_cast<const d2*>(&A[(size_t)(k0 + c) * ld + (size_t)(nb - 1) * NB + part * 16]);
#pragma unroll
    for (int r = 0; r < 8; ++r) lnext[r] = col[r];
  }
  double mrow[16];  // use_inv: M_j[16 wave + q][lane] -- this wave's share of M_j^T x_{j+1}
  if (use_inv) {
    const double* Mj = &inv[(size_t)j * kInvDoubles + (size_t)NB * NB];
#pragma unroll
    for (int q = 0; q < 16; ++q) mrow[q] = Mj[(size_t)(16 * wave + q) * NB + lane];
#pragma unroll
    for (int q = 0; q < 16; ++q) asm volatile("" : "+v"(mrow[q]));  // (the same)
  }
  if (!use_inv) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int p = tid + 256 * i, cc = p >> 5, r = 2 * (p & 31);
      sL[r][cc] = (r >= cc) ? td[i].x : 0.0;
      sL[r + 1][cc] = (r + 1 >= cc) ? td[i].y : 0.0;
    }
  }
  __syncthreads();  // (the only workgroup barrier: the polling wave joins it before its first poll)
#pragma unroll 1
  for (int i = nb - 1; i >= i_last; --i) {
    d2 lcur[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) lcur[r] = lnext[r];
    if (i - 1 >= i_last) {
      const d2* col = reinterpret_cast<const d2*>(&A[(size_t)(k0 + c) * ld + (size_t)(i - 1) * NB + part * 16]);
#pragma unroll
      for (int r = 0; r < 8; ++r) lnext[r] = col[r];
    }
    while (*v_seen > i) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (i < 10) STAMPB(20 + i);
    const double* xs = &sXr[i & (kRing - 1)][part * 16];
    double xv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) xv[r] = xs[r];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      acc = __builtin_fma(-lcur[r].x, xv[2 * r], acc);
      acc = __builtin_fma(-lcur[r].y, xv[2 * r + 1], acc);
    }
    if (lane == 0) v_prog[wave] = i;  // (LDS operations of a wave execute in order: the reads above are done)
  }
  STAMPB(35);
  acc += __shfl_xor(acc, 1, 64);
  acc += __shfl_xor(acc, 2, 64);
  double xj;
  if (use_inv) {
    // Every wave forms its 16 terms of c_j = T_j^T v (column 16 wave + q of block j sits in lane 4 q) for all 64 rows -- ready before
    // x_{j+1} shows up -- and then its 16 terms of M_j^T x_{j+1}; wave 0 adds the eight partial vectors.  (One wave holding a whole
    // column of M_j in registers spilled: 256 registers per wave with five waves in the workgroup.)
    double (*const sC)[NB] = reinterpret_cast<double (*)[NB]>(&sL[0][0]);  // [0..3] partial c_j, [4..7] partial M_j^T x_{j+1} (sL is free in this mode)
    // (the 16 column sums of this wave go through its own corner of LDS: read back as one batch of broadcast reads -- as 16 readlane
    //  pairs feeding the multiply-adds this was 0.36 us on the chain)
    if (part == 0) sV[c] = acc;
    wave_lds_fence();
    double av[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) av[q] = sV[16 * wave + q];
    __builtin_amdgcn_sched_barrier(0);
    double p0 = 0.0, p1 = 0.0;
#pragma unroll
    for (int q = 0; q < 16; q += 2) {
      p0 = __builtin_fma(trow[q], av[q], p0);
      p1 = __builtin_fma(trow[q + 1], av[q + 1], p1);
    }
    sC[wave][lane] = p0 + p1;
    STAMPB(36);
    STAMPB(32);
    while (*v_seen > j + 1) __builtin_amdgcn_s_sleep(0);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    STAMPB(33);
    const double* xn = &sXr[(j + 1) & (kRing - 1)][16 * wave];
    double xv[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) xv[q] = xn[q];
    double m0 = 0.0, m1 = 0.0;
#pragma unroll
    for (int q = 0; q < 16; q += 2) {
      m0 = __builtin_fma(mrow[q], xv[q], m0);
      m1 = __builtin_fma(mrow[q + 1], xv[q + 1], m1);
    }
    sC[4 + wave][lane] = m0 + m1;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) atomicAdd(&s_red, 1);
    if (wave != 0) return;
    while (*v_red < 4) __builtin_amdgcn_s_sleep(0);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    STAMPB(30);
    xj = ((sC[0][tid] + sC[1][tid]) + (sC[2][tid] + sC[3][tid])) - ((sC[4][tid] + sC[5][tid]) + (sC[6][tid] + sC[7][tid]));
    STAMPB(34);
  } else {
    if (part == 0) sV[c] = acc;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) atomicAdd(&s_red, 1);
    if (wave != 0) return;
    while (*v_red < 4) __builtin_amdgcn_s_sleep(0);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    double lcol[NB], inv_d = 1.0;  // lane r: the pre-scaled column r of L_jj (fetched here, not ahead of the loop: 128 registers the loop cannot spare)
    bwd_load_64(sL, lcol, inv_d);
    xj = bwd_vec_64(sV[tid], lcol, inv_d);
  }
  if (*(volatile int*)&s_dead) atomicMax(info, 1 << 30);
  store_sc1(&x[k0 + tid], xj);
  STAMPB(31);
}

#include "ppo_chol_small.inc"  // chol_small: n <= 192 in one launch of one workgroup, the matrix in LDS

struct GraphKey {
  double* A;
  const double* b;
  double* x;
  double* ws;
  int* info;
  int n;
  bool operator<(const GraphKey& o) const {
    if (A != o.A) return A < o.A;
    if (b != o.b) return b < o.b;
    if (x != o.x) return x < o.x;
    if (ws != o.ws) return ws < o.ws;
    if (info != o.info) return info < o.info;
    return n < o.n;
  }
};
std::mutex g_mu;
std::map<GraphKey, hipGraphExec_t> g_graphs;

#ifdef PPO_BA_TEST_HOOKS
// workspace: y [ld] | pieces [nb][4][kPiece] | P [nb][nb][NB] | flags (ints): tile [nb][nb], piece [nb][4], yflag [nb], pcnt [nb]
size_t ws_flag_ints(int nb) { return (size_t)nb * nb + 6 * (size_t)nb; }
FwdWs carve(double* ws, int ld) {
  const int nb = ld / NB;
  FwdWs w;
  w.y = ws;
  w.pieces = ws + ld;
  w.P = w.pieces + (size_t)nb * 4 * kPiece;
  w.tile = reinterpret_cast<int*>(w.P + (size_t)nb * nb * NB);
  w.piece = w.tile + (size_t)nb * nb;
  w.yflag = w.piece + 4 * nb;
  w.pcnt = w.yflag + nb;
  return w;
}
#endif

// doubles in front of the inverse blocks: y [ld] (+ the one-launch experiment's arrays in the test-hooks build)
size_t inv_offset(int ld) {
  const int nb = ld / NB;
#ifdef PPO_BA_TEST_HOOKS
  return ((size_t)ld + (size_t)nb * 4 * kPiece + (size_t)nb * nb * NB + (ws_flag_ints(nb) + 1) / 2 + 8 + 1) / 2 * 2;
#else
  (void)nb;
  return (size_t)ld + 8;
#endif
}

// One launch per panel + one for the backward substitution.  (The one-launch factorisation, ppo_chol_fwd.inc, exists only in the
// test-hooks build: PPO_BA_CHOL_ONELAUNCH=1 selects it there; DESIGN.md 4.1 has the measurement.)
int enqueue_all(double* A, int ld, const double* b, double* x, int* info, double* ws, hipStream_t s, const CholOverlap* ov = nullptr, LmGate gate = LmGate{nullptr, 0}, int n_active = -1,
                const CholFixed* fixed = nullptr) {
  const int nb = ld / NB;
  if (chol_small_applies(ld, n_active) && (!ov || ov->fin[0] >= nb)) {  // (an overlapped reduction must have left nothing for the panel launches to carry)
    const int n = (n_active > 0 && n_active <= ld) ? n_active : ld;
    if (fixed) hipLaunchKernelGGL(chol_small<true>, dim3(1), dim3(kSmallThreads), 0, s, gate.epoch_word, gate.expect, reinterpret_cast<const double*>(fixed->Afix), ld, n, reinterpret_cast<const double*>(fixed->bfix), x, info);
    else hipLaunchKernelGGL(chol_small<false>, dim3(1), dim3(kSmallThreads), 0, s, gate.epoch_word, gate.expect, (const double*)A, ld, n, ov ? (const double*)ov->ctx.bs : b, x, info);
    return hipGetLastError() == hipSuccess ? PPO_BA_OK : PPO_BA_ERR_HIP;
  }
  // [ld] forward-substituted right-hand side (overlapped reduction: in place, the pair workgroups add to it; deterministic mode: the pair workgroups add to
  // bfix, y is the factorisation's own -- the workspace)
  double* y = (ov && !fixed) ? ov->ctx.bs : ws;
  int* flags = nullptr;
  int nflags = 0;
  // T_k and M_k of every block but the last (inv_block_wg -> chol_back_all), at the end of the workspace
  double* inv = nb >= 2 ? ws + inv_offset(ld) : nullptr;
#ifdef PPO_BA_TEST_HOOKS
  static const bool no_inv = []() { const char* e = PPO_DEV_ENV("PPO_BA_CHOL_NO_INV"); return e && e[0] == '1'; }();  // A/B: the backward substitution by triangular sweeps
  if (no_inv) inv = nullptr;
  const FwdWs w = carve(ws, ld);
  flags = w.tile;
  nflags = (int)ws_flag_ints(nb);
  static const bool one_launch = []() { const char* e = PPO_DEV_ENV("PPO_BA_CHOL_ONELAUNCH"); return e && e[0] == '1'; }();
  if (one_launch) inv = nullptr;  // (no chol_step launches: nobody forms T and M)
  if (one_launch) {
#ifdef PPO_CHOL_FWD_ONE_XCD
    hipLaunchKernelGGL(chol_fwd_all, dim3(8 * (nb < 2 ? 1 : 1 + nb * (nb - 1) / 2)), dim3(256), 0, s, A, ld, b, x, info, w);
#else
    hipLaunchKernelGGL(chol_fwd_all, dim3(nb < 2 ? 1 : 1 + nb * (nb - 1) / 2), dim3(256), 0, s, A, ld, b, x, info, w);
#endif
  } else
#endif
  {
#if defined(PPO_BA_TEST_HOOKS) && defined(PPO_CHOL_EXPERIMENT_PAIR_BOUND)
    for (int k = 0; k < nb; k += 2) {  // (timing experiment: one launch per PAIR of panels, each walking two chains)
#else
    for (int k = 0; k < nb; ++k) {
#endif
      const int k0 = k * NB;
      const int m = nb - k;  // column workgroups: 1 (diagonal) + 2 per block row below (row halves); then the trailing tiles
      StepOverlap o = step_no_overlap(nb);
      if (ov) {  // ... then the pair workgroups of the column blocks [fin[k], fin[k + 1])
        o.ctx = ov->ctx;
        o.pc = ov->pc;
        o.atomic_lo = ov->fin[k];
        o.atomic_hi = ov->fin[k + 1] > ov->fin[k] ? ov->fin[k + 1] : ov->fin[k] - 1;  // (no pair workgroups in this launch: no atomic tiles)
        o.chunk_lo = ov->cb_start[ov->fin[k]];
        o.chunk_hi = ov->cb_start[ov->fin[k + 1]];
        const int per_wg = kStepThreads / 64;
        o.npair_wg = o.chunk_hi > o.chunk_lo ? ((o.chunk_hi - o.chunk_lo + per_wg - 1) / per_wg + 7) / 8 * 8 : 0;
      }
#ifdef PPO_BA_TEST_HOOKS
      if (PPO_DEV_PROBE("PPO_BA_OVERLAP_EMPTY")) o.chunk_hi = o.chunk_lo;  // timing experiment (WRONG results): the pair workgroups find nothing to do
#endif
      const dim3 grid(1 + 2 * (m - 1) + (k > 0 ? m * (m - 1) / 2 : 0) + o.npair_wg + (k > 0 && inv ? 1 : 0));
      if (fixed) {
        // deterministic mode: launch k folds the column blocks that are final by now and have not been folded yet -- [done, fin[k]) with
        // done = 0, 1, fin[k - 1] for k = 0, 1, >= 2 (launch 0 touches block column 0 only, every later launch every block >= k).  Nothing is
        // overlapped (ov == NULL): every block is final from the start.
        const int fin_k = ov ? ov->fin[k] : nb, fin_km1 = (ov && k > 0) ? ov->fin[k - 1] : nb;
        StepFix fx{fixed->Afix, fixed->bfix, fixed->detbad, k == 0 ? 0 : (k == 1 ? 1 : fin_km1), k == 0 ? 1 : fin_k};
        const int mode = (k >= fx.fold_lo && k < fx.fold_hi) ? 3 : 1;  // (launches 0 and 1 always fold their own block; later ones when fin[k - 1] == k)
#define PPO_STEP_FX(OV_, FX_) hipLaunchKernelGGL((chol_step_t<OV_, FX_>), grid, dim3(kStepThreads), 0, s, gate.epoch_word, gate.expect, A, ld, k0, reinterpret_cast<const double*>(fixed->Afix), y, x, info, 0, inv, o, fx)
        if (o.npair_wg > 0) {
          if (mode == 3) PPO_STEP_FX(true, 3);
          else PPO_STEP_FX(true, 1);
        } else {
          if (mode == 3) PPO_STEP_FX(false, 3);
          else PPO_STEP_FX(false, 1);
        }
#undef PPO_STEP_FX
        continue;
      }
      const StepFix nofx{nullptr, nullptr, nullptr, 0, 0};
      if (o.npair_wg > 0) hipLaunchKernelGGL(chol_step_t<true>, grid, dim3(kStepThreads), 0, s, gate.epoch_word, gate.expect, A, ld, k0, ov ? y : b, y, x, info, ov ? 1 : 0, inv, o, nofx);
      else hipLaunchKernelGGL(chol_step_t<false>, grid, dim3(kStepThreads), 0, s, gate.epoch_word, gate.expect, A, ld, k0, ov ? y : b, y, x, info, ov ? 1 : 0, inv, o, nofx);
    }
  }
  hipLaunchKernelGGL(chol_back_all, dim3(nb * 8), dim3(kBackThreads), 0, s, gate.epoch_word, gate.expect, (const double*)A, ld, (const double*)y, x, info, flags, nflags, (const double*)inv);
  return hipGetLastError() == hipSuccess ? PPO_BA_OK : PPO_BA_ERR_HIP;
}

}  // namespace

// A: ld x ld column-major with ld a multiple of 64 (chol_padded_ld), lower triangle valid, pad diagonal = 1.
int chol_factor_solve(double* A, int ld, const double* b, double* x, int* info, double* ws, hipStream_t stream, LmGate gate, int n_active) {
  if (ld % NB != 0) return PPO_BA_ERR_ARG;
  if (ld == 0) return PPO_BA_OK;
  if (gate.epoch_word || chol_small_applies(ld, n_active)) return enqueue_all(A, ld, b, x, info, ws, stream, nullptr, gate, n_active);
  // Plain launches by default: with ~16 us per panel kernel the host stays ahead of the device and a hipGraph replay only
  // adds its entry/exit boundaries (measured 1826 vs 1834 LM iterations/s); PPO_BA_CHOL_GRAPH=1 selects the captured graph
  // (it paid when the solve was 56 short launches).
  static const bool use_graph = []() { const char* e = PPO_DEV_ENV("PPO_BA_CHOL_GRAPH"); return e && e[0] == '1'; }();
  if (!use_graph) return enqueue_all(A, ld, b, x, info, ws, stream);
  hipGraphExec_t exec = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    const GraphKey key{A, b, x, ws, info, ld};
    auto it = g_graphs.find(key);
    if (it == g_graphs.end()) {
      hipGraph_t graph = nullptr;
      if (hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal) != hipSuccess) return PPO_BA_ERR_HIP;
      const int rc = enqueue_all(A, ld, b, x, info, ws, stream);
      if (hipStreamEndCapture(stream, &graph) != hipSuccess || rc != PPO_BA_OK) return PPO_BA_ERR_HIP;
      if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) return PPO_BA_ERR_HIP;
      (void)hipGraphDestroy(graph);
      g_graphs[key] = exec;
    } else {
      exec = it->second;
    }
  }
  return hipGraphLaunch(exec, stream) == hipSuccess ? PPO_BA_OK : PPO_BA_ERR_HIP;
}

// The overlapped variant: S's column blocks [fin[0], nb) are still being reduced by pair workgroups inside the first launches.
int chol_factor_solve_overlapped(double* A, int ld, double* x, int* info, double* ws, const CholOverlap& ov, hipStream_t stream, int n_active) {
  if (ld % NB != 0 || kStepThreads != 512 || !ov.fin || !ov.cb_start) return PPO_BA_ERR_ARG;
  if (ld == 0) return PPO_BA_OK;
  return enqueue_all(A, ld, ov.ctx.bs, x, info, ws, stream, &ov, ov.gate, n_active);
}

// Deterministic mode (StepFix): the system was summed as fixed-point integers into fixed.Afix / fixed.bfix; A is the factorisation's own array and
// needs no initialisation.  ov (may be NULL): an overlapped reduction whose pair workgroups add to the fixed-point arrays (ov->ctx.S == Afix).
int chol_factor_solve_fixed(double* A, int ld, double* x, int* info, double* ws, const CholFixed& fixed, const CholOverlap* ov, LmGate gate, hipStream_t stream, int n_active) {
  if (ld % NB != 0 || kStepThreads != 512 || !fixed.Afix || fixed.bfix != fixed.Afix + (size_t)ld * ld || (ov && (!ov->fin || !ov->cb_start))) return PPO_BA_ERR_ARG;
  if (ld == 0) return PPO_BA_OK;
  return enqueue_all(A, ld, nullptr, x, info, ws, stream, ov, gate, n_active, &fixed);
}

// The one-launch solve of small systems (ppo_chol_small.inc) handles up to 192 unknowns; it is USED where it is the faster of the two
// (tools/chol_small_check.hip on MI355X, us per solve, one launch | panel path): n = 16: 7 | 18.5, 48: 14 | 18.5, 64: 17 | 18.5, 84: 27 | 31.5,
// 96: 27 | 31.5, 128: 37 | 31.5, 174: 55 | 45, 192: 62 | 45 -- one compute unit's FP64 matrix rate (44 ns per v_mfma_f64_16x16x4 and SIMD,
// measured) bounds the trailing updates of the first tile columns once n > 100.  PPO_BA_CHOL_SMALL (test-hooks build): 0 keeps every window on
// the panel path, N > 0 moves the crossover to N unknowns (<= 192) -- so that both paths stay covered by the same tests.
bool chol_small_applies(int ld, int n_active) {
  const char* e = PPO_DEV_ENV("PPO_BA_CHOL_SMALL");  // (read on every call in the test-hooks build: a test switches it; the product build has no environment)
  const int limit = e ? std::max(0, std::min(atoi(e), 192)) : 96;
  const int n = (n_active > 0 && n_active <= ld) ? n_active : ld;
  return ld > 0 && ld <= 192 && n <= limit;
}

// fin[k], k = 0 .. nb: column blocks of S that are final when launch k starts (fin[nb] = nb).  The reduction ahead of the factorisation
// covers `first_blocks` (>= 1) column blocks; launch k then carries the next blocks as long as their chunks stay within a share of all
// chunks (share0 for launch 0, which has no trailing tiles of its own), and always enough of them to keep fin[k + 1] >= k + 2.
void chol_overlap_schedule(int nb, const int32_t* cb_start, int first_blocks, double share0, double share, int* fin) {
  const long total = cb_start[nb];
  fin[0] = std::min(nb, std::max(1, first_blocks));
  for (int k = 0; k < nb; ++k) {
    const int f = fin[k], need = std::min(nb, k + 2);
    const double sh = std::min(1.0, std::max(0.0, k == 0 ? share0 : share));  // (a share is a fraction of all chunks: NaN, inf and > 1 are clamped
    const long budget = (long)(sh * (double)total);                            //  before the conversion, which is undefined outside the range of long)
    int nf = f;
    while (nf < nb && (nf < need || (long)cb_start[nf + 1] - cb_start[f] <= budget)) ++nf;
    fin[k + 1] = nf;
  }
}

void chol_release(double* A) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (auto it = g_graphs.begin(); it != g_graphs.end();) {
    if (it->first.A == A) {
      (void)hipGraphExecDestroy(it->second);
      it = g_graphs.erase(it);
    } else {
      ++it;
    }
  }
}

size_t chol_workspace_doubles(int n) {  // must be ZERO when first used (the hand-off flags live in it)
  const int ld = chol_padded_ld(n), nb = ld / NB;
  return inv_offset(ld) + (size_t)nb * kInvDoubles;
}
int chol_padded_ld(int n) { return (n + NB - 1) / NB * NB; }
int chol_default_backend() { return PPO_SOLVER_NATIVE; }

}  // namespace ppo
