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
//   * chol_back(k): blocked backward substitution, the 64-vector held one entry per lane;
//   * the 2 x n/64 launches are captured once into a hipGraph and replayed per solve.
// FMAs are written explicitly (__builtin_fma): the library is built with -ffp-contract=off for the numeric Jacobians.
// info follows LAPACK potrf: 0 = positive definite, j > 0 = leading minor j is not (the LM loop treats that as
// "solver failed", levenberg.cpp:126-127).
#include "ppo_chol.hpp"

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

#ifndef STAMP
#define STAMP(i)  // tools/chol_probe.hip defines it to record wall-clock stamps
#endif

__device__ __forceinline__ double bcast(double v, int lane) {  // lane is wave-uniform (a constant after unrolling)
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, lane);
  hi = __builtin_amdgcn_readlane(hi, lane);
  return __hiloint2double(hi, lo);
}

// 64x64 tile at `base` (column-major, leading dimension ld) -> registers: thread t holds 8 double2, element pair
// p = t + 256 i covers column p >> 5, rows 2 (p & 31) and +1.  All 8 loads are issued before any use.
__device__ __forceinline__ void load_tile(const double* __restrict__ base, int ld, double2 (&t)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int p = threadIdx.x + 256 * i;
    t[i] = *reinterpret_cast<const double2*>(&base[(size_t)(p >> 5) * ld + 2 * (p & 31)]);
  }
}

// C[16x16] (rows r0.., cols c0.. of M) -= X[16x16] * Y[16x16]^T where X = rows xr.. / Y = rows yr.. of columns [o, o+16)
// in the padded LDS arrays MX / MY.  One wave, 4 MFMAs.  D layout of v_mfma_f64_16x16x4: col = lane & 15, row = (lane >> 4) + 4 reg.
__device__ __forceinline__ void mfma_sub_16(double (*M)[LP], int r0, int c0, const double (*MX)[LP], int xr, const double (*MY)[LP], int yr, int o) {
  const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
  v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int m0 = 0; m0 < SB; m0 += 4) {
    const double a = MX[xr + li][o + m0 + lk];  // A[i = li][k = lk]
    const double b = MY[yr + li][o + m0 + lk];  // B[k = lk][j = li]
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) M[r0 + lk + 4 * reg][c0 + li] -= acc[reg];
}

// C[16x16] = X[16x16] * T[16x16] written to rows r0.., cols c0.. of M, where X = rows xr.., columns [xo, xo+16) of MX and
// T is a 16x16 block stored [k][j] with row length 17.  In-place use (M == MX, same tile) is safe: the wave reads all of X
// before it writes.  With T = L^-T this is the triangular solve X L^T = B.
__device__ __forceinline__ void mfma_mul_16(double (*M)[LP], int r0, int c0, const double (*MX)[LP], int xr, int xo, const double (*T)[17]) {
  const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
  v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int m0 = 0; m0 < SB; m0 += 4) {
    const double a = MX[xr + li][xo + m0 + lk];  // A[i = li][k = lk]
    const double b = T[m0 + lk][li];             // B[k = lk][j = li]
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) M[r0 + lk + 4 * reg][c0 + li] = acc[reg];
}

// ---- f64 DPP: v_fmac_f64 / v_mov_b64 accept `row_newbcast:N` (source operand taken from lane N of the reader's own row of
// 16 lanes), which turns "broadcast + multiply-add" into ONE instruction.  The wave that factors a 16x16 block is issue-bound
// (tools/chol_probe.hip: ~4.5 cycles per instruction whatever the dependencies), so instruction count is what matters: the
// v_readlane formulation needed 2 broadcasts + 1 FMA per element update.  `s_nop 1` covers the VALU-write -> DPP-read hazard,
// which the compiler cannot see inside inline assembly.
template <int C, bool FRESH>
__device__ __forceinline__ void fmac_bcast(double& acc, double l_src, double mul) {  // acc -= l_src[lane C of my row] * mul
  if constexpr (FRESH)  // l_src was written by the preceding VALU instruction
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(l_src), "v"(mul), "n"(C));
  else
    asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(l_src), "v"(mul), "n"(C));
}
template <int C>
__device__ __forceinline__ double row_bcast(double v) {
  double d;
  asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(v), "n"(C));
  return d;
}

template <int J, int C>
struct ColUpdate {
  static __device__ __forceinline__ void run(double (&a)[SB], double (&e)[SB], double l, double le) {
    fmac_bcast<C, C == J + 1>(a[C], l, l);
    fmac_bcast<C, false>(e[C], l, le);
    if constexpr (C + 1 < SB) ColUpdate<J, C + 1>::run(a, e, l, le);
  }
};

template <int J>
struct ColSweep {
  static __device__ __forceinline__ void run(double (&a)[SB], double (&e)[SB], int& bad) {
    const double d = row_bcast<J>(a[J]);
    if (!(d > 0.0) && bad == 0) bad = J + 1;
    // 1/sqrt(d): v_rsq_f64 (measured 5e-8 relative) + two Newton steps y <- y (1.5 - (d/2) y^2): 3.8e-15, then 2.2e-16
    const double h = 0.5 * d;
    double y = __builtin_amdgcn_rsq(d);
    y = y * __builtin_fma(-h, y * y, 1.5);
    y = y * __builtin_fma(-h, y * y, 1.5);
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
__device__ __forceinline__ int potf2_16(double (*sL)[LP], double (*sT)[17], int o) {
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  const int xrow = o + SB * g + r;  // extra row of groups 1..3
  const bool has_x = g > 0 && xrow < NB;
  double a[SB], e[SB];
#pragma unroll
  for (int c = 0; c < SB; ++c) {
    a[c] = sL[o + r][o + c];
    e[c] = g == 0 ? (c == r ? 1.0 : 0.0) : (has_x ? sL[xrow][o + c] : 0.0);
  }
  int bad = 0;
  ColSweep<0>::run(a, e, bad);
  if (g == 0) {
#pragma unroll
    for (int c = 0; c < SB; ++c) {
      if (c <= r) sL[o + r][o + c] = a[c];
      sT[r][c] = e[c];
    }
  } else if (has_x) {
#pragma unroll
    for (int c = 0; c < SB; ++c) sL[xrow][o + c] = e[c];
  }
  return bad;
}

// Factor the 64x64 block held in sL (lower part valid) in place; sT[kk] receives L_kk^-T of the four 16x16 diagonal
// blocks.  All 256 threads participate.  Returns (in every thread) the 1-based index of the first bad pivot or 0.
__device__ int factor_64(double (*sL)[LP], double (*sT)[SB][17], int* sBad) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (tid == 0) *sBad = 0;
  __syncthreads();
#pragma unroll
  for (int kk = 0; kk < NB / SB; ++kk) {
    const int o = kk * SB;
    if (wave == 0) {  // diagonal 16x16 block + the sub-diagonal tiles of this block column
      const int bad = potf2_16(sL, sT[kk], o);
      if (lane == 0 && bad && *sBad == 0) *sBad = o + bad;
    }
    __syncthreads();
    STAMP(8 + 2 * kk);
    if (kk == NB / SB - 1) break;
    // trailing 16x16 tiles (bi >= bj > kk): A[bi][bj] -= X_bi X_bj^T, tiles dealt round-robin to the 4 waves
    int t = 0;
#pragma unroll
    for (int bi = kk + 1; bi < NB / SB; ++bi)
#pragma unroll
      for (int bj = kk + 1; bj <= bi; ++bj, ++t)
        if ((t & 3) == wave) mfma_sub_16(sL, bi * SB, bj * SB, sL, bi * SB, sL, bj * SB, o);
    __syncthreads();
    STAMP(9 + 2 * kk);
  }
  return *sBad;
}

// x <- L^-T v for the factored 64x64 block in sL, executed by one wave: lane i holds v_i in a register, the solution
// sweeps from the last unknown to the first, one broadcast + one FMA per step (the L reads do not depend on data).
__device__ __forceinline__ double bwd_vec_64(double v, const double (*sL)[LP], const double* sInv) {
  const int lane = threadIdx.x & 63;
  const double myinv = sInv[lane];
#pragma unroll
  for (int c = NB - 1; c >= 0; --c) {
    const double xc = bcast(v * myinv, c);
    if (lane == c) v = xc;
    else if (lane < c) v = __builtin_fma(-sL[c][lane], xc, v);
  }
  return v;
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

// ------------------------------------------------------------------------------------------------------------------
// chol_step(k): ONE launch per 64-wide panel.  With m = (ld - k0) / 64 block rows left, the grid is
//   * m "column" workgroups (blockIdx < m, block row b): apply the rank-64 update of the PREVIOUS panel (k0 - 64) to the
//     diagonal block (k,k) and to their own block (k+b, k), factor the diagonal block (every workgroup redundantly, in LDS),
//     solve their block against it, carry the right-hand side along as a 65th row;
//   * m (m - 1) / 2 "trailing" workgroups: apply the previous panel's update to one 64x64 tile (i >= j > k) each.
// Fusing the update of panel k-1 with the factorisation of panel k (look-ahead by one) halves the launches on the critical
// path: the column workgroups no longer wait for the whole trailing update to finish and be relaunched.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void chol_step(double* __restrict__ A, int ld, int k0, const double* rhs_in, double* rhs,
                                                 int* __restrict__ info) {
  __shared__ double sL[NB][LP];
  __shared__ double sB[NB + 1][LP];  // row NB carries r_k^T: the forward substitution is one more row of the block solve
  __shared__ double sT[NB / SB][SB][17];
  __shared__ __attribute__((aligned(16))) double sP[NB][UP];  // [m][row of block i] of the previous panel
  __shared__ __attribute__((aligned(16))) double sQ[NB][UP];  // [m][row of block j]
  __shared__ int sBad;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  const int wr = (wave >> 1) * 32, wc = (wave & 1) * 32;
  const int m = (ld - k0) / NB;
  const int kp = k0 - NB;  // previous panel's column
  STAMP(0);
  if ((int)blockIdx.x >= m) {
    // ---- trailing tile (bi >= bj >= 1 relative to block k) --------------------------------------------------------------
    int t = blockIdx.x - m, bi = 0;
    while ((bi + 1) * (bi + 2) / 2 <= t) ++bi;
    const int bj = t - bi * (bi + 1) / 2;
    const int r0 = k0 + NB + bi * NB, c0 = k0 + NB + bj * NB;
    double2 tp[8], tq[8];
    load_tile(&A[(size_t)kp * ld + r0], ld, tp);
    load_tile(&A[(size_t)kp * ld + c0], ld, tq);
    double cold[2][2][4];  // old values of the output quadrant, fetched while the panel tiles land
#pragma unroll
    for (int jc = 0; jc < 2; ++jc)
#pragma unroll
      for (int ir = 0; ir < 2; ++ir)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) cold[jc][ir][reg] = A[(size_t)(c0 + wc + 16 * jc + lk + 4 * reg) * ld + r0 + wr + 16 * ir + li];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int p = tid + 256 * i, mm = p >> 5, r = 2 * (p & 31);
      *reinterpret_cast<double2*>(&sP[mm][r]) = tp[i];
      *reinterpret_cast<double2*>(&sQ[mm][r]) = tq[i];
    }
    __syncthreads();
    v4d acc[2][2] = {};
    quadrant_product(sP, sQ, wr, wc, li, lk, acc);
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
  // ---- column workgroup ---------------------------------------------------------------------------------------------------
  const int b = blockIdx.x;
  const int row0 = k0 + b * NB;
  // step 0 takes the right-hand side from rhs_in (the caller's b stays intact), later steps continue in rhs
  const double* rsrc = (k0 == 0) ? rhs_in : rhs;
  const double rv = (tid < NB) ? rsrc[k0 + tid] : 0.0;
  if (k0 == 0) {
    double2 td[8], tb[8];
    load_tile(&A[(size_t)k0 * ld + k0], ld, td);
    if (b > 0) load_tile(&A[(size_t)k0 * ld + row0], ld, tb);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int p = tid + 256 * i, c = p >> 5, r = 2 * (p & 31);
      sL[r][c] = (r >= c) ? td[i].x : 0.0;
      sL[r + 1][c] = (r + 1 >= c) ? td[i].y : 0.0;
      if (b > 0) { sB[r][c] = tb[i].x; sB[r + 1][c] = tb[i].y; }
    }
  } else {
    double2 tp[8], tq[8];
    load_tile(&A[(size_t)kp * ld + k0], ld, tq);  // L[k][k-1]
    if (b > 0) load_tile(&A[(size_t)kp * ld + row0], ld, tp);  // L[k+b][k-1]
    double cd[2][2][4], co[2][2][4];
#pragma unroll
    for (int jc = 0; jc < 2; ++jc)
#pragma unroll
      for (int ir = 0; ir < 2; ++ir)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const size_t col = (size_t)(k0 + wc + 16 * jc + lk + 4 * reg) * ld;
          cd[jc][ir][reg] = (wr >= wc) ? A[col + k0 + wr + 16 * ir + li] : 0.0;
          co[jc][ir][reg] = (b > 0) ? A[col + row0 + wr + 16 * ir + li] : 0.0;
        }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int p = tid + 256 * i, mm = p >> 5, r = 2 * (p & 31);
      *reinterpret_cast<double2*>(&sQ[mm][r]) = tq[i];
      if (b > 0) *reinterpret_cast<double2*>(&sP[mm][r]) = tp[i];
    }
    __syncthreads();
    STAMP(5);
    v4d ad[2][2] = {}, ao[2][2] = {};
    if (wr >= wc) quadrant_product(sQ, sQ, wr, wc, li, lk, ad);  // the strictly upper quadrant of the diagonal block is not needed
    if (b > 0) quadrant_product(sP, sQ, wr, wc, li, lk, ao);
#pragma unroll
    for (int jc = 0; jc < 2; ++jc)
#pragma unroll
      for (int ir = 0; ir < 2; ++ir)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int c = wc + 16 * jc + lk + 4 * reg, r = wr + 16 * ir + li;
          sL[r][c] = (r >= c) ? cd[jc][ir][reg] - ad[jc][ir][reg] : 0.0;
          if (b > 0) sB[r][c] = co[jc][ir][reg] - ao[jc][ir][reg];
        }
  }
  if (tid < NB) sB[NB][tid] = rv;
  __syncthreads();
  STAMP(1);
  const int bad = factor_64(sL, sT, &sBad);
  STAMP(2);
  // X = B L_kk^-T by 16-wide block substitution.  Wave w owns rows 16w..16w+15 of the block for the whole sweep (its
  // tiles only read the shared L / L^-T), so the four waves run without any workgroup barrier; wave 3 also carries the
  // right-hand-side row.
#pragma unroll
  for (int kk = 0; kk < NB / SB; ++kk) {
    const int o = kk * SB;
    if (b > 0) {
      mfma_mul_16(sB, wave * SB, o, sB, wave * SB, o, sT[kk]);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int ct = kk + 1; ct < NB / SB; ++ct) mfma_sub_16(sB, wave * SB, ct * SB, sB, wave * SB, sL, ct * SB, o);
    }
    if (wave == 3) {  // rhs row: x = b T (16 lanes), then b[rest] -= x L[rest]^T (48 lanes at most)
      double xv = 0.0;
      if (lane < SB) {
#pragma unroll
        for (int mm = 0; mm < SB; ++mm) xv = __builtin_fma(sB[NB][o + mm], sT[kk][mm][lane], xv);
      }
      __builtin_amdgcn_wave_barrier();
      if (lane < SB) sB[NB][o + lane] = xv;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (lane < NB - o - SB) {
        const int c = o + SB + lane;
        double sacc = 0.0;
#pragma unroll
        for (int mm = 0; mm < SB; ++mm) sacc = __builtin_fma(sB[NB][o + mm], sL[c][o + mm], sacc);
        sB[NB][c] -= sacc;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
  __syncthreads();
  STAMP(3);
  if (b == 0) {
    if (tid == 0) {
      if (k0 == 0) *info = bad;  // also resets the flag of the previous solve
      else if (bad) atomicCAS(info, 0, k0 + bad);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int p = tid + 256 * i, c = p >> 5, r = 2 * (p & 31);
      if (r + 1 >= c) *reinterpret_cast<double2*>(&A[(size_t)(k0 + c) * ld + k0 + r]) = make_double2(sL[r][c], sL[r + 1][c]);
    }
    if (tid < NB) rhs[k0 + tid] = sB[NB][tid];  // y_k
    return;
  }
  const double rold = (tid < NB) ? rsrc[row0 + tid] : 0.0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int p = tid + 256 * i, c = p >> 5, r = 2 * (p & 31);
    *reinterpret_cast<double2*>(&A[(size_t)(k0 + c) * ld + row0 + r]) = make_double2(sB[r][c], sB[r + 1][c]);
  }
  if (tid < NB) {  // r_b -= L_bk y_k
    double sacc = 0.0;
#pragma unroll 16
    for (int mm = 0; mm < NB; ++mm) sacc = __builtin_fma(sB[tid][mm], sB[NB][mm], sacc);
    rhs[row0 + tid] = rold - sacc;
  }
  STAMP(4);
}

// ------------------------------------------------------------------------------------------------------------------
// chol_back: backward substitution step for block k (k0 = 64 k).  grid = max(k, 1): every workgroup recomputes
// x_k = L_kk^-T v_k; workgroup j < k applies v_j -= L_kj^T x_k; workgroup 0 also stores x_k.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void chol_back(const double* __restrict__ A, int ld, int k0, double* __restrict__ rhs) {
  __shared__ double sL[NB][LP];
  __shared__ double sInv[NB], sV[NB];
  const int tid = threadIdx.x, j = blockIdx.x;
  const int nblk_above = k0 / NB;
  // every global load is issued before the serial part
  const int c = tid >> 2, part = tid & 3;  // 4 threads per column, 16 rows each
  double lcol[16];
  double vold = 0.0, dia = 1.0, vk = 0.0;
  if (j < nblk_above) {
    const double* col = &A[(size_t)(j * NB + c) * ld + k0 + part * 16];
#pragma unroll
    for (int r = 0; r < 16; ++r) lcol[r] = col[r];
    if (part == 0) vold = rhs[j * NB + c];
  }
  if (tid < NB) { dia = A[(size_t)(k0 + tid) * ld + k0 + tid]; vk = rhs[k0 + tid]; }
  {
    double2 td[8];
    load_tile(&A[(size_t)k0 * ld + k0], ld, td);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int p = tid + 256 * i, cc = p >> 5, r = 2 * (p & 31);
      sL[r][cc] = (r >= cc) ? td[i].x : 0.0;
      sL[r + 1][cc] = (r + 1 >= cc) ? td[i].y : 0.0;
    }
  }
  if (tid < NB) sInv[tid] = 1.0 / dia;
  __syncthreads();
  if (tid < 64) {
    const double xk = bwd_vec_64(vk, sL, sInv);
    sV[tid] = xk;
    if (j == 0) rhs[k0 + tid] = xk;
  }
  __syncthreads();
  if (j < nblk_above) {
    double sacc = 0.0;
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc = __builtin_fma(lcol[r], sV[part * 16 + r], sacc);
    sacc += __shfl_xor(sacc, 1, 64);
    sacc += __shfl_xor(sacc, 2, 64);
    if (part == 0) rhs[j * NB + c] = vold - sacc;
  }
}

struct GraphKey {
  double* A;
  const double* b;
  double* x;
  int* info;
  int n;
  bool operator<(const GraphKey& o) const {
    if (A != o.A) return A < o.A;
    if (b != o.b) return b < o.b;
    if (x != o.x) return x < o.x;
    if (info != o.info) return info < o.info;
    return n < o.n;
  }
};
std::mutex g_mu;
std::map<GraphKey, hipGraphExec_t> g_graphs;

int enqueue_all(double* A, int ld, const double* b, double* x, int* info, hipStream_t s) {
  const int nb = ld / NB;
  for (int k = 0; k < nb; ++k) {
    const int k0 = k * NB;
    const int m = nb - k;
    hipLaunchKernelGGL(chol_step, dim3(m + (k > 0 ? m * (m - 1) / 2 : 0)), dim3(256), 0, s, A, ld, k0, b, x, info);
  }
  for (int k = nb - 1; k >= 0; --k) hipLaunchKernelGGL(chol_back, dim3(k > 0 ? k : 1), dim3(256), 0, s, (const double*)A, ld, k * NB, x);
  return hipGetLastError() == hipSuccess ? PPO_BA_OK : PPO_BA_ERR_HIP;
}

}  // namespace

// A: ld x ld column-major with ld a multiple of 64 (chol_padded_ld), lower triangle valid, pad diagonal = 1.
int chol_factor_solve(double* A, int ld, const double* b, double* x, int* info, double* /*workspace*/, hipStream_t stream) {
  if (ld % NB != 0) return PPO_BA_ERR_ARG;
  if (ld == 0) return PPO_BA_OK;
  static const bool use_graph = []() { const char* e = getenv("PPO_BA_CHOL_NOGRAPH"); return !(e && e[0] == '1'); }();
  if (!use_graph) return enqueue_all(A, ld, b, x, info, stream);
  hipGraphExec_t exec = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    const GraphKey key{A, b, x, info, ld};
    auto it = g_graphs.find(key);
    if (it == g_graphs.end()) {
      hipGraph_t graph = nullptr;
      if (hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal) != hipSuccess) return PPO_BA_ERR_HIP;
      const int rc = enqueue_all(A, ld, b, x, info, stream);
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

int chol_padded_ld(int n) { return (n + NB - 1) / NB * NB; }
size_t chol_workspace_doubles(int) { return 0; }
int chol_default_backend() { return PPO_SOLVER_NATIVE; }

}  // namespace ppo
