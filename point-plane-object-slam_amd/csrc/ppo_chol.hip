// ppo_chol.hip -- native FP64 Cholesky solve of the dense reduced system (replaces LinearSolverDense /
// Eigen::LDLT of Thirdparty/g2o/g2o/solvers/linear_solver_dense.h:65-113 on the device).
//
// Why not rocSOLVER: at n ~ 1.6k dpotrf + dpotrs spend 5 ms in latency-bound single-workgroup kernels
// (profiles/r01_v1_kernel_stats.txt).  This solver is built for that regime:
//   * right-looking, 64-wide panels, the matrix padded to a multiple of 64 (pad diagonal = 1);
//   * chol_panel(k): every workgroup factors the 64x64 diagonal block redundantly in LDS -- four 16x16 wave-level
//     register factorizations (pivot broadcast by v_readlane, no barrier inside), 16-wide substitutions with one row per
//     lane, and v_mfma_f64_16x16x4 for every 16x16x16 block update -- then workgroup b solves its own 64-row block
//     against it; the right-hand side rides along as a 65th row, so the forward substitution costs nothing extra;
//   * chol_update(k): one 64x64 tile of the trailing lower triangle per workgroup, 64 FP64 MFMAs per wave;
//   * chol_back(k): blocked backward substitution, the 64-vector held one entry per lane;
//   * the 3 x n/64 launches are captured once into a hipGraph and replayed per solve.
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

// One wave factors the 16x16 block sL[o..o+16)[o..o+16) in place (lower) with the rows held in registers: lane r < 16
// holds row r.  Lanes 16..31 carry the rows of an identity matrix through the SAME column operations (the Cholesky
// elimination is a right-multiplication by L^-T), so they end up holding L^-T row by row at no extra instruction:
// sT[i][c] = (L^-T)[i][c].  Returns the 1-based local index of the first non-positive pivot, 0 if none.
__device__ __forceinline__ int potf2_16(double (*sL)[LP], double (*sT)[17], double* sInv, int o) {
  const int lane = threadIdx.x & 63, r = lane & 15;
  const bool ident = (lane & 16) != 0;
  double a[SB];
#pragma unroll
  for (int c = 0; c < SB; ++c) a[c] = ident ? (c == r ? 1.0 : 0.0) : sL[o + r][o + c];
  int bad = 0;
  // Column sweep.  Measured on gfx950 (tools/chol_probe.hip): this wave is ISSUE-bound (~4.5 cycles per instruction,
  // ~60 instructions per column: 2 v_readlane + 1 v_fma per rank-1 column update), not latency-bound -- interleaving the
  // pivot chain with the updates by hand did not help, so the loop stays in its plain form; one scheduling barrier per
  // column keeps the broadcasts from piling up in SGPRs (129 SGPR spills without it).
#pragma unroll
  for (int j = 0; j < SB; ++j) {
    const double d = bcast(a[j], j);
    if (!(d > 0.0) && bad == 0) bad = j + 1;
    // 1/sqrt(d): v_rsq_f64 (measured 5e-8 relative) + two Newton steps y <- y (1.5 - (d/2) y^2): 3.8e-15, then 2.2e-16
    const double h = 0.5 * d;
    double y = __builtin_amdgcn_rsq(d);
    y = y * __builtin_fma(-h, y * y, 1.5);
    y = y * __builtin_fma(-h, y * y, 1.5);
    const double l = a[j] * y;
    a[j] = l;
    if (lane == j) sInv[o + j] = y;
#pragma unroll
    for (int c = j + 1; c < SB; ++c) a[c] = __builtin_fma(-l, bcast(l, c), a[c]);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (lane < SB) {
#pragma unroll
    for (int c = 0; c < SB; ++c)
      if (c <= r) sL[o + r][o + c] = a[c];
  } else if (lane < 2 * SB) {
#pragma unroll
    for (int c = 0; c < SB; ++c) sT[r][c] = a[c];
  }
  return bad;
}

// Factor the 64x64 block held in sL (lower part valid) in place; sT[kk] receives L_kk^-T of the four 16x16 diagonal
// blocks.  All 256 threads participate.  Returns (in every thread) the 1-based index of the first bad pivot or 0.
__device__ int factor_64(double (*sL)[LP], double (*sT)[SB][17], double* sInv, int* sBad) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (tid == 0) *sBad = 0;
  __syncthreads();
#pragma unroll
  for (int kk = 0; kk < NB / SB; ++kk) {
    const int o = kk * SB;
    if (wave == 0) {
      const int bad = potf2_16(sL, sT[kk], sInv, o);
      if (lane == 0 && bad && *sBad == 0) *sBad = o + bad;
    }
    __syncthreads();
    STAMP(8 + 3 * kk);
    if (kk == NB / SB - 1) break;
    // sub-diagonal tiles of this block column: X = B L^-T, one 16x16 tile per wave
    if (kk + 1 + wave < NB / SB) mfma_mul_16(sL, (kk + 1 + wave) * SB, o, sL, (kk + 1 + wave) * SB, o, sT[kk]);
    __syncthreads();
    STAMP(9 + 3 * kk);
    // trailing 16x16 tiles (bi >= bj > kk): A[bi][bj] -= X_bi X_bj^T, tiles dealt round-robin to the 4 waves
    int t = 0;
#pragma unroll
    for (int bi = kk + 1; bi < NB / SB; ++bi)
#pragma unroll
      for (int bj = kk + 1; bj <= bi; ++bj, ++t)
        if ((t & 3) == wave) mfma_sub_16(sL, bi * SB, bj * SB, sL, bi * SB, sL, bj * SB, o);
    __syncthreads();
    STAMP(10 + 3 * kk);
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

// ------------------------------------------------------------------------------------------------------------------
// chol_panel: grid = (ld - k0) / 64 workgroups of 256.  Block 0 owns the diagonal block, block b >= 1 the 64 rows below.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void chol_panel(double* __restrict__ A, int ld, int k0, double* __restrict__ rhs, int* __restrict__ info) {
  __shared__ double sL[NB][LP];
  __shared__ double sB[NB + 1][LP];  // row NB carries r_k^T: the forward substitution is one more row of the block solve
  __shared__ double sT[NB / SB][SB][17];
  __shared__ double sInv[NB];
  __shared__ int sBad;
  const int tid = threadIdx.x, b = blockIdx.x, wave = tid >> 6, lane = tid & 63;
  const int row0 = k0 + b * NB;
  STAMP(0);
  {
    double2 td[8], tb[8];
    load_tile(&A[(size_t)k0 * ld + k0], ld, td);
    if (b > 0) load_tile(&A[(size_t)k0 * ld + row0], ld, tb);
    const double rv = (tid < NB) ? rhs[k0 + tid] : 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int p = tid + 256 * i, c = p >> 5, r = 2 * (p & 31);
      sL[r][c] = (r >= c) ? td[i].x : 0.0;
      sL[r + 1][c] = (r + 1 >= c) ? td[i].y : 0.0;
      if (b > 0) { sB[r][c] = tb[i].x; sB[r + 1][c] = tb[i].y; }
    }
    if (tid < NB) sB[NB][tid] = rv;
  }
  __syncthreads();
  STAMP(1);
  const int bad = factor_64(sL, sT, sInv, &sBad);
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
        for (int m = 0; m < SB; ++m) xv = __builtin_fma(sB[NB][o + m], sT[kk][m][lane], xv);
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
        for (int m = 0; m < SB; ++m) sacc = __builtin_fma(sB[NB][o + m], sL[c][o + m], sacc);
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
    if (bad && tid == 0) atomicCAS(info, 0, k0 + bad);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int p = tid + 256 * i, c = p >> 5, r = 2 * (p & 31);
      if (r + 1 >= c) *reinterpret_cast<double2*>(&A[(size_t)(k0 + c) * ld + k0 + r]) = make_double2(sL[r][c], sL[r + 1][c]);
    }
    if (tid < NB) rhs[k0 + tid] = sB[NB][tid];  // y_k
    return;
  }
  const double rold = (tid < NB) ? rhs[row0 + tid] : 0.0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int p = tid + 256 * i, c = p >> 5, r = 2 * (p & 31);
    *reinterpret_cast<double2*>(&A[(size_t)(k0 + c) * ld + row0 + r]) = make_double2(sB[r][c], sB[r + 1][c]);
  }
  if (tid < NB) {  // r_b -= L_bk y_k
    double sacc = 0.0;
#pragma unroll 16
    for (int m = 0; m < NB; ++m) sacc = __builtin_fma(sB[tid][m], sB[NB][m], sacc);
    rhs[row0 + tid] = rold - sacc;
  }
  STAMP(4);
}

// ------------------------------------------------------------------------------------------------------------------
// chol_update: A[i-block][j-block] -= P_i P_j^T for the panel at column k0; one 64x64 tile per workgroup,
// blockIdx.x enumerates the lower-triangular tiles of the trailing matrix.  Wave w owns a 32x32 quadrant as 2x2 MFMA
// tiles computed transposed (A operand = columns, B operand = rows) so that 16 lanes write 16 consecutive rows.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void chol_update(double* __restrict__ A, int ld, int k0) {
  __shared__ __attribute__((aligned(16))) double sP[NB][NB + 4];  // [m][row of block i]
  __shared__ __attribute__((aligned(16))) double sQ[NB][NB + 4];  // [m][row of block j]
  int t = blockIdx.x, bi = 0;
  while ((bi + 1) * (bi + 2) / 2 <= t) ++bi;
  const int bj = t - bi * (bi + 1) / 2;
  const int r0 = k0 + NB + bi * NB, c0 = k0 + NB + bj * NB;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  const int wr = (wave >> 1) * 32, wc = (wave & 1) * 32;
  double2 tp[8], tq[8];
  load_tile(&A[(size_t)k0 * ld + r0], ld, tp);
  load_tile(&A[(size_t)k0 * ld + c0], ld, tq);
  // old values of the output quadrant, fetched while the panel tiles land
  double cold[2][2][4];
#pragma unroll
  for (int jc = 0; jc < 2; ++jc)
#pragma unroll
    for (int ir = 0; ir < 2; ++ir)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg)
        cold[jc][ir][reg] = A[(size_t)(c0 + wc + 16 * jc + lk + 4 * reg) * ld + r0 + wr + 16 * ir + li];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int p = tid + 256 * i, m = p >> 5, r = 2 * (p & 31);
    *reinterpret_cast<double2*>(&sP[m][r]) = tp[i];
    *reinterpret_cast<double2*>(&sQ[m][r]) = tq[i];
  }
  __syncthreads();
  v4d acc[2][2] = {};
#pragma unroll 4
  for (int m0 = 0; m0 < NB; m0 += 4) {
    const double q0 = sQ[m0 + lk][wc + li], q1 = sQ[m0 + lk][wc + 16 + li];  // A operand: i = column of C
    const double p0 = sP[m0 + lk][wr + li], p1 = sP[m0 + lk][wr + 16 + li];  // B operand: j = row of C
    acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(q0, p0, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(q0, p1, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(q1, p0, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(q1, p1, acc[1][1], 0, 0, 0);
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
  double* x;
  int* info;
  int n;
  bool operator<(const GraphKey& o) const {
    if (A != o.A) return A < o.A;
    if (x != o.x) return x < o.x;
    if (info != o.info) return info < o.info;
    return n < o.n;
  }
};
std::mutex g_mu;
std::map<GraphKey, hipGraphExec_t> g_graphs;

int enqueue_all(double* A, int ld, double* x, int* info, hipStream_t s) {
  const int nb = ld / NB;
  if (hipMemsetAsync(info, 0, sizeof(int), s) != hipSuccess) return PPO_BA_ERR_HIP;
  for (int k = 0; k < nb; ++k) {
    const int k0 = k * NB;
    hipLaunchKernelGGL(chol_panel, dim3(nb - k), dim3(256), 0, s, A, ld, k0, x, info);
    const int m = nb - k - 1;
    if (m > 0) hipLaunchKernelGGL(chol_update, dim3(m * (m + 1) / 2), dim3(256), 0, s, A, ld, k0);
  }
  for (int k = nb - 1; k >= 0; --k) hipLaunchKernelGGL(chol_back, dim3(k > 0 ? k : 1), dim3(256), 0, s, (const double*)A, ld, k * NB, x);
  return hipGetLastError() == hipSuccess ? PPO_BA_OK : PPO_BA_ERR_HIP;
}

}  // namespace

// A: ld x ld column-major with ld a multiple of 64 (chol_padded_ld), lower triangle valid, pad diagonal = 1.
int chol_factor_solve(double* A, int ld, double* x, int* info, double* /*workspace*/, hipStream_t stream) {
  if (ld % NB != 0) return PPO_BA_ERR_ARG;
  if (ld == 0) return PPO_BA_OK;
  static const bool use_graph = []() { const char* e = getenv("PPO_BA_CHOL_NOGRAPH"); return !(e && e[0] == '1'); }();
  if (!use_graph) return enqueue_all(A, ld, x, info, stream);
  hipGraphExec_t exec = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    const GraphKey key{A, x, info, ld};
    auto it = g_graphs.find(key);
    if (it == g_graphs.end()) {
      hipGraph_t graph = nullptr;
      if (hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal) != hipSuccess) return PPO_BA_ERR_HIP;
      const int rc = enqueue_all(A, ld, x, info, stream);
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
