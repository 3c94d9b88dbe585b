// ppo_pairs.hpp -- the Schur accumulation by pose pair (S -= W_j (D_l + lambda I)^-1 W_i'), as a device function that both the
// reduction launch (ppo_kernels.hpp: k_reduce_stage3) and the fused launches of the dense factorisation (ppo_chol.hip: chol_step,
// which hides most of the pair reduction in the shadow of its first panels) instantiate.
// Follows Thirdparty/g2o/g2o/core/block_solver.hpp:354-439 (the Schur complement of BlockSolver::buildSystem / solve).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ppo_math.hpp"

namespace ppo {

// ---- plain data (also seen by host-only translation units through ppo_chol.hpp) ----------------------------------------------------------
struct PairChunks {
  const int32_t* start;  // [n_chunks+1] into sj / si
  const int32_t* colj;   // [n_chunks] reduced column of pose j (row block of S)
  const int32_t* coli;   // [n_chunks] reduced column of pose i (column block of S)
  const int32_t* sj;     // [n_entries]
  const int32_t* si;     // [n_entries]
  const int32_t* lm;     // [n_entries] landmark of the pair (saves the dependent slot -> landmark -> (D + lambda I)^-1 gather chain)
  int n_chunks;
};

// LM control on the device (ppo_lm.hpp, LmCtl): a launch queued under epoch `expect` does nothing once the record has moved on.  (Host-side
// carrier only: the kernels take the two members as scalar arguments -- an aggregate argument ends the kernel-argument preload.)
struct LmGate {
  const int* epoch_word;  // &LmCtl::epoch; NULL: not gated
  int expect;
};

// What the pair reduction reads and writes (a subset of Graph: the fused launches of the dense factorisation carry only this)
struct PairCtx {
  const double* linR;  // [ld / 6 + 2][12] rotation + fx, fy, bf at the linearisation point, by reduced column / 6
  const double* F;     // [n_pt_slots][4] factor records of the point slots
  const double* W;     // [n_slots - n_pt_slots][18] coupling blocks of the plane slots
  const double* Dinv;  // [n_lm][6] (D + lambda I)^-1, packed upper
  const double* ul;    // [n_lm][3] (D + lambda I)^-1 b_l
  double* S;           // [ld][ld] column-major, lower triangle
  double* bs;          // [ld]
  int n_points, n_pt_slots, ld;
  // FLY variant (the chunks reduced in the launch that also forms Dinv / ul: they invert their landmarks' blocks themselves)
  const double* D;     // [n_lm][6] packed upper
  const double* bl;    // [n_lm][3]
  double lambda;
};

#ifdef __HIPCC__  // ---- device code ------------------------------------------------------------------------------------------------------
// Sum `NV` per-lane values over the 64 lanes of a wave through LDS: lane t < NV returns the total of value t.
// buf is this wave's [NV][65] scratch.  (Cheaper than 6 ds_bpermute rounds per value, and it leaves value t in lane t,
// which is exactly the lane-contiguous layout the atomics that follow want.)
template <int NV>
__device__ __forceinline__ double wave_transpose_sum(const double* vals, double (*buf)[65]) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int v = 0; v < NV; ++v) buf[v][lane] = vals[v];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  double t = 0.0;
  if (lane < NV) {
#pragma unroll 16
    for (int i = 0; i < 64; ++i) t += buf[lane][i];
  }
  __builtin_amdgcn_wave_barrier();
  return t;
}

// Schur accumulation by POSE PAIR.  At create time every (slot_j, slot_i) pair of one landmark (pose_j >= pose_i) is
// filed under its pose pair (ordered by COLUMN block of S: pose i first, then pose j -- the order in which the dense
// factorisation consumes the columns, see the overlapped reduction in ppo_chol.hip) and each pair's list is cut into chunks; one wave reduces one chunk into 36 register
// accumulators per lane, S(pj, pi) -= sum W_j (D_l + lambda I)^-1 W_i', and issues one atomic per entry of the 6x6 block
// per CHUNK (instead of one per landmark).  Slots are numbered pose-major, so a pair's list walks two contiguous W
// segments in order; the 3x3 inverse of the pair's landmark is an L2-resident gather.  (An earlier version first streamed
// Z = W L^-T to memory in a launch of its own: 73 MB of traffic and 23 us that forming W D^-1 on the fly does not need.)
// The self pairs (slot_j == slot_i) also carry the right-hand side: bs(pj) -= W_j (D_l + lambda I)^-1 b_l.
#ifndef PPO_PAIRS_NO_PREFETCH
#define PPO_PAIRS_PREFETCH 1  // the next trip's gathers in flight during this trip's arithmetic: 43.8 -> 42.1 us (234 VGPRs, still 2 waves per SIMD)
#endif
// One workgroup of WAVES waves reduces WAVES chunks of [chunk_lo, chunk_hi); `add(p, v)` is the accumulation (*p += v across waves).
// (D + lambda I)^-1 of a 3x3 landmark block (packed upper in, packed upper out) through its Cholesky factor, reciprocal square roots with
// one third-order correction (2.2e-16) instead of square roots and divisions
__device__ __forceinline__ double pairs_rsqrt(double x) {
  const double y0 = __builtin_amdgcn_rsq(x);
  const double e = __builtin_fma(-(x * y0), y0, 1.0);
  return (x > 0.0) ? __builtin_fma(y0, e * __builtin_fma(0.375, e, 0.5), y0) : __builtin_nan("");
}
__device__ __forceinline__ void pairs_inv3(double d00, double d01, double d02, double d11, double d12, double d22, double lambda, double (&v)[6]) {
  d00 += lambda; d11 += lambda; d22 += lambda;
  const double i00 = pairs_rsqrt(d00), l10 = d01 * i00, l20 = d02 * i00;
  const double i11 = pairs_rsqrt(d11 - l10 * l10), l21 = (d12 - l20 * l10) * i11;
  const double i22 = pairs_rsqrt(d22 - l20 * l20 - l21 * l21);
  const double i10 = -l10 * i00 * i11, i21 = -l21 * i11 * i22, i20 = -(l20 * i00 + l21 * i10) * i22;
  v[0] = i00 * i00 + i10 * i10 + i20 * i20; v[1] = i10 * i11 + i20 * i21; v[2] = i20 * i22;
  v[3] = i11 * i11 + i21 * i21; v[4] = i21 * i22; v[5] = i22 * i22;
}
// FLY: the landmark blocks are inverted here from D and lambda (and (D + lambda I)^-1 b_l formed from b_l) instead of being read from Dinv / ul
template <int WAVES, bool FLY, class Add>
__device__ __forceinline__ void schur_pairs_body(const PairCtx& g, const PairChunks& pc, int chunk_lo, int chunk_hi, int bid, int nblocks, double (*s_buf)[21][65], Add add) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // One wave per chunk of one pose pair.  Chunks are ordered by (pose i, pose j) and every Z record is needed by ~7
  // pairs, so the launch is made XCD-aware: workgroup b runs on XCD b % 8 (observed dispatch rule, used for speed only),
  // and each XCD is given a CONTIGUOUS range of chunks -- the Z segments of "its" keyframes then stay in its 4 MB L2
  // instead of being streamed from HBM once per pair (264 MB -> ~36 MB at the BASELINE size).
  const int per_xcd = (nblocks + 7) >> 3;
  const int vb = (bid & 7) * per_xcd + (bid >> 3);
  const int t = __builtin_amdgcn_readfirstlane(chunk_lo + vb * WAVES + wave);  // (uniform: the chunk header and the keyframe data become scalar loads)
  if (t >= chunk_hi) return;
  const int e1 = pc.start[t + 1];
  const int colj = pc.colj[t], coli = pc.coli[t];
  // rotation + fx, fy, bf of both keyframes at the linearisation point (left there by the pose side of the linearisation):
  // uniform over the wave, kept in scalar registers
  auto uni = [](double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
  };
  // plan entries of the first trip, requested together with the keyframe data below (both hang on the chunk header only);
  // inside the loop the next trip's entries are requested before this trip's arithmetic.  (clamped indices: unconditional loads)
  int e = pc.start[t] + lane;
  int sj, si, l;
  {
    const int ec = e < e1 ? e : (e1 > 0 ? e1 - 1 : 0);
    sj = pc.sj[ec]; si = pc.si[ec]; l = pc.lm[ec];
  }
  double Rj[12], Ri[12];
  {
    const double* rj = &g.linR[(size_t)(colj / 6) * 12];
    const double* ri = &g.linR[(size_t)(coli / 6) * 12];
#pragma unroll
    for (int k = 0; k < 12; ++k) { Rj[k] = uni(rj[k]); Ri[k] = uni(ri[k]); }
  }
  const FactorRec* const F4 = reinterpret_cast<const FactorRec*>(g.F);
  // vals[6 r + c] = element (row c of pose j, column r of pose i) so that consecutive values are consecutive in memory
  double vals[42];
#pragma unroll
  for (int k = 0; k < 42; ++k) vals[k] = 0.0;
#ifdef PPO_PAIRS_PREFETCH
  // (variant: the gathers of the NEXT trip -- (D + lambda I)^-1 and the two factor records -- are requested before this trip's arithmetic)
  double2 p01, p23, p45;
  FactorRec pfj{}, pfi{};
  {
    const double2* I2 = reinterpret_cast<const double2*>(&(FLY ? g.D : g.Dinv)[(size_t)l * 6]);
    p01 = I2[0]; p23 = I2[1]; p45 = I2[2];
    if (l < g.n_points) { pfj = F4[sj]; pfi = F4[si]; }
  }
#endif
  for (; e < e1; e += 64) {
#ifdef PPO_PAIRS_PREFETCH
    const double2 i01 = p01, i23 = p23, i45 = p45;
    const FactorRec fj_cur = pfj, fi_cur = pfi;
#else
    const double2* I2 = reinterpret_cast<const double2*>(&(FLY ? g.D : g.Dinv)[(size_t)l * 6]);  // 48-byte records: three 16-byte loads
    const double2 i01 = I2[0], i23 = I2[1], i45 = I2[2];
#endif
    double v00 = i01.x, v01 = i01.y, v02 = i23.x, v11 = i23.y, v12 = i45.x, v22 = i45.y;
    if constexpr (FLY) {
      double vi[6];
      pairs_inv3(v00, v01, v02, v11, v12, v22, g.lambda, vi);
      v00 = vi[0]; v01 = vi[1]; v02 = vi[2]; v11 = vi[3]; v12 = vi[4]; v22 = vi[5];
    }
    const int en = e + 64 < e1 ? e + 64 : e1 - 1;
    const int sj_n = pc.sj[en], si_n = pc.si[en], l_n = pc.lm[en];
#ifdef PPO_PAIRS_PREFETCH
    if (!FLY && e + 64 < e1) {  // (uniform: the chunk has another trip.  FLY: no prefetch -- with the inversion's registers on top the wave would need
                                //  more than 256 and the launch would hold half as many workgroups; the gathers are requested at the end of the trip)
      const double2* J2 = reinterpret_cast<const double2*>(&(FLY ? g.D : g.Dinv)[(size_t)l_n * 6]);
      p01 = J2[0]; p23 = J2[1]; p45 = J2[2];
      if (l_n < g.n_points) { pfj = F4[sj_n]; pfi = F4[si_n]; }
    }
#endif
    // S(j, i) -= L T Y' with  points: L = M_j, Y = M_i, T = R_j (D + lambda I)^-1 R_i'   planes: L = W_j, Y = W_i, T = (D + lambda I)^-1
    double L[18], y[18], T[9], w[18];
    const bool self = sj == si;
    double u0 = 0.0, u1 = 0.0, u2 = 0.0;  // self pairs: (D + lambda I)^-1 b_l, rotated like T
    if (self) {
      if constexpr (FLY) {
        const double b0 = g.bl[(size_t)l * 3], b1 = g.bl[(size_t)l * 3 + 1], b2 = g.bl[(size_t)l * 3 + 2];
        u0 = v00 * b0 + v01 * b1 + v02 * b2; u1 = v01 * b0 + v11 * b1 + v12 * b2; u2 = v02 * b0 + v12 * b1 + v22 * b2;
      } else {
        u0 = g.ul[(size_t)l * 3]; u1 = g.ul[(size_t)l * 3 + 1]; u2 = g.ul[(size_t)l * 3 + 2];
      }
    }
    if (l < g.n_points) {
#ifdef PPO_PAIRS_PREFETCH
      const FactorRec fj = fj_cur, fi = fi_cur;
#else
      const FactorRec fj = F4[sj], fi = F4[si];
#endif
      const bool st_rows = __any(fj.ws < 0.0 || fi.ws < 0.0);
      factor_M(fj, Rj[9], Rj[10], Rj[11], st_rows, L);
      factor_M(fi, Ri[9], Ri[10], Ri[11], st_rows, y);
      double A[9];
#pragma unroll
      for (int r = 0; r < 3; ++r) {  // A = R_j (D + lambda I)^-1
        A[3 * r] = __builtin_fma(Rj[3 * r], v00, __builtin_fma(Rj[3 * r + 1], v01, Rj[3 * r + 2] * v02));
        A[3 * r + 1] = __builtin_fma(Rj[3 * r], v01, __builtin_fma(Rj[3 * r + 1], v11, Rj[3 * r + 2] * v12));
        A[3 * r + 2] = __builtin_fma(Rj[3 * r], v02, __builtin_fma(Rj[3 * r + 1], v12, Rj[3 * r + 2] * v22));
      }
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
          T[3 * r + c] = __builtin_fma(A[3 * r], Ri[3 * c], __builtin_fma(A[3 * r + 1], Ri[3 * c + 1], A[3 * r + 2] * Ri[3 * c + 2]));
      if (self) {
        const double t0 = u0, t1 = u1, t2 = u2;
        u0 = __builtin_fma(Rj[0], t0, __builtin_fma(Rj[1], t1, Rj[2] * t2));
        u1 = __builtin_fma(Rj[3], t0, __builtin_fma(Rj[4], t1, Rj[5] * t2));
        u2 = __builtin_fma(Rj[6], t0, __builtin_fma(Rj[7], t1, Rj[8] * t2));
      }
    } else {
      const double2* wp = reinterpret_cast<const double2*>(&g.W[(size_t)(sj - g.n_pt_slots) * 18]);
      const double2* yp = reinterpret_cast<const double2*>(&g.W[(size_t)(si - g.n_pt_slots) * 18]);
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        const double2 a = wp[q], b = yp[q];
        L[2 * q] = a.x; L[2 * q + 1] = a.y; y[2 * q] = b.x; y[2 * q + 1] = b.y;
      }
      T[0] = v00; T[1] = v01; T[2] = v02; T[3] = v01; T[4] = v11; T[5] = v12; T[6] = v02; T[7] = v12; T[8] = v22;
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) {  // w = L T
#pragma unroll
      for (int c = 0; c < 3; ++c)
        w[a * 3 + c] = __builtin_fma(L[a * 3], T[c], __builtin_fma(L[a * 3 + 1], T[3 + c], L[a * 3 + 2] * T[6 + c]));
    }
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int c = 0; c < 6; ++c)
        vals[6 * r + c] = __builtin_fma(w[c * 3], y[r * 3], __builtin_fma(w[c * 3 + 1], y[r * 3 + 1], __builtin_fma(w[c * 3 + 2], y[r * 3 + 2], vals[6 * r + c])));
    if (self) {
#pragma unroll
      for (int c = 0; c < 6; ++c) vals[36 + c] += L[c * 3] * u0 + L[c * 3 + 1] * u1 + L[c * 3 + 2] * u2;
    }
    sj = sj_n; si = si_n; l = l_n;
#ifdef PPO_PAIRS_PREFETCH
    if (FLY && e + 64 < e1) {
      const double2* J2 = reinterpret_cast<const double2*>(&g.D[(size_t)l * 6]);
      p01 = J2[0]; p23 = J2[1]; p45 = J2[2];
      if (l < g.n_points) { pfj = F4[sj]; pfi = F4[si]; }
    }
#endif
  }
  const int n = g.ld;
  const bool diag = colj == coli;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const double tot = wave_transpose_sum<21>(&vals[21 * half], s_buf[wave]);
    const int v = 21 * half + lane;
    if (lane < 21 && tot != 0.0) {
      if (v < 36) {
        const int r = v / 6, c = v - 6 * r;
        if (!diag || c >= r) add(&g.S[(size_t)(coli + r) * n + colj + c], -tot);
      } else if (diag) {
        add(&g.bs[colj + (v - 36)], -tot);
      }
    }
  }
}

#endif  // __HIPCC__

}  // namespace ppo
