// ppo_kernels.hpp -- CDNA4 (gfx950, wave64) kernels of one Levenberg-Marquardt iteration.
//
// Launches of one LM iteration (DESIGN.md section 4 has the table with bounds, bytes and measured times).  Independent parts
// are fused HORIZONTALLY: one launch, block ranges select the part, one shared LDS allocation.
//   k_reduce_stage1   9x9 inverse per cuboid | assemble Hpp + lambda I into S | (D_l + lambda I)^-1 per landmark
//   k_reduce_stage3   cuboid edge pairs | S -= W_j (D_l + lambda I)^-1 W_i' by pose pair (wave per pair chunk; W rebuilt from the
//                     32-byte factor records of the two slots, ppo_math.hpp factor_M)
//   [dense Cholesky of S: ppo_chol.hip (chol_step x ld/64, chol_back_all) or rocSOLVER]
//   k_update_all      cuboid back-substitution + update | pose updates | landmark back-substitution + update (matrix-free)
//                     | clearing S for the next reduction
//   k_zero_ranges     the accumulation ranges of the linearisation
//   k_linearize_all   (on the TRIAL state: its sum of rho is the trial's chi2)
//                     numerically differentiated edges (2 NCOL + 1 lanes per plane / bbox / corner / point-cuboid edge: the two
//                     signs of a Jacobian column in neighbouring lanes, central differences through the vertex box-plus exactly
//                     like g2o, delta = 1e-9)
//                     | pose side of the reprojection edges (wave per keyframe segment: factor records of W = Jp' w Jl, Hpp, b_p)
//                     | landmark side (2 lanes per point: D_l, b_l, per-edge chi2, robust sum)
//   k_publish         chi2 / scale / factorisation flag -> mapped host memory
//   k_chi2_all        residual-only pass on the trial state (last trial of a pass only), per-edge chi2 kept (g2o's stored _error)
// The single-part kernels further down (k_lin_points, k_lin_poses, k_schur_pairs, k_backsub, k_chi2_points) wrap the same
// bodies for the measurement hook ppo_ba_time_kernel().
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ppo_math.hpp"
#include "ppo_pairs.hpp"
#include "ppo_lm.hpp"
#include "ppo_dpp.hpp"

namespace ppo {

// ---- device-resident graph -----------------------------------------------------------------------------
struct MonoEdge {  // 48 B
  double u, v, info, huber;
  int32_t pose, slot;
  uint32_t flags;  // bit0 active (level 0), bit1 robust kernel on
  int32_t point;
};
struct StereoEdge {  // 64 B
  double u, v, ur, info, huber;
  int32_t pose, slot;
  uint32_t flags;
  int32_t point;
  int32_t pad[2];
};
struct MonoEdgeP {  // 48 B, pose-major copy of the static part (k_lin_poses streams it)
  double u, v, info, huber;
  int32_t point, eid, slot, pad;  // eid = position in the landmark-major arrays (flags, chi2)
};
struct StereoEdgeP {  // 64 B
  double u, v, ur, info, huber;
  int32_t point, eid, slot;
  int32_t pad[3];
};
struct PlaneEdge {  // 88 B
  double meas[4], info[3], huber;
  int32_t plane, pose, slot;
  uint32_t flags;  // bit0 active, bit1 robust, bits 8..9 kind
  int32_t pad[2];
};
struct CubEdge {  // bbox (dim 4) and corner (dim 16) edges share one record; so does the 9-D camera-object edge EdgeSE3Cuboid (dim 9,
                   // meas = the measured cuboid qx qy qz qw tx ty tz sx sy sz, K unused), stored behind the corner edges
  double meas[16], info[16], huber, K[9];
  int32_t pose, cuboid;
  uint32_t flags;
  int32_t dim;
};
struct PtCubEdge {
  double margin, prior[3], info[3], huber;
  int32_t cuboid, p0, p1;
  uint32_t flags;
};

enum { F_ACTIVE = 1u, F_ROBUST = 2u };
enum { ACC_CHI2 = 0, ACC_SCALE = 1, ACC_STOP = 2, ACC_MAXDIAG = 3, ACC_DETBAD = 4, ACC_DETRANGE = 5, ACC_N = 8 };  // ACC_STOP: sharded runs, the ranks' stop votes (summed)
// chi2 and the gain-ratio scale are sums over ALL workgroups of a launch: 1 300 atomics on one address retire one after the other and
// were 4 us of the 31 us fused linearisation (PPO_BA_LIN_DBG=4).  Workgroups add to one of kAccParts partial sums, 128 bytes apart,
// behind the accumulator block; whoever reads the block folds them in first (fold_acc_parts: k_publish, k_fold_acc).
constexpr int kAccParts = 64, kAccPartStride = 16, ACC_TOTAL = ACC_N + kAccParts * kAccPartStride;
static_assert(ACC_N == 8 && kAccParts == 64 && kAccPartStride == 16, "fixed_add tells a chi2 term by its address inside the accumulator block");

struct State {       // one copy of all estimates
  double* pose_qt;   // [Np][7]
  double* pose_rt;   // [Np][12] rotation matrix (row-major) + translation, derived from pose_qt
  double* point;     // [Npt][3]
  double* plane;     // [Npl][4]
  double* cub;       // [Nc][10] qx qy qz qw tx ty tz sx sy sz
};

struct Graph {
  // (first: inside the kernel-argument registers that are preloaded, so the gate costs one scalar load, not two in a row)
  const LmCtl* ctl;       // LM control on the device: the kernels of a trial read lambda here and do nothing unless ctl->epoch == epoch
  int epoch;              // (NULL: the host passes lambda and every launch is live)
  int n_poses, n_points, n_planes, n_cuboids, n_pose_free, n_red, n_lm, n_slots;
  int ld;  // leading dimension of S (n_red rounded up to the Cholesky panel width; pad rows/cols are identity)
  int n_mono, n_stereo, n_pl, n_bbox, n_corner, n_ptcub;
  int n_corner2d;  // corner[0 .. n_corner2d) are corner-projection edges, corner[n_corner2d .. n_corner) 9-D camera-object edges (dim 9)
  const double* pose_cam;      // [Np][5]
  const int32_t* pose_col;     // [Np] column of the pose block in the reduced system, -1 fixed / unused
  const int32_t* pose_free_id; // [Np] dense id among free poses (index into HB), -1 otherwise
  const int32_t* cub_col;      // [Nc]
  const uint8_t* point_fixed;  // [Npt]
  const uint8_t* cub_flags;    // [Nc]
  const double* cub_fixedscale;  // [Nc][3]
  const int32_t* mo_off;       // [Npt+1] mono edges of a point (edges sorted by point)
  const int32_t* st_off;       // [Npt+1]
  MonoEdge* mono;
  StereoEdge* stereo;
  PlaneEdge* pl;
  CubEdge* bbox;
  CubEdge* corner;
  PtCubEdge* ptcub;
  const double* ptcub_pts;
  // (landmark, pose) slots, numbered POSE-MAJOR: point slots sorted by (pose, point), then plane slots by (pose, plane)
  const int32_t* slot_col;     // [n_slots] reduced column of the slot's pose
  const int32_t* slot_lm;      // [n_slots] landmark (point id, or n_points + plane id)
  const int32_t* pl_slot_off;  // [n_planes+1] slots of a plane ...
  const int32_t* pl_slots;     // ... listed here
  uint32_t* mono_flags;        // [n_mono]   bit0 active (level 0), bit1 robust kernel on -- the single source of truth
  uint32_t* stereo_flags;      // [n_stereo]
  const MonoEdgeP* monoP;      // pose-major static copies
  const StereoEdgeP* stereoP;
  // per-edge outputs (stored-_error semantics)
  double* mono_chi2;  double* stereo_chi2;  double* pl_chi2;
  double* bbox_chi2;  double* bbox_norm;    double* corner_chi2;  double* corner_norm;  double* ptcub_chi2;
  // linear system
  double* HB;    // [n_pose_free][27]: 21 packed lower entries of the 6x6 block, then b (6)
  double* HC;    // [Nc][54]: 45 packed lower entries of the 9x9 block, then b (9)
  double* D;     // [n_lm][6] packed upper (00,01,02,11,12,22)
  double* bl;    // [n_lm][3]
  double* W;     // [n_slots - n_pt_slots][18] coupling blocks of the PLANE slots (slot s at (s - n_pt_slots) * 18), pose rows x landmark cols
  double* F;     // [n_pt_slots][4] factor record of a point slot's reprojection edge: x/z, y/z, 1/z in the camera frame and the weight
                 // rho1 * info (negated for a stereo edge, 0 for an inactive one): W = M(record, intrinsics) * R is rebuilt where it is used
  double* linR;  // [ld / 6 + 2][12] rotation (9) + fx, fy, bf of the linearisation point of the pose with reduced column c, at c / 6
  int n_pt_slots;
  double* Wpc_bbox;    // [n_bbox][54] pose rows x cuboid cols
  double* Wpc_corner;  // [n_corner][54]
  double* Dinv;  // [n_lm][6] (D + lambda I)^-1, packed upper
  double* ul;    // [n_lm][3] (D + lambda I)^-1 b_l
  double* S;     // [n_red][n_red] column-major, lower triangle
  double* bs;    // [n_red]
  double* xr;    // [n_red]
  double* acc;   // [ACC_TOTAL]: ACC_N values, then kAccParts partial sums of (chi2, scale), kAccPartStride doubles apart
  // landmark range owned by this rank (multi-GPU) -- single GPU: [0, n_points) and [0, n_planes)
  // cuboid elimination (second-level Schur complement on the 9x9 cuboid blocks; the dense system is poses only)
  int elim_cub;
  const int32_t* cub_edge_off;  // [Nc+1] bbox / corner edges of a cuboid ...
  const int32_t* cub_edges;     // ... as (edge index << 1) | is_corner
  const int32_t* cub_edge_col;  // [n_cub_edges] reduced column of the listed edge's keyframe (-1: fixed)
  const int32_t* cp_cub;        // [n_cub_pairs] cuboid of an (edge a >= edge b) pair
  const int32_t* cp_a;          // [n_cub_pairs] edge a as (edge index << 1) | is_corner
  const int32_t* cp_b;
  const int32_t* cp_off;        // [Nc+1] pairs of a cuboid (they are listed cuboid by cuboid)
  const int32_t* cp_cola;       // [n_cub_pairs] reduced column of edge a's keyframe (-1: fixed)
  const int32_t* cp_colb;
  int n_cub_pairs;
  double* cub_Linv;             // [Nc][81] inverse Cholesky factor of Hcc + lambda I (row-major, lower)
  double* cub_Dinv;             // [Nc][81] (Hcc + lambda I)^-1, full symmetric
  double* xc;                   // [Nc][9]  cuboid update
  int pt_first, pt_last, plane_first, plane_last;
  int owns_reduced;  // 1 on the rank that adds Hcc / lambda / cuboid terms
  int jac;           // options.jacobian_mode: 0 numeric (central differences, what the reference does), 1 analytic for the plane and the
                     // cuboid-projection edges (ppo_math.hpp: plane_edge_analytic, cuboid_corner_analytic); the point-cuboid edge stays numeric
  int det;           // options.deterministic: every cross-wave accumulation is an INTEGER atomic on a 2^-24 fixed-point image of the
                     // addend (exact, hence independent of the order the atomics arrive in); k_fix_to_double turns the ranges back into doubles
  const double* hb_diag;  // sharded runs, first iteration: diagonal of Hpp summed over ranks (NULL: read it from HB)
  // deterministic mode with the LM control on the device: a linearisation that RUNS (its epoch is live) leaves its tag here; the decide launch
  // behind it converts the ranges it summed into back to doubles only when it finds that tag (k_decide_fix) -- a launch that found its epoch gone
  // has written nothing, and its target buffer may hold an earlier linearisation that is doubles already
  unsigned long long* lin_tag_word;
  unsigned long long lin_tag;
  int lin_raw;  // deterministic mode: the accumulation ranges of the linearisation this launch reads still hold fixed-point integers (their conversion
                // rides in the publish launch behind: k_publish_fix) -- k_maxdiag converts what it reads
#ifdef PPO_BA_TEST_HOOKS
  int dbg;                     // development aid (PPO_BA_LIN_DBG): bit 0 numeric classes skip their scatter, bit 1 plain stores instead of atomics
  unsigned long long* stamps;  // development aid (PPO_BA_LIN_STAMPS): [grid][8] wall-clock stamps of the fused linearisation's workgroups
#endif
};
#ifdef PPO_BA_TEST_HOOKS
#define LIN_STAMP(g_, k_) do { if ((g_).stamps && threadIdx.x == 0) (g_).stamps[(size_t)blockIdx.x * 8 + (k_)] = wall_clock64(); } while (0)
#else
#define LIN_STAMP(g_, k_) ((void)0)
#endif

__device__ __forceinline__ bool lm_dead(const Graph& g) { return g.ctl != nullptr && g.ctl->epoch != g.epoch; }
__device__ __forceinline__ double lm_lambda(const Graph& g, double host_lambda) { return g.ctl != nullptr ? g.ctl->lambda : host_lambda; }

__device__ __forceinline__ int packed_lower(int r, int c) { return r * (r + 1) / 2 + c; }  // r >= c

__device__ __forceinline__ double wave_sum(double v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
// block-wide sum to one global atomic (blockDim.x multiple of 64, <= 1024)
// ---- deterministic mode (options.deterministic = 1) -------------------------------------------------------------------
// Floating-point atomics make every sum that crosses a wave depend on arrival order (chi2 repeats to 1e-9 from run to run).
// In deterministic mode the same atomics add INTEGERS: the addend is rounded to a multiple of 2^-24 and added as a 64-bit
// two's-complement number, which is exact and therefore order-independent; the accumulation ranges are converted back to
// doubles by k_fix_to_double before anybody reads them.  Range +-5.5e11 (the largest entries, diagonal of H_pp / S, are ~1e9),
// resolution 6e-8 absolute, i.e. <= 1e-15 of the norm of the matrices that are factored.  An addend that cannot be represented is left out and
// flagged.  A NON-FINITE term of chi2 raises ACC_DETBAD: the sum would be NaN in the reference's arithmetic and is treated as NaN by whoever reads
// the flag (chi2 of the estimate NaN: the window freezes exactly as it does in the reference, levenberg.cpp:113-149 with every comparison on NaN
// false).  Every other term that does not fit -- finite and out of range, or a non-finite entry of a matrix, a right-hand side or the scale of the
// gain ratio -- raises ACC_DETRANGE: the system is not usable, reported like a factorisation that failed (what a NaN in S is to the reference's
// LDLT, linear_solver_dense.h:108-112).
constexpr double kFixScale = 16777216.0, kFixInv = 1.0 / 16777216.0;
constexpr int kAccPartsFirst = 8;  // (= ACC_N, declared below with the partial sums)
__device__ __forceinline__ void fixed_add(double* p, double v, double* acc) {  // acc: the accumulator block (flags; and p inside it <=> a chi2 / scale term)
  if (!(fabs(v) < 4.0e11)) {
    const long off = p - acc;
    const bool chi2_term = off >= kAccPartsFirst && off < kAccPartsFirst + 64 * 16 && ((off - kAccPartsFirst) & 15) == ACC_CHI2;  // (kAccParts x kAccPartStride)
    acc[(chi2_term && !isfinite(v)) ? ACC_DETBAD : ACC_DETRANGE] = 1.0;
    return;
  }
  atomicAdd(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double2ll_rn(v * kFixScale));
}
#ifdef PPO_BA_TEST_HOOKS  // PPO_BA_LIN_DBG bit 1: plain (racy) stores instead of atomics -- timing experiments only
#define acc_add(g_, p_, v_) do { if ((g_).det) fixed_add((p_), (v_), (g_).acc); else if ((g_).dbg & 2) *(p_) = (v_); else atomicAdd((p_), (v_)); } while (0)
#else
#define acc_add(g_, p_, v_) do { if ((g_).det) fixed_add((p_), (v_), (g_).acc); else atomicAdd((p_), (v_)); } while (0)
#endif

// where this workgroup (wave) adds its share of acc[which], which = ACC_CHI2 or ACC_SCALE
__device__ __forceinline__ double* acc_part(const Graph& g, int which, unsigned salt) {
  return g.acc + ACC_N + ((blockIdx.x + salt) & (kAccParts - 1)) * kAccPartStride + which;
}
// (one wave) partial sums -> acc[0..1], partial sums cleared.  Deterministic mode adds the fixed-point images as integers.
__device__ __forceinline__ void fold_acc_parts(double* acc, int det) {
  const int t = threadIdx.x & 63;
  double* p = acc + ACC_N + t * kAccPartStride;
  if (det) {
    long long c = __double_as_longlong(p[0]), sc = __double_as_longlong(p[1]);
    for (int off = 32; off > 0; off >>= 1) { c += __shfl_down(c, off, 64); sc += __shfl_down(sc, off, 64); }
    if (t == 0) {
      acc[ACC_CHI2] = __longlong_as_double(__double_as_longlong(acc[ACC_CHI2]) + c);
      acc[ACC_SCALE] = __longlong_as_double(__double_as_longlong(acc[ACC_SCALE]) + sc);
    }
  } else {
    double c = p[0], sc = p[1];
    for (int off = 32; off > 0; off >>= 1) { c += __shfl_down(c, off, 64); sc += __shfl_down(sc, off, 64); }
    if (t == 0) { acc[ACC_CHI2] += c; acc[ACC_SCALE] += sc; }
  }
  p[0] = 0.0;
  p[1] = 0.0;
}
// the same inside the launch whose workgroups are still adding elsewhere (lm_tail): the partial sums are read and cleared with device-scope
// operations, at the memory side where the atomics that formed them were performed
__device__ __forceinline__ void fold_acc_parts_coherent(double* acc) {
  const int t = threadIdx.x & 63;
  double* p = acc + ACC_N + t * kAccPartStride;
  double c = __hip_atomic_load(&p[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), sc = __hip_atomic_load(&p[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (int off = 32; off > 0; off >>= 1) { c += __shfl_down(c, off, 64); sc += __shfl_down(sc, off, 64); }
  if (t == 0) {
    __hip_atomic_store(&acc[ACC_CHI2], __hip_atomic_load(&acc[ACC_CHI2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&acc[ACC_SCALE], __hip_atomic_load(&acc[ACC_SCALE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + sc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __hip_atomic_store(&p[0], 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(&p[1], 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void lds_wave_fence() {  // make this wave's LDS writes visible to its own other lanes
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// one atomic per WAVE, no barrier: waves that are done leave (their slots free up for the workgroups still queued)
__device__ __forceinline__ void wave_atomic_add(const Graph& g, double v, int which) {
  v = wave_sum(v);
#ifdef PPO_BA_TEST_HOOKS
  if (g.dbg & 4) v = 0.0;
#endif
  if ((threadIdx.x & 63) == 0 && v != 0.0) acc_add(g, acc_part(g, which, 17u * (threadIdx.x >> 6)), v);
}
__device__ __forceinline__ void block_atomic_add(const Graph& g, double v, int which, double* s_red) {
  double* const dst = acc_part(g, which, 0);
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) s_red[wid] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += s_red[i];
#ifdef PPO_BA_TEST_HOOKS
    if (g.dbg & 4) t = 0.0;  // timing experiment: no same-address atomics from the workgroups
#endif
    if (t != 0.0) acc_add(g, dst, t);
  }
  // (no barrier behind it: every caller ends with this call, nothing reuses s_red)
}
__device__ __forceinline__ void atomic_max_pos(double* dst, double v) {  // v >= 0: IEEE order == integer order
  atomicMax(reinterpret_cast<unsigned long long*>(dst), (unsigned long long)__double_as_longlong(v));
}

__device__ __forceinline__ Pose load_pose(const double* qt) { return Pose{Quat{qt[0], qt[1], qt[2], qt[3]}, Vec3{qt[4], qt[5], qt[6]}}; }
__device__ __forceinline__ void store_pose(double* qt, double* rt, const Pose& P) {
  qt[0] = P.q.x; qt[1] = P.q.y; qt[2] = P.q.z; qt[3] = P.q.w; qt[4] = P.t.x; qt[5] = P.t.y; qt[6] = P.t.z;
  const Mat3 R = rotmat(P.q);
  for (int i = 0; i < 9; ++i) rt[i] = R.a[i];
  rt[9] = P.t.x; rt[10] = P.t.y; rt[11] = P.t.z;
}
__device__ __forceinline__ Cub load_cub(const double* c) {
  return Cub{Pose{Quat{c[0], c[1], c[2], c[3]}, Vec3{c[4], c[5], c[6]}}, Vec3{c[7], c[8], c[9]}};
}
__device__ __forceinline__ PlaneC load_plane(const double* p) { return PlaneC{p[0], p[1], p[2], p[3]}; }

// (acc_landmark / form_W / acc_pose: the quadratic-form pieces of one reprojection edge live in ppo_math.hpp, where the host
// test build checks them against constructQuadraticForm of the reference's prebuilt libg2o.so)

// Camera table of a workgroup: R, t (12) and fx, fy, cx, cy of every keyframe, 16 doubles each, into LDS.  Eight loads per thread are
// issued before the first store (sixteen measure the same): written as one load + one store per trip the loop waited for memory 13 times in a row
// (200 keyframes, 256 threads) before the workgroup could touch its first edge.
#ifndef PPO_CAM_ROW
#define PPO_CAM_ROW 17  // doubles per keyframe row of the table in LDS: 16 used + 1 of padding.  Lanes gather WHOLE rows by keyframe index; with 128-byte rows every row starts in one of two bank groups (SQ_WAIT_INST_LDS 18 % of the wave cycles of k_update_all).  Measured (tools/r06/ab_libs.sh): k_linearize_all 27.5 -> 24.9 us, k_update_all 18.8 -> 18.2 with 17 (18: 25.0 / 18.2, 20: 25.3 / 18.4)
#endif
constexpr int kCamRow = PPO_CAM_ROW;
__device__ __forceinline__ void stage_camera_table(double* smem, const double* __restrict__ pose_rt, const double* __restrict__ pose_cam, int n_poses) {
  const int n = n_poses * 16;
  for (int base = threadIdx.x; base < n; base += 8 * 256) {
    double v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = base + 256 * k < n ? base + 256 * k : n - 1;  // clamped: the loads stay unconditional
      const int p = i >> 4, q = i & 15;
      v[k] = q < 12 ? pose_rt[p * 12 + q] : pose_cam[p * 5 + (q - 12)];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (base + 256 * k < n) smem[((base + 256 * k) >> 4) * kCamRow + ((base + 256 * k) & 15)] = v[k];
  }
}

// ======================================================================================================
// k_lin_points: landmark side of the reprojection edges, one thread per point.  The camera table (R, t, intrinsics of
// every keyframe: 128 B each) is staged in LDS once per workgroup (USE_LDS); D_l / b_l / W are written exactly once,
// no atomics.  The pose blocks are accumulated by k_lin_poses below.
// ======================================================================================================
template <bool USE_LDS>
__device__ __forceinline__ void lin_points_body(const Graph& g, const State& st, int bid, int nblocks, double* smem, double* s_red) {
  if (USE_LDS) {
    stage_camera_table(smem, st.pose_rt, g.pose_cam, g.n_poses);
    __syncthreads();
  }
#ifndef PPO_LIN_LPP
#define PPO_LIN_LPP 2  // 2 lanes per point: half the workgroups (and camera-table loads) of 4 lanes at ~2.5 edges per lane: 19 vs 27 us alone, 33.0 vs 34.5 us fused
#endif
  constexpr int LPP = PPO_LIN_LPP;  // lanes per point
  const int sub = threadIdx.x % LPP, ppb = blockDim.x / LPP;
  double chi_local = 0.0;
  for (int p = g.pt_first + bid * ppb + threadIdx.x / LPP; p < g.pt_last; p += nblocks * ppb) {
    const Vec3 X = v3(st.point[3 * p], st.point[3 * p + 1], st.point[3 * p + 2]);
    const bool lm_free = !g.point_fixed[p];
    double D[6] = {0, 0, 0, 0, 0, 0}, bl[3] = {0, 0, 0};
    for (int e = g.mo_off[p] + sub; e < g.mo_off[p + 1]; e += LPP) {
      const uint32_t fl = g.mono_flags[e];
      if (!(fl & F_ACTIVE)) continue;
      const MonoEdge ed = g.mono[e];
      Mat3 R; Vec3 t; double fx, fy, cx, cy;
      if (USE_LDS) {
        const double* c = &smem[ed.pose * kCamRow];
#pragma unroll
        for (int k = 0; k < 9; ++k) R.a[k] = c[k];
        t = v3(c[9], c[10], c[11]); fx = c[12]; fy = c[13]; cx = c[14]; cy = c[15];
      } else {
        const double* rt = &st.pose_rt[ed.pose * 12];
#pragma unroll
        for (int k = 0; k < 9; ++k) R.a[k] = rt[k];
        t = v3(rt[9], rt[10], rt[11]);
        const double* cm = &g.pose_cam[ed.pose * 5];
        fx = cm[0]; fy = cm[1]; cx = cm[2]; cy = cm[3];
      }
      double ev[2], Jl[6], Jp[12], zc;
      mono_eval(R, t, X, fx, fy, cx, cy, ed.u, ed.v, ev, Jl, Jp, zc);
      const double chi2 = ev[0] * (ed.info * ev[0]) + ev[1] * (ed.info * ev[1]);
      g.mono_chi2[e] = chi2;
      double rho0, rho1;
      huber(chi2, (fl & F_ROBUST) ? ed.huber : 0.0, rho0, rho1);
      chi_local += rho0;
      const double gr[2] = {-(ed.info * ev[0]) * rho1, -(ed.info * ev[1]) * rho1};
      if (lm_free) acc_landmark<2>(Jl, rho1 * ed.info, gr, D, bl);
    }
    for (int e = g.st_off[p] + sub; e < g.st_off[p + 1]; e += LPP) {
      const uint32_t fl = g.stereo_flags[e];
      if (!(fl & F_ACTIVE)) continue;
      const StereoEdge ed = g.stereo[e];
      const double* rt = &st.pose_rt[ed.pose * 12];
      Mat3 R;
#pragma unroll
      for (int k = 0; k < 9; ++k) R.a[k] = rt[k];
      const double* cm = &g.pose_cam[ed.pose * 5];
      double ev[3], Jl[9], Jp[18], zc;
      stereo_eval(R, v3(rt[9], rt[10], rt[11]), X, cm[0], cm[1], cm[2], cm[3], cm[4], ed.u, ed.v, ed.ur, ev, Jl, Jp, zc);
      double chi2 = 0;
#pragma unroll
      for (int r = 0; r < 3; ++r) chi2 += ev[r] * (ed.info * ev[r]);
      g.stereo_chi2[e] = chi2;
      double rho0, rho1;
      huber(chi2, (fl & F_ROBUST) ? ed.huber : 0.0, rho0, rho1);
      chi_local += rho0;
      const double gr[3] = {-(ed.info * ev[0]) * rho1, -(ed.info * ev[1]) * rho1, -(ed.info * ev[2]) * rho1};
      if (lm_free) acc_landmark<3>(Jl, rho1 * ed.info, gr, D, bl);
    }
#pragma unroll
    for (int off = 1; off < LPP; off <<= 1) {
#pragma unroll
      for (int k = 0; k < 6; ++k) D[k] += __shfl_xor(D[k], off, 64);
#pragma unroll
      for (int k = 0; k < 3; ++k) bl[k] += __shfl_xor(bl[k], off, 64);
    }
    if (lm_free && sub == 0) {
#pragma unroll
      for (int k = 0; k < 6; ++k) g.D[(size_t)p * 6 + k] = D[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) g.bl[(size_t)p * 3 + k] = bl[k];
    }
  }
  (void)s_red;
  wave_atomic_add(g, chi_local, ACC_CHI2);
}
template <bool USE_LDS>
__global__ __launch_bounds__(256) void k_lin_points(Graph g, State st) {
  extern __shared__ double smem[];
  __shared__ double s_red[4];
  lin_points_body<USE_LDS>(g, st, blockIdx.x, gridDim.x, smem, s_red);
}

// (wave_transpose_sum: ppo_pairs.hpp)

// ======================================================================================================
// k_lin_poses: pose side + coupling blocks of the reprojection edges, streamed POSE-MAJOR.  The edges of every free
// keyframe are stored contiguously (static copies monoP / stereoP) and cut into chunks; one thread re-evaluates the
// edges of one chunk against its keyframe (held in registers), writes the factor record of W = Jp' w Jl to consecutive slots and
// accumulates Jp' w Jp / Jp' g in 27 registers: 27 atomics per CHUNK.  Points are gathered from the L2-resident array.
// ======================================================================================================
struct PoseChunks {
  const int32_t* start;  // [n_chunks] range [start, end) in monoP (type 0) or stereoP (type 1)
  const int32_t* end;    // [n_chunks]
  const int32_t* pose;   // [n_chunks] (keyframe index << 1) | type
  int n_chunks;
};
// (factor records: factor_record / factor_M in ppo_math.hpp)
constexpr int kPoseWaveLds = 14 * 65;  // doubles per wave: the 14 x 65 reduction buffer
__device__ __forceinline__ void lin_poses_body(const Graph& g, const State& st, const PoseChunks& pc, int bid, double* s_raw) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double* const wbuf = s_raw + wave * kPoseWaveLds;
  double (*const red)[65] = reinterpret_cast<double (*)[65]>(wbuf);
  const int t = __builtin_amdgcn_readfirstlane(bid * 4 + wave);  // one wave per chunk (uniform index: scalar loads of its header and keyframe): lanes stride the chunk, records are read fully coalesced
  if (t >= pc.n_chunks) return;
  const int code = pc.pose[t], pose = code >> 1;
  // the keyframe is the same for the whole wave: keep R, t and the intrinsics in scalar registers
  auto uni = [](double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
  };
  const double* rt = &st.pose_rt[pose * 12];
  Mat3 R;
#pragma unroll
  for (int k = 0; k < 9; ++k) R.a[k] = uni(rt[k]);
  const Vec3 tr = v3(uni(rt[9]), uni(rt[10]), uni(rt[11]));
  const double* cm = &g.pose_cam[pose * 5];
  const double fx = uni(cm[0]), fy = uni(cm[1]), cx = uni(cm[2]), cy = uni(cm[3]), bf = uni(cm[4]);
  // the Schur pair kernel rebuilds W = M R from the factor records: leave it the linearisation point's rotation and intrinsics
  // (every chunk of a keyframe writes the same twelve values)
  if (lane < 12) g.linR[(size_t)(g.pose_col[pose] / 6) * 12 + lane] = lane < 9 ? rt[lane] : cm[lane == 9 ? 0 : (lane == 10 ? 1 : 4)];
  FactorRec* const F4 = reinterpret_cast<FactorRec*>(g.F);
  double A[28];
#pragma unroll
  for (int k = 0; k < 28; ++k) A[k] = 0.0;
  const int q1 = pc.end[t];
  if (!(code & 1)) {
    for (int q0 = pc.start[t]; q0 < q1; q0 += 64) {
      const int q = q0 + lane;
      const bool in = q < q1;
      MonoEdgeP ed{};
      ed.slot = -1;
      uint32_t fl = 0;
      if (in) { ed = g.monoP[q]; fl = g.mono_flags[ed.eid]; }
      FactorRec rec{0.0, 0.0, 0.0, 0.0};
      if (in && (fl & F_ACTIVE)) {
        const Vec3 X = v3(st.point[3 * ed.point], st.point[3 * ed.point + 1], st.point[3 * ed.point + 2]);
        double ev[2], Jl[6], Jp[12], zc;
        mono_eval(R, tr, X, fx, fy, cx, cy, ed.u, ed.v, ev, Jl, Jp, zc);
        const double chi2 = ev[0] * (ed.info * ev[0]) + ev[1] * (ed.info * ev[1]);
        double rho0, rho1;
        huber(chi2, (fl & F_ROBUST) ? ed.huber : 0.0, rho0, rho1);
        const double w = rho1 * ed.info;
        const double gr[2] = {-(ed.info * ev[0]) * rho1, -(ed.info * ev[1]) * rho1};
        acc_pose<2>(Jp, w, gr, A);
        rec = factor_record(R, tr, X, w);
      }
      if (in && ed.slot >= 0) F4[ed.slot] = rec;  // consecutive lanes, consecutive slots: whole cache lines
    }
  } else {
    for (int q0 = pc.start[t]; q0 < q1; q0 += 64) {
      const int q = q0 + lane;
      const bool in = q < q1;
      StereoEdgeP ed{};
      ed.slot = -1;
      uint32_t fl = 0;
      if (in) { ed = g.stereoP[q]; fl = g.stereo_flags[ed.eid]; }
      FactorRec rec{0.0, 0.0, 0.0, 0.0};
      if (in && (fl & F_ACTIVE)) {
        const Vec3 X = v3(st.point[3 * ed.point], st.point[3 * ed.point + 1], st.point[3 * ed.point + 2]);
        double ev[3], Jl[9], Jp[18], zc;
        stereo_eval(R, tr, X, fx, fy, cx, cy, bf, ed.u, ed.v, ed.ur, ev, Jl, Jp, zc);
        double chi2 = 0;
#pragma unroll
        for (int r = 0; r < 3; ++r) chi2 += ev[r] * (ed.info * ev[r]);
        double rho0, rho1;
        huber(chi2, (fl & F_ROBUST) ? ed.huber : 0.0, rho0, rho1);
        const double w = rho1 * ed.info;
        const double gr[3] = {-(ed.info * ev[0]) * rho1, -(ed.info * ev[1]) * rho1, -(ed.info * ev[2]) * rho1};
        acc_pose<3>(Jp, w, gr, A);
        rec = factor_record(R, tr, X, -w);
      }
      if (in && ed.slot >= 0) F4[ed.slot] = rec;
    }
  }
  double* hb = &g.HB[(size_t)g.pose_free_id[pose] * 27];
#pragma unroll
  for (int half = 0; half < 2; ++half) {  // 27 values in two rounds of 14: lane v ends up with value 14 half + v
    const double tot = wave_transpose_sum<14>(&A[14 * half], red);
    const int v = 14 * half + lane;
    if (lane < 14 && v < 27 && tot != 0.0) acc_add(g, &hb[v], tot);
  }
}
__global__ __launch_bounds__(256) void k_lin_poses(Graph g, State st, PoseChunks pc) {
  __shared__ double s_buf[4 * kPoseWaveLds];
  lin_poses_body(g, st, pc, blockIdx.x, s_buf);
}

// residual-only pass over the reprojection edges (trial state)
constexpr int kLanesPerPoint = 1;  // lanes per point in the residual-only pass (4 measured slower here: 16.3 vs 12.9 us; it pays in
                                   // k_backsub and in the linearisation, whose per-edge work is an order of magnitude heavier)
__device__ __forceinline__ void chi2_points_body(const Graph& g, const State& st, int bid, int nblocks, double* s_red) {
  constexpr int LPP = kLanesPerPoint;
  const int sub = threadIdx.x % LPP, ppb = blockDim.x / LPP;
  double chi_local = 0.0;
  for (int p = g.pt_first + bid * ppb + threadIdx.x / LPP; p < g.pt_last; p += nblocks * ppb) {
    const Vec3 X = v3(st.point[3 * p], st.point[3 * p + 1], st.point[3 * p + 2]);
    for (int e = g.mo_off[p] + sub; e < g.mo_off[p + 1]; e += LPP) {
      const uint32_t fl = g.mono_flags[e];
      if (!(fl & F_ACTIVE)) continue;
      const MonoEdge ed = g.mono[e];
      const double* rt = &st.pose_rt[ed.pose * 12];
      const double* cm = &g.pose_cam[ed.pose * 5];
      Mat3 R; for (int k = 0; k < 9; ++k) R.a[k] = rt[k];
      double ev[2], zc;
      mono_residual(R, v3(rt[9], rt[10], rt[11]), X, cm[0], cm[1], cm[2], cm[3], ed.u, ed.v, ev, zc);
      const double chi2 = ev[0] * (ed.info * ev[0]) + ev[1] * (ed.info * ev[1]);
      g.mono_chi2[e] = chi2;
      double rho0, rho1;
      huber(chi2, (fl & F_ROBUST) ? ed.huber : 0.0, rho0, rho1);
      chi_local += rho0;
    }
    for (int e = g.st_off[p] + sub; e < g.st_off[p + 1]; e += LPP) {
      const uint32_t fl = g.stereo_flags[e];
      if (!(fl & F_ACTIVE)) continue;
      const StereoEdge ed = g.stereo[e];
      const double* rt = &st.pose_rt[ed.pose * 12];
      const double* cm = &g.pose_cam[ed.pose * 5];
      Mat3 R; for (int k = 0; k < 9; ++k) R.a[k] = rt[k];
      double ev[3], zc;
      stereo_residual(R, v3(rt[9], rt[10], rt[11]), X, cm[0], cm[1], cm[2], cm[3], cm[4], ed.u, ed.v, ed.ur, ev, zc);
      double chi2 = 0;
      for (int r = 0; r < 3; ++r) chi2 += ev[r] * (ed.info * ev[r]);
      g.stereo_chi2[e] = chi2;
      double rho0, rho1;
      huber(chi2, (fl & F_ROBUST) ? ed.huber : 0.0, rho0, rho1);
      chi_local += rho0;
    }
  }
  block_atomic_add(g, chi_local, ACC_CHI2, s_red);
}
__global__ __launch_bounds__(256) void k_chi2_points(Graph g, State st) {
  __shared__ double s_red[4];
  chi2_points_body(g, st, blockIdx.x, gridDim.x, s_red);
}

// Optimizer.cc:2739-2769: level 1 if chi2 > threshold or depth <= 0 (depth from the CURRENT estimates), kernel off
__global__ void k_classify_points(Graph g, State st, double th_mono, double th_stereo) {
  for (int p = g.pt_first + blockIdx.x * blockDim.x + threadIdx.x; p < g.pt_last; p += gridDim.x * blockDim.x) {
    const Vec3 X = v3(st.point[3 * p], st.point[3 * p + 1], st.point[3 * p + 2]);
    for (int e = g.mo_off[p]; e < g.mo_off[p + 1]; ++e) {
      const int pose = g.mono[e].pose;
      const double* rt = &st.pose_rt[pose * 12];
      const double z = rt[6] * X.x + rt[7] * X.y + rt[8] * X.z + rt[11];
      uint32_t f = g.mono_flags[e];
      if (g.mono_chi2[e] > th_mono || !(z > 0.0)) f &= ~F_ACTIVE;
      f &= ~F_ROBUST;
      g.mono_flags[e] = f;
    }
    for (int e = g.st_off[p]; e < g.st_off[p + 1]; ++e) {
      const int pose = g.stereo[e].pose;
      const double* rt = &st.pose_rt[pose * 12];
      const double z = rt[6] * X.x + rt[7] * X.y + rt[8] * X.z + rt[11];
      uint32_t f = g.stereo_flags[e];
      if (g.stereo_chi2[e] > th_stereo || !(z > 0.0)) f &= ~F_ACTIVE;
      f &= ~F_ROBUST;
      g.stereo_flags[e] = f;
    }
  }
}
// Optimizer.cc:2771-2814 for the plane and cuboid-projection edges
__global__ void k_classify_small(Graph g, double th_plane, double th_vp, double th_bbox, double th_corner, double th_cub3d) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < g.n_pl) {
    uint32_t f = g.pl[i].flags;
    const int kind = (f >> 8) & 3;
    if (g.pl_chi2[i] > (kind == 0 ? th_plane : th_vp)) f &= ~F_ACTIVE;
    f &= ~F_ROBUST;
    g.pl[i].flags = f;
  }
  if (i < g.n_bbox && g.bbox_norm[i] > th_bbox) g.bbox[i].flags &= ~F_ACTIVE;      // kernel stays on
  if (i < g.n_corner && g.corner_norm[i] > (i < g.n_corner2d ? th_corner : th_cub3d)) g.corner[i].flags &= ~F_ACTIVE;  // (Optimizer.cc:1875-1882 for the 9-D edge)
}

// ======================================================================================================
// Numerically differentiated edges.  A model provides: DIM_MAX, NCOL, and
//   bool  active(i), int dim(i)
//   void  residual(i, column c or -1, sign, delta, st, e)   -- residual with vertex column c perturbed by sign*delta
//   scatter of the quadratic form.
// ======================================================================================================
struct PlaneModel {
  static constexpr int DIM_MAX = 3, NCOL = 9;  // cols 0-2 plane, 3-8 pose
  const Graph& g;
  bool corner;  // unused
  // (sharded runs: a plane edge belongs to the rank that owns its plane landmark)
  __device__ bool active(int i) const {
    const PlaneEdge& ed = g.pl[i];
    return (ed.flags & F_ACTIVE) && ed.plane >= g.plane_first && ed.plane < g.plane_last;
  }
  __device__ int dim(int i) const { return ((g.pl[i].flags >> 8) & 3) == 0 ? 3 : 2; }
  __device__ bool col_free(int i, int c) const { return c < 3 || g.pose_col[g.pl[i].pose] >= 0; }
  __device__ void residual(int i, int c, double step, const State& st, double* e) const {
    const PlaneEdge& ed = g.pl[i];
    PlaneC pl = load_plane(&st.plane[4 * ed.plane]);
    Pose T = load_pose(&st.pose_qt[7 * ed.pose]);
    double u[6];  // (filled by selects: u[c] = step with a run-time c would live in scratch memory)
#pragma unroll
    for (int k = 0; k < 6; ++k) u[k] = (k == (c < 3 ? c : c - 3)) ? step : 0.0;
    if (c >= 0 && c < 3) pl = plane_boxplus(pl, u);
    else if (c >= 3) T = pose_boxplus(T, u);
    const PlaneC local = plane_into_camera(T, pl);
    plane_residual(local, PlaneC{ed.meas[0], ed.meas[1], ed.meas[2], ed.meas[3]}, (ed.flags >> 8) & 3, e);
  }
  __device__ double info(int i, int r) const { return g.pl[i].info[r]; }
  __device__ double huber_delta(int i) const { return (g.pl[i].flags & F_ROBUST) ? g.pl[i].huber : 0.0; }
  __device__ void store_chi2(int i, double chi2, double) const { g.pl_chi2[i] = chi2; }
  // row c of H (ncol entries) and b[c]
  __device__ void scatter(int i, int c, const double* Hrow, double bc) const {
    const PlaneEdge& ed = g.pl[i];
    const int l = g.n_points + ed.plane;
    if (c < 3) {
      for (int c2 = c; c2 < 3; ++c2) acc_add(g, &g.D[(size_t)l * 6 + (c == 0 ? c2 : (c == 1 ? 2 + c2 : 5))], Hrow[c2]);
      acc_add(g, &g.bl[(size_t)l * 3 + c], bc);
    } else {
      const int fid = g.pose_free_id[ed.pose];
      if (fid < 0) return;
      const int a = c - 3;
      double* hb = &g.HB[(size_t)fid * 27];
      for (int c2 = 0; c2 <= a; ++c2) acc_add(g, &hb[packed_lower(a, c2)], Hrow[3 + c2]);
      acc_add(g, &hb[21 + a], bc);
      if (ed.slot >= 0)
        for (int c2 = 0; c2 < 3; ++c2) acc_add(g, &g.W[(size_t)(ed.slot - g.n_pt_slots) * 18 + a * 3 + c2], Hrow[c2]);
    }
  }
};

template <bool CORNER>
struct CubProjModelT {  // bbox (CORNER = false, 4 residual rows) or corner edges (16 rows): cols 0-5 pose, 6-14 cuboid
  static constexpr int DIM_MAX = CORNER ? 16 : 4, NCOL = 15;
  static constexpr bool corner = CORNER;
  const Graph& g;
  bool unused;
  __device__ const CubEdge& E(int i) const { return corner ? g.corner[i] : g.bbox[i]; }
  __device__ bool active(int i) const { return E(i).flags & F_ACTIVE; }
  __device__ int dim(int i) const { return E(i).dim; }
  __device__ bool col_free(int i, int c) const { return c >= 6 || g.pose_col[E(i).pose] >= 0; }
  __device__ void residual(int i, int c, double step, const State& st, double* e) const {
    const CubEdge& ed = E(i);
    Pose T = load_pose(&st.pose_qt[7 * ed.pose]);
    Cub cb = load_cub(&st.cub[10 * ed.cuboid]);
    double u[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) u[k] = (k == (c < 6 ? c : c - 6)) ? step : 0.0;
    if (c >= 0 && c < 6) T = pose_boxplus(T, u);
    else if (c >= 6) {
      const double* fs = &g.cub_fixedscale[3 * ed.cuboid];
      cb = cuboid_boxplus(cb, u, g.cub_flags[ed.cuboid], fs[0], fs[1], fs[2]);
    }
    if (corner) {
      if (ed.dim == 9) cub3d_residual(T, cb, ed.meas, e);
      else corner_residual(T, cb, ed.K, ed.meas, e);
    } else bbox_residual(T, cb, ed.K, ed.meas, e);
  }
  __device__ double info(int i, int r) const { return E(i).info[r]; }
  __device__ double huber_delta(int i) const { return (E(i).flags & F_ROBUST) ? E(i).huber : 0.0; }
  __device__ void store_chi2(int i, double chi2, double nrm) const {
    if (corner) { g.corner_chi2[i] = chi2; g.corner_norm[i] = nrm; } else { g.bbox_chi2[i] = chi2; g.bbox_norm[i] = nrm; }
  }
  __device__ void scatter(int i, int c, const double* Hrow, double bc) const {
    const CubEdge& ed = E(i);
    if (c < 6) {
      const int fid = g.pose_free_id[ed.pose];
      if (fid < 0) return;
      double* hb = &g.HB[(size_t)fid * 27];
      for (int c2 = 0; c2 <= c; ++c2) acc_add(g, &hb[packed_lower(c, c2)], Hrow[c2]);
      acc_add(g, &hb[21 + c], bc);
      double* w = (corner ? g.Wpc_corner : g.Wpc_bbox) + (size_t)i * 54 + c * 9;
      for (int c2 = 0; c2 < 9; ++c2) w[c2] = Hrow[6 + c2];
    } else {
      const int a = c - 6;
      double* hc = &g.HC[(size_t)ed.cuboid * 54];
      for (int c2 = 0; c2 <= a; ++c2) acc_add(g, &hc[packed_lower(a, c2)], Hrow[6 + c2]);
      acc_add(g, &hc[45 + a], bc);
    }
  }
};

struct PtCubModel {  // unary: cols 0-8 cuboid
  static constexpr int DIM_MAX = 3, NCOL = 9;
  const Graph& g;
  bool corner;  // unused
  __device__ bool active(int i) const { return g.ptcub[i].flags & F_ACTIVE; }
  __device__ int dim(int) const { return 3; }
  __device__ bool col_free(int, int) const { return true; }
  __device__ void residual(int i, int c, double step, const State& st, double* e) const {
    const PtCubEdge& ed = g.ptcub[i];
    Cub cb = load_cub(&st.cub[10 * ed.cuboid]);
    if (c >= 0) {
      double u[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) u[k] = (k == c) ? step : 0.0;
      const double* fs = &g.cub_fixedscale[3 * ed.cuboid];
      cb = cuboid_boxplus(cb, u, g.cub_flags[ed.cuboid], fs[0], fs[1], fs[2]);
    }
    ptcub_residual(cb, &g.ptcub_pts[3 * (size_t)ed.p0], ed.p1 - ed.p0, ed.margin, ed.prior, e);
  }
  __device__ double info(int i, int r) const { return g.ptcub[i].info[r]; }
  __device__ double huber_delta(int i) const { return (g.ptcub[i].flags & F_ROBUST) ? g.ptcub[i].huber : 0.0; }
  __device__ void store_chi2(int i, double chi2, double) const { g.ptcub_chi2[i] = chi2; }
  __device__ void scatter(int i, int c, const double* Hrow, double bc) const {
    double* hc = &g.HC[(size_t)g.ptcub[i].cuboid * 54];
    for (int c2 = 0; c2 <= c; ++c2) acc_add(g, &hc[packed_lower(c, c2)], Hrow[c2]);
    acc_add(g, &hc[45 + c], bc);
  }
};

// 2 NCOL + 1 lanes per edge, packed densely into each wave (3 edges per wave for the 9-column models -- plane, point-cuboid --
// and 2 for the 15-column cuboid projections): lanes 2c and 2c + 1 evaluate e(x [+] delta e_c) and e(x [-] delta e_c), lane
// 2 NCOL the base residual (base_binary_edge.hpp:217-320, base_unary_edge.hpp:82-123, then :55-120 / :43-72).  A residual is a
// chain of several thousand dependent FP64 instructions and a lone wave issues one every ~8 cycles: with both signs in one
// lane every numeric class took ~21 us however few edges it had (50 point-cuboid edges: 22 us), which set the floor of the
// whole fused launch.
template <class Model>
struct NumericShape {
  static constexpr int LPE = 2 * Model::NCOL + 1;   // lanes per edge
  static constexpr int EPW = 64 / LPE;              // edges per wave
  static constexpr int EPB = 4 * EPW;               // edges per 256-thread workgroup
};
// (a, c) of the packed lower-triangle index k = a (a + 1) / 2 + c
__device__ __forceinline__ void unpack_lower(int k, int& a, int& c) {  // (closed form: no data-dependent loop, the lanes of a wave stay together)
  a = (int)((__fsqrt_rn(8.0f * (float)k + 1.0f) - 1.0f) * 0.5f);  // exact for the k < 64 used here: 8k + 1 is a perfect square exactly at the row starts
  c = k - a * (a + 1) / 2;
}
// element (row, col) of J' (rho1 Omega) J and element row of -J' (rho1 Omega e) from the Jacobian staged in LDS ([r][column], 16 per row)
template <int DM>
__device__ __forceinline__ double quad_H(const double (*sJ)[16], const double* sW, int dim, int row, int col) {
  double h = 0;
#pragma unroll
  for (int r = 0; r < DM; ++r)
    if (r < dim) h += sJ[r][row] * sW[r] * sJ[r][col];
  return h;
}
template <int DM>
__device__ __forceinline__ double quad_b(const double (*sJ)[16], const double* sG, int dim, int row) {
  double b = 0;
#pragma unroll
  for (int r = 0; r < DM; ++r)
    if (r < dim) b += sJ[r][row] * sG[r];
  return b;
}

// The accumulation of a numerically differentiated edge is LANE-CONTIGUOUS: the entries one edge adds to a block (D_l | b_l, the packed
// pose block, W, the packed cuboid block) are consecutive doubles, and consecutive lanes add consecutive entries, so one atomic
// instruction touches 1-4 cache lines.  With one lane per Jacobian column adding "its" row (up to 15 single-lane atomics in a loop) the
// 4 000 plane edges alone issued 216 000 line operations, 135-180 of them per hot line (a plane's D_l, a keyframe's pose block): the
// accumulation phase took 6 us of a 14 us workgroup (tools/lin_stamps.py), against 2 us now.

// entry o (= lane + 19 * round) of the 54 a plane edge adds: row | col << 4 | is_b << 8 | segment (0 D_l, 1 b_l, 2 pose block, 3 W) << 9 | offset << 11
__device__ const uint16_t kPlaneEmit[57] = {0x0000, 0x0810, 0x1020, 0x1811, 0x2021, 0x2822, 0x0300, 0x0b01, 0x1302, 0x0433, 0x0c34, 0x1444, 0x1c35, 0x2445, 0x2c55, 0x3436, 0x3c46, 0x4456, 0x4c66, 0x5437, 0x5c47, 0x6457, 0x6c67, 0x7477, 0x7c38, 0x8448, 0x8c58, 0x9468, 0x9c78, 0xa488, 0xad03, 0xb504, 0xbd05, 0xc506, 0xcd07, 0xd508, 0x0603, 0x0e13, 0x1623, 0x1e04, 0x2614, 0x2e24, 0x3605, 0x3e15, 0x4625, 0x4e06, 0x5616, 0x5e26, 0x6607, 0x6e17, 0x7627, 0x7e08, 0x8618, 0x8e28, 0xffff, 0xffff, 0xffff};

// ---- plane edges: 19 lanes per edge (the two signs of a column in neighbouring lanes + the base residual), 3 edges per wave ----------
__device__ __forceinline__ void lin_plane_body(const Graph& g, const State& st, int n_edges, double delta, int vblock, double* smem) {
  using Model = PlaneModel;
  constexpr int DM = Model::DIM_MAX, NC = Model::NCOL;
  constexpr int LPE = NumericShape<Model>::LPE, EPW = NumericShape<Model>::EPW, EPB = NumericShape<Model>::EPB;
  double (*sJ)[DM][16] = reinterpret_cast<double (*)[DM][16]>(smem);  // [group][row][col]
  double (*sW)[DM] = reinterpret_cast<double (*)[DM]>(smem + EPB * DM * 16);
  double (*sG)[DM] = reinterpret_cast<double (*)[DM]>(smem + EPB * DM * 16 + EPB * DM);
  double* s_red = smem + EPB * DM * 16 + 2 * EPB * DM;
  const Model m{g, false};
  const int wl = threadIdx.x & 63, ew = wl / LPE;
  const int lane = wl - ew * LPE;
  const int grp = (threadIdx.x >> 6) * EPW + ew;
  const int i = vblock * EPB + grp;
  LIN_STAMP(g, 0);
  const bool act = ew < EPW && i < n_edges && m.active(i);
  const int dim = act ? m.dim(i) : 0;
  double chi_local = 0.0;
  if (act) {
    // ONE call site of the residual (thousands of FP64 instructions: the instruction cache matters); the two lanes of a column
    // are neighbours, so the central difference is one lane shuffle per residual row
    const int c = lane >> 1;
    const bool col = lane < 2 * NC, base = lane == 2 * NC;
    const bool fr = col && m.col_free(i, c);
    double eo[DM];
#pragma unroll
    for (int r = 0; r < DM; ++r) eo[r] = 0.0;
    if (fr || base) m.residual(i, base ? -1 : c, base ? 0.0 : ((lane & 1) ? -delta : delta), st, eo);
    LIN_STAMP(g, 1);
    if (col) {
      const double scalar = 1.0 / (2 * delta);
#pragma unroll
      for (int r = 0; r < DM; ++r) {
        const double em = __shfl_down(eo[r], 1, 64);  // (lane 2c + 1 holds e(x [-] delta e_c); both lanes of a column are active together)
        if (r < dim && !(lane & 1)) sJ[grp][r][c] = fr ? scalar * (eo[r] - em) : 0.0;
      }
    } else if (base) {
      double chi2 = 0, nrm = 0;
#pragma unroll
      for (int r = 0; r < DM; ++r)
        if (r < dim) { chi2 += eo[r] * (m.info(i, r) * eo[r]); nrm += eo[r] * eo[r]; }
      m.store_chi2(i, chi2, sqrt(nrm));
      double rho0, rho1;
      huber(chi2, m.huber_delta(i), rho0, rho1);
      chi_local = rho0;
#pragma unroll
      for (int r = 0; r < DM; ++r)
        if (r < dim) { sW[grp][r] = rho1 * m.info(i, r); sG[grp][r] = -(m.info(i, r) * eo[r]) * rho1; }
    }
  }
  // (the three edges of a wave stage their Jacobians for lanes of the SAME wave: a wave-local fence, no workgroup barrier -- the
  //  waves of a plane workgroup finish their residual chains up to 4 us apart, tools/lin_stamps.py)
  lds_wave_fence();
  LIN_STAMP(g, 2);
#ifdef PPO_BA_TEST_HOOKS
  if (!(g.dbg & 1))
#endif
  if (act) {
    // 54 entries per edge: D_l (6, packed upper) | b_l (3) | pose block (21 packed lower + 6) | W (6 x 3)
    // (written without branches on o: the 19 lanes of an edge sit in up to four of the segments at once)
    const PlaneEdge& ed = g.pl[i];
    const int l = g.n_points + ed.plane, fid = g.pose_free_id[ed.pose];
    double* const pD = &g.D[(size_t)l * 6];
    double* const pb = &g.bl[(size_t)l * 3];
    double* const pH = &g.HB[(size_t)(fid >= 0 ? fid : 0) * 27];
    double* const pW = &g.W[(size_t)(ed.slot >= 0 ? ed.slot - g.n_pt_slots : 0) * 18];
#pragma unroll
    for (int rd = 0; rd < 3; ++rd) {
      const int o = lane + rd * LPE;
      const unsigned en = kPlaneEmit[o];  // (a table: decoding o arithmetically was a third of this phase)
      const int row = en & 15, col = (en >> 4) & 15, seg = (en >> 9) & 3, off = en >> 11;
      const bool isb = (en >> 8) & 1;
      double* const dst = (seg == 0 ? pD : seg == 1 ? pb : seg == 2 ? pH : pW) + off;
      const bool valid = o < 54 && (seg < 2 || (fid >= 0 && (seg == 2 || ed.slot >= 0)));
      double h = 0;
#pragma unroll
      for (int r = 0; r < DM; ++r) {
        const double x = sJ[grp][r][row], y = isb ? sG[grp][r] : sW[grp][r], z = isb ? 1.0 : sJ[grp][r][col];
        if (r < dim) h += x * y * z;
      }
      if (valid) acc_add(g, dst, h);
    }
  }
  LIN_STAMP(g, 3);
  (void)s_red;
  wave_atomic_add(g, chi_local, ACC_CHI2);
  LIN_STAMP(g, 4);
}

// ---- cuboid projections (bbox / corner edges): ONE edge per workgroup, 8 lanes -- one per cuboid corner -- per residual ------------------
// 31 residuals (the two signs of 15 columns + the base) x 8 corners = 248 lanes.  A residual is a latency chain (box-plus, two rotation
// matrices, then the corners one after the other: ~1 400 dependent FP64 instructions, 7 us for a lone lane); with the corners across lanes
// the chain is the box-plus and ONE corner, and the hull (bbox) is three lane exchanges.
template <bool CORNER>
struct NumericShape<CubProjModelT<CORNER>> {
  static constexpr int EPB = 1;
};
template <bool CORNER>
__device__ __forceinline__ void lin_cubproj_body(const Graph& g, const State& st, int i, double delta, double* smem) {
  constexpr int DM = CORNER ? 16 : 4, NC = 15;
  double (*sJ)[16] = reinterpret_cast<double (*)[16]>(smem);  // [row][col]
  double* sW = smem + DM * 16;
  double* sG = sW + DM;
  double* s_red = sG + DM;
  const CubEdge& ed = CORNER ? g.corner[i] : g.bbox[i];
  const int tid = threadIdx.x, r = tid >> 3, k = tid & 7;
  const int c = r >> 1;
  const bool act = ed.flags & F_ACTIVE;  // (uniform)
  const bool col = r < 2 * NC, base = r == 2 * NC;
  const int fid = g.pose_free_id[ed.pose];
  const bool fr = col && (c >= 6 || g.pose_col[ed.pose] >= 0);
  LIN_STAMP(g, 0);
  double chi_local = 0.0;
  if (act) {
    double u = 0.0, v = 0.0;
    if (fr || base) {
      Pose T = load_pose(&st.pose_qt[7 * ed.pose]);
      Cub cb = load_cub(&st.cub[10 * ed.cuboid]);
      const double step = (r & 1) ? -delta : delta;
      double d[9];
#pragma unroll
      for (int q = 0; q < 9; ++q) d[q] = (!base && q == (c < 6 ? c : c - 6)) ? step : 0.0;
      if (!base && c < 6) T = pose_boxplus(T, d);
      else if (!base) {
        const double* fs = &g.cub_fixedscale[3 * ed.cuboid];
        cb = cuboid_boxplus(cb, d, g.cub_flags[ed.cuboid], fs[0], fs[1], fs[2]);
      }
      const Mat3 Rwo = rotmat(cb.T.q), Rcw = rotmat(T.q);
      cuboid_corner_px(cb, Rwo, Rcw, T.t, ed.K, k, u, v);
    }
    LIN_STAMP(g, 1);
    // residual rows owned by this lane: bbox row k (k < 4) of the hull, corner rows 2k and 2k + 1
    double e0, e1 = 0.0;
    if (CORNER) {
      e0 = u - ed.meas[2 * k];
      e1 = v - ed.meas[2 * k + 1];
    } else {
      double ulo = u, uhi = u, vlo = v, vhi = v;
#pragma unroll
      for (int sft = 1; sft < 8; sft <<= 1) {  // (min / max are exact: the order of the corners does not matter)
        ulo = fmin(ulo, __shfl_xor(ulo, sft, 64)); uhi = fmax(uhi, __shfl_xor(uhi, sft, 64));
        vlo = fmin(vlo, __shfl_xor(vlo, sft, 64)); vhi = fmax(vhi, __shfl_xor(vhi, sft, 64));
      }
      e0 = k == 0 ? (uhi + ulo) / 2 - ed.meas[0] : (k == 1 ? (vhi + vlo) / 2 - ed.meas[1] : (k == 2 ? (uhi - ulo) - ed.meas[2] : (vhi - vlo) - ed.meas[3]));
    }
    const double scalar = 1.0 / (2 * delta);
    const double m0 = __shfl_down(e0, 8, 64), m1 = __shfl_down(e1, 8, 64);  // the minus-sign residual sits 8 lanes up
    if (col && !(r & 1)) {
      if (CORNER) { sJ[2 * k][c] = fr ? scalar * (e0 - m0) : 0.0; sJ[2 * k + 1][c] = fr ? scalar * (e1 - m1) : 0.0; }
      else if (k < 4) sJ[k][c] = fr ? scalar * (e0 - m0) : 0.0;
    }
    if (base) {  // rows gathered in order: chi2 is summed exactly like the thread-per-edge residual pass does
      double chi2 = 0, nrm = 0;
      const int lane0 = (threadIdx.x & 63) & ~7;
#pragma unroll
      for (int row = 0; row < DM; ++row) {
        const double er = CORNER ? __shfl((row & 1) ? e1 : e0, lane0 + (row >> 1), 64) : __shfl(e0, lane0 + row, 64);
        chi2 += er * (ed.info[row] * er);
        nrm += er * er;
      }
      double rho0, rho1;
      huber(chi2, (ed.flags & F_ROBUST) ? ed.huber : 0.0, rho0, rho1);
      if (k == 0) {
        if (CORNER) { g.corner_chi2[i] = chi2; g.corner_norm[i] = sqrt(nrm); } else { g.bbox_chi2[i] = chi2; g.bbox_norm[i] = sqrt(nrm); }
        chi_local = rho0;
      }
      if (CORNER) {
        sW[2 * k] = rho1 * ed.info[2 * k]; sG[2 * k] = -(ed.info[2 * k] * e0) * rho1;
        sW[2 * k + 1] = rho1 * ed.info[2 * k + 1]; sG[2 * k + 1] = -(ed.info[2 * k + 1] * e1) * rho1;
      } else if (k < 4) {
        sW[k] = rho1 * ed.info[k]; sG[k] = -(ed.info[k] * e0) * rho1;
      }
    }
  }
  __syncthreads();
  LIN_STAMP(g, 2);
#ifdef PPO_BA_TEST_HOOKS
  if (!(g.dbg & 1))
#endif
  if (act && tid < 135) {
    // 135 entries, one per lane: pose block (21 packed lower + 6) | cuboid block (45 packed lower + 9) | W pose rows x cuboid cols (6 x 9)
    if (tid < 27) {
      if (fid >= 0) {
        double val;
        if (tid < 21) { int a, c2; unpack_lower(tid, a, c2); val = quad_H<DM>(sJ, sW, DM, a, c2); }
        else val = quad_b<DM>(sJ, sG, DM, tid - 21);
        acc_add(g, &g.HB[(size_t)fid * 27 + tid], val);
      }
    } else if (tid < 81) {
      const int q = tid - 27;
      double val;
      if (q < 45) { int a, c2; unpack_lower(q, a, c2); val = quad_H<DM>(sJ, sW, DM, 6 + a, 6 + c2); }
      else val = quad_b<DM>(sJ, sG, DM, 6 + (q - 45));
      acc_add(g, &g.HC[(size_t)ed.cuboid * 54 + q], val);
    } else if (fid >= 0) {
      const int q = tid - 81;
      ((CORNER ? g.Wpc_corner : g.Wpc_bbox) + (size_t)i * 54)[q] = quad_H<DM>(sJ, sW, DM, q / 9, 6 + q % 9);
    }
  }
  LIN_STAMP(g, 3);
  block_atomic_add(g, chi_local, ACC_CHI2, s_red);
  LIN_STAMP(g, 4);
}

// ---- the 9-D camera-object edge (EdgeSE3Cuboid, Optimizer::LocalBACameraPointCuboids2D): ONE edge per workgroup, one lane per residual -------
// 31 lanes: the two signs of 15 columns + the base residual, each the whole chain (four SE3 logarithms).  This edge class is switched
// off in the reference's own call site (optimize_with_cuboid_3d, src/LocalMapping.cc:97): it is built for completeness of the path, not
// tuned; the records live behind the corner-projection edges (g.corner[n_corner2d ..]) and share their Hessian blocks and W storage.
__device__ __forceinline__ void lin_cub3d_body(const Graph& g, const State& st, int i, double delta, double* smem) {
  constexpr int DM = 16, NC = 15, DIM = 9;
  double (*sJ)[16] = reinterpret_cast<double (*)[16]>(smem);  // [row][col]
  double* sW = smem + DM * 16;
  double* sG = sW + DM;
  double* s_red = sG + DM;
  const CubEdge& ed = g.corner[i];
  const int tid = threadIdx.x, c = tid >> 1;
  const bool act = ed.flags & F_ACTIVE;  // (uniform)
  const bool col = tid < 2 * NC, base = tid == 2 * NC;
  const int fid = g.pose_free_id[ed.pose];
  const bool fr = col && (c >= 6 || g.pose_col[ed.pose] >= 0);
  double chi_local = 0.0;
  if (act && tid < 64) {
    double e[DIM];
#pragma unroll
    for (int q = 0; q < DIM; ++q) e[q] = 0.0;
    if (fr || base) {
      Pose T = load_pose(&st.pose_qt[7 * ed.pose]);
      Cub cb = load_cub(&st.cub[10 * ed.cuboid]);
      const double step = (tid & 1) ? -delta : delta;
      double d[9];
#pragma unroll
      for (int q = 0; q < 9; ++q) d[q] = (!base && q == (c < 6 ? c : c - 6)) ? step : 0.0;
      if (!base && c < 6) T = pose_boxplus(T, d);
      else if (!base) {
        const double* fs = &g.cub_fixedscale[3 * ed.cuboid];
        cb = cuboid_boxplus(cb, d, g.cub_flags[ed.cuboid], fs[0], fs[1], fs[2]);
      }
      cub3d_residual(T, cb, ed.meas, e);
    }
    const double scalar = 1.0 / (2 * delta);
#pragma unroll
    for (int q = 0; q < DIM; ++q) {
      const double em = __shfl_down(e[q], 1, 64);  // lane 2c + 1 holds e(x [-] delta e_c)
      if (col && !(tid & 1)) sJ[q][c] = fr ? scalar * (e[q] - em) : 0.0;
    }
    if (base) {
      double chi2 = 0, nrm = 0;
#pragma unroll
      for (int q = 0; q < DIM; ++q) { chi2 += e[q] * (ed.info[q] * e[q]); nrm += e[q] * e[q]; }
      g.corner_chi2[i] = chi2;
      g.corner_norm[i] = sqrt(nrm);
      double rho0, rho1;
      huber(chi2, (ed.flags & F_ROBUST) ? ed.huber : 0.0, rho0, rho1);
      chi_local = rho0;
#pragma unroll
      for (int q = 0; q < DIM; ++q) { sW[q] = rho1 * ed.info[q]; sG[q] = -(ed.info[q] * e[q]) * rho1; }
    }
  }
  __syncthreads();
#ifdef PPO_BA_TEST_HOOKS
  if (!(g.dbg & 1))
#endif
  if (act && tid < 135) {
    // 135 entries, one per lane: pose block (21 packed lower + 6) | cuboid block (45 packed lower + 9) | W pose rows x cuboid cols (6 x 9)
    if (tid < 27) {
      if (fid >= 0) {
        double val;
        if (tid < 21) { int a, c2; unpack_lower(tid, a, c2); val = quad_H<DM>(sJ, sW, DIM, a, c2); }
        else val = quad_b<DM>(sJ, sG, DIM, tid - 21);
        acc_add(g, &g.HB[(size_t)fid * 27 + tid], val);
      }
    } else if (tid < 81) {
      const int q = tid - 27;
      double val;
      if (q < 45) { int a, c2; unpack_lower(q, a, c2); val = quad_H<DM>(sJ, sW, DIM, 6 + a, 6 + c2); }
      else val = quad_b<DM>(sJ, sG, DIM, 6 + (q - 45));
      acc_add(g, &g.HC[(size_t)ed.cuboid * 54 + q], val);
    } else if (fid >= 0) {
      const int q = tid - 81;
      (g.Wpc_corner + (size_t)i * 54)[q] = quad_H<DM>(sJ, sW, DIM, q / 9, 6 + q % 9);
    }
  }
  block_atomic_add(g, chi_local, ACC_CHI2, s_red);
}

// ---- analytic mode (options.jacobian_mode = PPO_JAC_ANALYTIC) ------------------------------------------------------------------------
// The numeric classes above spend 19 (planes) or 248 (cuboid projections) lanes per edge on the 2 x columns + 1 residual evaluations of a
// central difference.  With the exact derivative an edge needs ONE evaluation: 8 lanes per edge -- for a cuboid projection one lane
// per corner -- , 32 edges per workgroup: 4 000 plane edges are 125 workgroups instead of 334, 500 bbox edges 16 instead of 500, and
// the fused launch is back to one round of workgroups.  The accumulation (entries, order inside an edge, lane-contiguous atomics) is
// the numeric classes': sJ / sW / sG in LDS, then consecutive lanes add consecutive entries.
constexpr int kAnaLanes = 8, kAnaPlaneEPB = 32, kAnaBboxEPB = 32, kAnaCornerEPB = 12;
__device__ __forceinline__ void lin_plane_body_analytic(const Graph& g, const State& st, int n_edges, int vblock, double* smem) {
  constexpr int DM = 3, EPB = kAnaPlaneEPB;
  double (*sJ)[DM][16] = reinterpret_cast<double (*)[DM][16]>(smem);  // [edge of the workgroup][row][col]
  double (*sW)[DM] = reinterpret_cast<double (*)[DM]>(smem + EPB * DM * 16);
  double (*sG)[DM] = reinterpret_cast<double (*)[DM]>(smem + EPB * DM * 16 + EPB * DM);
  const PlaneModel m{g, false};
  const int grp = threadIdx.x >> 3, lane = threadIdx.x & 7;
  const int i = vblock * EPB + grp;
  LIN_STAMP(g, 0);
  const bool act = i < n_edges && m.active(i);
  const int dim = act ? m.dim(i) : 0;
  double chi_local = 0.0;
  if (act) {
    const PlaneEdge& ed = g.pl[i];
    double e[3] = {0.0, 0.0, 0.0}, J[27];
    plane_edge_analytic(load_pose(&st.pose_qt[7 * ed.pose]), load_plane(&st.plane[4 * ed.plane]), PlaneC{ed.meas[0], ed.meas[1], ed.meas[2], ed.meas[3]},
                        (ed.flags >> 8) & 3, e, J);
    const bool pose_free = g.pose_col[ed.pose] >= 0;  // nothing is differentiated for a fixed vertex (base_binary_edge.hpp:275-278)
#pragma unroll
    for (int r = 0; r < DM; ++r)
      if (r < dim) {
        // (every lane holds the whole Jacobian; lane q stages column q, lane 0 also column 8 -- selected without a run-time register index)
        double v = 0.0;
#pragma unroll
        for (int c = 0; c < 8; ++c) v = (c == lane) ? J[r * 9 + c] : v;
        sJ[grp][r][lane] = (lane < 3 || pose_free) ? v : 0.0;
        if (lane == 0) sJ[grp][r][8] = pose_free ? J[r * 9 + 8] : 0.0;
      }
    if (lane == 0) {
      double chi2 = 0, nrm = 0;
#pragma unroll
      for (int r = 0; r < DM; ++r)
        if (r < dim) { chi2 += e[r] * (ed.info[r] * e[r]); nrm += e[r] * e[r]; }
      m.store_chi2(i, chi2, sqrt(nrm));
      double rho0, rho1;
      huber(chi2, m.huber_delta(i), rho0, rho1);
      chi_local = rho0;
#pragma unroll
      for (int r = 0; r < DM; ++r)
        if (r < dim) { sW[grp][r] = rho1 * ed.info[r]; sG[grp][r] = -(ed.info[r] * e[r]) * rho1; }
    }
  }
  LIN_STAMP(g, 1);
  lds_wave_fence();  // (the eight lanes of an edge sit in one wave)
  LIN_STAMP(g, 2);
  if (act) {
    const PlaneEdge& ed = g.pl[i];
    const int l = g.n_points + ed.plane, fid = g.pose_free_id[ed.pose];
    double* const pD = &g.D[(size_t)l * 6];
    double* const pb = &g.bl[(size_t)l * 3];
    double* const pH = &g.HB[(size_t)(fid >= 0 ? fid : 0) * 27];
    double* const pW = &g.W[(size_t)(ed.slot >= 0 ? ed.slot - g.n_pt_slots : 0) * 18];
#pragma unroll
    for (int rd = 0; rd < 7; ++rd) {  // 54 entries per edge (kPlaneEmit), eight per round
      const int o = lane + rd * kAnaLanes;
      const unsigned en = kPlaneEmit[o < 54 ? o : 54];
      const int row = en & 15, col = (en >> 4) & 15, seg = (en >> 9) & 3, off = en >> 11;
      const bool isb = (en >> 8) & 1;
      double* const dst = (seg == 0 ? pD : seg == 1 ? pb : seg == 2 ? pH : pW) + off;
      const bool valid = o < 54 && (seg < 2 || (fid >= 0 && (seg == 2 || ed.slot >= 0)));
      double h = 0;
#pragma unroll
      for (int r = 0; r < DM; ++r) {
        const double x = sJ[grp][r][row], y = isb ? sG[grp][r] : sW[grp][r], z = isb ? 1.0 : sJ[grp][r][col];
        if (r < dim) h += x * y * z;
      }
      if (valid) acc_add(g, dst, h);
    }
  }
  LIN_STAMP(g, 3);
  wave_atomic_add(g, chi_local, ACC_CHI2);
  LIN_STAMP(g, 4);
}

template <bool CORNER>
__device__ __forceinline__ void lin_cubproj_body_analytic(const Graph& g, const State& st, int n_edges, int vblock, double* smem) {
  constexpr int DM = CORNER ? 16 : 4, EPB = CORNER ? kAnaCornerEPB : kAnaBboxEPB;
  double (*sJ)[DM][16] = reinterpret_cast<double (*)[DM][16]>(smem);  // [edge of the workgroup][row][col]
  double (*sW)[DM] = reinterpret_cast<double (*)[DM]>(smem + EPB * DM * 16);
  double (*sG)[DM] = reinterpret_cast<double (*)[DM]>(smem + EPB * DM * 16 + EPB * DM);
  const int grp = threadIdx.x >> 3, k = threadIdx.x & 7;  // lane k = cuboid corner k
  const int i = vblock * EPB + grp;
  // These few workgroups (16 for 500 bbox edges) share their SIMDs with three other waves each of the point / pose parts: time-sliced evenly
  // their chain took 23 us against 6 us alone and ended the launch (tools/lin_stamps.py, JAC=1).  Raised issue priority gives it back.
  __builtin_amdgcn_s_setprio(3);
  const bool in = grp < EPB && i < n_edges;
  const CubEdge& ed = (CORNER ? g.corner : g.bbox)[in ? i : 0];
  const bool act = in && (ed.flags & F_ACTIVE);
  const int fid = g.pose_free_id[ed.pose];
  LIN_STAMP(g, 0);
  double chi_local = 0.0;
  if (act) {
    const Pose T = load_pose(&st.pose_qt[7 * ed.pose]);
    const Cub cb = load_cub(&st.cub[10 * ed.cuboid]);
    const Mat3 Rwo = rotmat(cb.T.q), Rcw = rotmat(T.q);
    double u, v, Ju[15], Jv[15];
    cuboid_corner_analytic(cb, Rwo, Rcw, T.t, ed.K, g.cub_flags[ed.cuboid], g.cub_fixedscale[3 * ed.cuboid] > 0, k, u, v, Ju, Jv);
    if (g.pose_col[ed.pose] < 0) {  // fixed pose: its columns are not differentiated
#pragma unroll
      for (int c = 0; c < 6; ++c) { Ju[c] = 0.0; Jv[c] = 0.0; }
    }
    const int lane0 = (threadIdx.x & 63) & ~7;
    double e0, e1 = 0.0;
    if (CORNER) {
      e0 = u - ed.meas[2 * k];
      e1 = v - ed.meas[2 * k + 1];
#pragma unroll
      for (int c = 0; c < 15; ++c) { sJ[grp][2 * k][c] = Ju[c]; sJ[grp][2 * k + 1][c] = Jv[c]; }
    } else {
      // the hull and the corners attaining it (the lowest corner index on a tie, like the sequential scan of bbox_edge_analytic)
      double ulo = u, uhi = u, vlo = v, vhi = v;
#pragma unroll
      for (int sft = 1; sft < 8; sft <<= 1) {
        ulo = fmin(ulo, __shfl_xor(ulo, sft, 64)); uhi = fmax(uhi, __shfl_xor(uhi, sft, 64));
        vlo = fmin(vlo, __shfl_xor(vlo, sft, 64)); vhi = fmax(vhi, __shfl_xor(vhi, sft, 64));
      }
      int kulo = (u == ulo) ? k : 8, kuhi = (u == uhi) ? k : 8, kvlo = (v == vlo) ? k : 8, kvhi = (v == vhi) ? k : 8;
#pragma unroll
      for (int sft = 1; sft < 8; sft <<= 1) {
        kulo = min(kulo, __shfl_xor(kulo, sft, 64)); kuhi = min(kuhi, __shfl_xor(kuhi, sft, 64));
        kvlo = min(kvlo, __shfl_xor(kvlo, sft, 64)); kvhi = min(kvhi, __shfl_xor(kvhi, sft, 64));
      }
      e0 = k == 0 ? (uhi + ulo) / 2 - ed.meas[0] : (k == 1 ? (vhi + vlo) / 2 - ed.meas[1] : (k == 2 ? (uhi - ulo) - ed.meas[2] : (vhi - vlo) - ed.meas[3]));
#pragma unroll
      for (int c = 0; c < 15; ++c) {
        const double a = __shfl(Ju[c], lane0 + kuhi, 64), b = __shfl(Ju[c], lane0 + kulo, 64);
        const double p = __shfl(Jv[c], lane0 + kvhi, 64), q = __shfl(Jv[c], lane0 + kvlo, 64);
        if ((c & 7) == k) { sJ[grp][0][c] = (a + b) / 2; sJ[grp][1][c] = (p + q) / 2; sJ[grp][2][c] = a - b; sJ[grp][3][c] = p - q; }
      }
    }
    // rows gathered in order: chi2 is summed exactly like the residual-only pass does
    double chi2 = 0, nrm = 0;
#pragma unroll
    for (int row = 0; row < DM; ++row) {
      const double er = CORNER ? __shfl((row & 1) ? e1 : e0, lane0 + (row >> 1), 64) : __shfl(e0, lane0 + row, 64);
      chi2 += er * (ed.info[row] * er);
      nrm += er * er;
    }
    double rho0, rho1;
    huber(chi2, (ed.flags & F_ROBUST) ? ed.huber : 0.0, rho0, rho1);
    if (k == 0) {
      if (CORNER) { g.corner_chi2[i] = chi2; g.corner_norm[i] = sqrt(nrm); } else { g.bbox_chi2[i] = chi2; g.bbox_norm[i] = sqrt(nrm); }
      chi_local = rho0;
    }
    if (CORNER) {
      sW[grp][2 * k] = rho1 * ed.info[2 * k]; sG[grp][2 * k] = -(ed.info[2 * k] * e0) * rho1;
      sW[grp][2 * k + 1] = rho1 * ed.info[2 * k + 1]; sG[grp][2 * k + 1] = -(ed.info[2 * k + 1] * e1) * rho1;
    } else if (k < 4) {
      sW[grp][k] = rho1 * ed.info[k]; sG[grp][k] = -(ed.info[k] * e0) * rho1;
    }
  }
  LIN_STAMP(g, 1);
  lds_wave_fence();  // (the eight lanes of an edge sit in one wave)
  LIN_STAMP(g, 2);
  if (act) {
    // 135 entries: pose block (21 packed lower + 6) | cuboid block (45 packed lower + 9) | W pose rows x cuboid cols (6 x 9), eight per round
    const double (*J)[16] = sJ[grp];
#pragma unroll 1
    for (int o = k; o < 135; o += kAnaLanes) {
      if (o < 27) {
        if (fid >= 0) {
          double val;
          if (o < 21) { int a, c2; unpack_lower(o, a, c2); val = quad_H<DM>(J, sW[grp], DM, a, c2); }
          else val = quad_b<DM>(J, sG[grp], DM, o - 21);
          acc_add(g, &g.HB[(size_t)fid * 27 + o], val);
        }
      } else if (o < 81) {
        const int q = o - 27;
        double val;
        if (q < 45) { int a, c2; unpack_lower(q, a, c2); val = quad_H<DM>(J, sW[grp], DM, 6 + a, 6 + c2); }
        else val = quad_b<DM>(J, sG[grp], DM, 6 + (q - 45));
        acc_add(g, &g.HC[(size_t)ed.cuboid * 54 + q], val);
      } else if (fid >= 0) {
        const int q = o - 81;
        ((CORNER ? g.Wpc_corner : g.Wpc_bbox) + (size_t)i * 54)[q] = quad_H<DM>(J, sW[grp], DM, q / 9, 6 + q % 9);
      }
    }
  }
  LIN_STAMP(g, 3);
  wave_atomic_add(g, chi_local, ACC_CHI2);
  LIN_STAMP(g, 4);
}
constexpr int kAnalyticLdsDoubles = kAnaCornerEPB * 16 * 18 > kAnaPlaneEPB * 3 * 18 ? (kAnaCornerEPB * 16 * 18 > kAnaBboxEPB * 4 * 18 ? kAnaCornerEPB * 16 * 18 : kAnaBboxEPB * 4 * 18)
                                                                                   : kAnaPlaneEPB * 3 * 18;

// ---- point-cuboid edges: ONE edge per workgroup, 8 lanes per residual, each summing every eighth point -----------------------------------
// (19 residuals x 8 = 152 lanes.  The residual is a mean over the cuboid's points, ~13 us as one chain for the 30-40 points of a
//  BASELINE cuboid; the eight partial sums are combined by a fixed exchange tree, so the result does not depend on anything but the edge.)
template <>
struct NumericShape<PtCubModel> {
  static constexpr int EPB = 1;
};
__device__ __forceinline__ void lin_ptcub_body(const Graph& g, const State& st, int i, double delta, double* smem) {
  constexpr int DM = 3, NC = 9;
  __builtin_amdgcn_s_setprio(3);  // (50 workgroups with the longest chain of the launch: see lin_cubproj_body_analytic)
  double (*sJ)[16] = reinterpret_cast<double (*)[16]>(smem);
  double* sW = smem + DM * 16;
  double* sG = sW + DM;
  double* s_red = sG + DM;
  const PtCubEdge& ed = g.ptcub[i];
  const int tid = threadIdx.x, r = tid >> 3, k = tid & 7;
  const int c = r >> 1;
  const bool act = ed.flags & F_ACTIVE;
  const bool col = r < 2 * NC, base = r == 2 * NC;
  LIN_STAMP(g, 0);
  double chi_local = 0.0;
  if (act && (col || base)) {
    Cub cb = load_cub(&st.cub[10 * ed.cuboid]);
    if (col) {
      double d[9];
#pragma unroll
      for (int q = 0; q < 9; ++q) d[q] = (q == c) ? ((r & 1) ? -delta : delta) : 0.0;
      const double* fs = &g.cub_fixedscale[3 * ed.cuboid];
      cb = cuboid_boxplus(cb, d, g.cub_flags[ed.cuboid], fs[0], fs[1], fs[2]);
    }
    const int npts = ed.p1 - ed.p0;
    double acc[3];
    ptcub_partial(cb, &g.ptcub_pts[3 * (size_t)ed.p0], npts, k, 8, ed.margin, acc);
#pragma unroll
    for (int sft = 1; sft < 8; sft <<= 1)
#pragma unroll
      for (int q = 0; q < 3; ++q) acc[q] += __shfl_xor(acc[q], sft, 64);
    double e[3];
    ptcub_finish(cb, acc, npts, ed.prior, e);
    LIN_STAMP(g, 1);
    const double scalar = 1.0 / (2 * delta);
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const double em = __shfl_down(e[q], 8, 64);
      if (col && !(r & 1) && k == q) sJ[q][c] = scalar * (e[q] - em);
    }
    if (base && k == 0) {
      double chi2 = 0;
#pragma unroll
      for (int q = 0; q < 3; ++q) chi2 += e[q] * (ed.info[q] * e[q]);
      g.ptcub_chi2[i] = chi2;
      double rho0, rho1;
      huber(chi2, (ed.flags & F_ROBUST) ? ed.huber : 0.0, rho0, rho1);
      chi_local = rho0;
#pragma unroll
      for (int q = 0; q < 3; ++q) { sW[q] = rho1 * ed.info[q]; sG[q] = -(ed.info[q] * e[q]) * rho1; }
    }
  }
  __syncthreads();
  LIN_STAMP(g, 2);
#ifdef PPO_BA_TEST_HOOKS
  if (!(g.dbg & 1))
#endif
  if (act && tid < 54) {  // the cuboid block: 45 packed lower + 9
    double val;
    if (tid < 45) { int a, c2; unpack_lower(tid, a, c2); val = quad_H<DM>(sJ, sW, DM, a, c2); }
    else val = quad_b<DM>(sJ, sG, DM, tid - 45);
    acc_add(g, &g.HC[(size_t)ed.cuboid * 54 + tid], val);
  }
  LIN_STAMP(g, 3);
  block_atomic_add(g, chi_local, ACC_CHI2, s_red);
  LIN_STAMP(g, 4);
}

template <class Model>
__device__ __forceinline__ void chi2_numeric_body(const Graph& g, const State& st, int n_edges, bool corner, int vblock) {
  constexpr int DM = Model::DIM_MAX;
  __shared__ double s_red[4];
  const Model m{g, corner};
  const int i = vblock * 64 + threadIdx.x;  // one active wave per workgroup: these residuals are long, spread them over the CUs
  double chi_local = 0.0;
  if (threadIdx.x < 64 && i < n_edges && m.active(i)) {
    double e0[DM];
    const int dim = m.dim(i);
    m.residual(i, -1, 0.0, st, e0);
    double chi2 = 0, nrm = 0;
    for (int r = 0; r < dim; ++r) { chi2 += e0[r] * (m.info(i, r) * e0[r]); nrm += e0[r] * e0[r]; }
    m.store_chi2(i, chi2, sqrt(nrm));
    double rho0, rho1;
    huber(chi2, m.huber_delta(i), rho0, rho1);
    chi_local = rho0;
  }
  block_atomic_add(g, chi_local, ACC_CHI2, s_red);
}

// All numerically differentiated edge classes in ONE launch (they are tiny and latency-bound: 4000 plane, 500 bbox,
// 50 point-cuboid edges at the BASELINE size), block ranges select the model.  Runs on a side stream next to the
// reprojection kernels.
struct NumericPlan {
  int nb_plane, nb_bbox, nb_corner, nb_ptcub;  // workgroups per edge class
};
constexpr int kNumericLdsDoubles = NumericShape<PlaneModel>::EPB * PlaneModel::DIM_MAX * 18 + 8 > 16 * 18 + 8 ? NumericShape<PlaneModel>::EPB * PlaneModel::DIM_MAX * 18 + 8
                                                                                                         : 16 * 18 + 8;  // planes: sJ + sW + sG + s_red of 12 edges; corner edges: 16 rows
using BboxModel = CubProjModelT<false>;
using CornerModel = CubProjModelT<true>;
// ONE launch for the whole linearisation (horizontal fusion): numerically differentiated edge classes | one wave per pose-major chunk
// (pose side, factor records) | landmark-major point workgroups.  The parts are independent (they meet only in atomics on H_pp), so
// the launch lasts as long as its slowest part instead of their sum; as separate kernels on two streams the fork/join events alone
// cost ~12 us per linearisation (tools/gap_report.py).  All parts share one dynamic LDS allocation.
// The launch holds more waves than the chip has slots (6 000 against 4 096 at the BASELINE size), so the ORDER of the parts in the
// grid is a scheduling decision: longest-running workgroups first (tools/lin_stamps.py prints every part's timeline).
// ---- LM control on the device: the decision behind every trial (OptimizationAlgorithmLevenberg::solve, levenberg.cpp:102-163) -------------
__global__ void k_ctl_set(LmCtl* ctl, LmCtl v) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *ctl = v;
}
// One wave.  Reads (chi2, scale) of the trial state and the factorisation flag, takes the accept / reject decision, updates the control
// record and writes the trial's numbers into its slot of the DEVICE ring of trial records.  `epoch`: the epoch the trial was queued
// under (a trial that did not run leaves a record that says so).  n_publish > 0 (the last trial of a batch): the records of the last
// n_publish trials go to the mapped host ring, then the sequence number the host waits for -- ONE hand-over per batch: a kernel that
// writes host memory holds its successor back until the write has crossed the bus (6 - 10 us, measured as idle queue in front of
// k_reduce_stage1 with a hand-over per trial; tools/small_window_gap.sh).
struct DecidePlan {
  double* rec;    // device ring, kRecSlots x REC_DOUBLES
  double* host;   // mapped host ring of the same shape, and behind it the sequence word
  unsigned long long seq;
  int n_publish;
  const double* host_stop;  // mapped host word: the host's mirror of the caller's stop flag, read before EVERY decision (round 6; NULL: not looked at)
};
constexpr int kRecSlots = 16;
// the body, by ONE wave (t = its lane): k_decide's, and (round 5) the tail of the launch that evaluates the trial state
// DET (deterministic mode): chi2 and the scale of the gain ratio were summed as fixed-point integers -- folded as integers, converted here; an addend
// that did not fit (ACC_DETBAD) makes the trial's system unusable: reported like a factorisation that failed (what lm_iteration does on the host).
template <bool COHERENT = false, bool DET = false>
__device__ __forceinline__ void decide_wave(int t, double* acc, const int* __restrict__ info_p, LmCtl* ctl, int epoch, const DecidePlan& dp) {
  double* rec = dp.rec + (dp.seq % kRecSlots) * REC_DOUBLES;
  // EVERYTHING the decision reads is requested here, side by side -- the epoch, the 64 partial sums, the totals in front of them, the flags, the control
  // record, and the stop flag's mirror over the bus (levenberg.cpp:149 reads terminate() before every retry) -- and folded in registers: one memory round
  // trip in front of the decision instead of four in a row with a device-scope fence between them (round 6: 5.3 -> 3.x us per decision).
  const double host_stop = dp.host_stop != nullptr ? __hip_atomic_load(dp.host_stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0.0;
  LmCtl c = *ctl;  // (every lane the same record: one request; its epoch member decides whether this attempt is live)
  int info = *info_p;
  [[maybe_unused]] const double f_range = acc[ACC_DETRANGE], f_bad = acc[ACC_DETBAD];
  const bool dead = c.epoch != epoch;
  double chi = 0.0, sc = 0.0;
  if constexpr (COHERENT) {  // (the tail of the evaluation launch: other workgroups' atomics were performed at the memory side -- device-scope operations, folded in memory)
    if (!dead) fold_acc_parts_coherent(acc);
    __threadfence();
    __builtin_amdgcn_wave_barrier();
    chi = __hip_atomic_load(&acc[ACC_CHI2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    sc = __hip_atomic_load(&acc[ACC_SCALE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    double* part = acc + ACC_N + (t & 63) * kAccPartStride;
    const double p0 = part[0], p1 = part[1], b0 = (t == 0) ? acc[ACC_CHI2] : 0.0, b1 = (t == 0) ? acc[ACC_SCALE] : 0.0;
    if constexpr (DET) {
      long long c = __double_as_longlong(p0) + __double_as_longlong(b0), q = __double_as_longlong(p1) + __double_as_longlong(b1);  // (integers: any order)
      for (int off = 32; off > 0; off >>= 1) { c += __shfl_down(c, off, 64); q += __shfl_down(q, off, 64); }
      chi = (double)c * kFixInv;
      sc = (double)q * kFixInv;
    } else {
      double c = p0, q = p1;  // (the order of fold_acc_parts: the partial sums by a shuffle tree, then onto the total)
      for (int off = 32; off > 0; off >>= 1) { c += __shfl_down(c, off, 64); q += __shfl_down(q, off, 64); }
      chi = b0 + c;
      sc = b1 + q;
    }
    if (!dead) { part[0] = 0.0; part[1] = 0.0; }  // (what fold_acc_parts leaves behind: the partial sums cleared)
  }
  if (t == 0 && dead) {
    rec[REC_DEAD] = 1.0;
    rec[REC_SEQ] = (double)dp.seq;
  }
  if (t == 0 && !dead) {
    if constexpr (DET) {
      if (f_range != 0.0) {  // a term that does not fit the fixed-point range, or a non-finite entry of a matrix: the trial's system is not usable
        if (info < (1 << 30)) info = 1 << 29;
        acc[ACC_DETRANGE] = 0.0;  // handed over; armed again for the next trial
      }
      if (f_bad != 0.0) {  // a non-finite term of chi2: the trial state's chi2 is NaN in the reference's arithmetic
        chi = __builtin_nan("");
        acc[ACC_DETBAD] = 0.0;
      }
    }
    if constexpr (!COHERENT) { acc[ACC_CHI2] = chi; acc[ACC_SCALE] = sc; }  // (as fold_acc_parts: the totals, as doubles, for whoever looks afterwards)
    lm_decide(c, chi, sc, info, rec);
    if (host_stop != 0.0) lm_decide_stop(c, epoch, rec);
    *ctl = c;
    rec[REC_SEQ] = (double)dp.seq;
  }
  if (dp.n_publish <= 0) return;
  __threadfence();
  __builtin_amdgcn_wave_barrier();
  for (int i = t; i < dp.n_publish * REC_DOUBLES; i += 64) {
    const unsigned long long s = dp.seq - (unsigned long long)(dp.n_publish - 1) + (unsigned long long)(i / REC_DOUBLES);
    const int at = (int)(s % kRecSlots) * REC_DOUBLES + i % REC_DOUBLES;
    dp.host[at] = __hip_atomic_load(dp.rec + at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __threadfence_system();
  __builtin_amdgcn_wave_barrier();
  if (t == 0) __hip_atomic_store(reinterpret_cast<unsigned long long*>(dp.host + kRecSlots * REC_DOUBLES), dp.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ __launch_bounds__(64) void k_decide(double* acc, const int* __restrict__ info_p, LmCtl* ctl, int epoch, DecidePlan dp) {
  decide_wave(threadIdx.x, acc, info_p, ctl, epoch, dp);
}
// Deterministic mode.  Workgroup 0 decides; the others turn the ranges the trial's (speculated) linearisation summed into -- H_pp | H_cc, the planes' D /
// b_l / W: fixed-point integers -- back into doubles, in the shadow of the decision, so that the mode needs no conversion launch of its own.  They act
// only when the linearisation in front of this launch left ITS tag (Graph::lin_tag): a linearisation that found its epoch gone wrote nothing and its target
// may hold an earlier linearisation that is doubles already.  (Not gated by the epoch itself: workgroup 0 may be moving it at this very moment; a
// converted linearisation of a trial that ends up rejected is simply never read.)
struct DecideFix {
  const unsigned long long* tag_word;
  unsigned long long tag;
  double* p[4];
  unsigned long long n[4];
};
__global__ __launch_bounds__(256) void k_decide_fix(double* acc, const int* __restrict__ info_p, LmCtl* ctl, int epoch, DecidePlan dp, DecideFix df) {
  if (blockIdx.x == 0) {
    if (threadIdx.x < 64) decide_wave<false, true>(threadIdx.x, acc, info_p, ctl, epoch, dp);
    return;
  }
  if (*df.tag_word != df.tag) return;
  const size_t t0 = (size_t)(blockIdx.x - 1) * 256 + threadIdx.x, stride = (size_t)(gridDim.x - 1) * 256;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    double* p = df.p[q];
    const size_t n = df.n[q];
    for (size_t i = t0; i < n; i += stride) p[i] = (double)__double_as_longlong(p[i]) * kFixInv;
  }
}
// The decision as the TAIL of the launch that evaluates the trial state (k_linearize_all on a speculated trial, k_chi2_all otherwise) instead of
// a 5 us launch of its own behind it: the workgroup that finishes LAST runs decide_wave.  "Last" through two levels of arrival counters
// (64 + 1, zero between launches: a launch of 1 500 workgroups would otherwise queue 1 500 atomics on one address at its very end).  Every
// workgroup's sums are in memory before its ticket is drawn (barrier, then a device-scope release fence by the thread that draws it).
struct DecideTail {
  DecidePlan dp;
  const int* info;
  LmCtl* ctl;
  int* cnt;  // [65]; NULL: no tail (the host takes the decisions, or k_decide follows)
};
__device__ __forceinline__ void lm_tail(const Graph& g, const DecideTail& dt) {
  if (dt.cnt == nullptr) return;
  __shared__ int s_last;
  // What the decision reads of this launch are SUMS -- device-scope atomics, performed at the memory side, complete once acknowledged.  So every
  // wave waits for its own memory operations (a workgroup-scope release: s_waitcnt, no cache maintenance), the barrier collects the waves, and
  // the ticket is drawn.  A device-scope release fence per workgroup instead (an L2 write-back each, 1 500 times per launch) cost 33 us per
  // LM iteration at the BASELINE size -- six times what the separate launch of k_decide costs.  The last workgroup reads the sums with
  // device-scope loads (fold_acc_parts_coherent).
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  if (threadIdx.x == 0) {
    const int G = (int)gridDim.x, p = (int)blockIdx.x & 63, expect = (G + 63 - p) >> 6;
    int last = 0;
    if (atomicAdd(&dt.cnt[p], 1) == expect - 1) {
      __hip_atomic_store(&dt.cnt[p], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (every workgroup of this part has drawn: nobody touches the counter again before the next launch)
      if (atomicAdd(&dt.cnt[64], 1) == min(G, 64) - 1) {
        __hip_atomic_store(&dt.cnt[64], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = 1;
      }
    }
    s_last = last;
  }
  __syncthreads();
  if (s_last && threadIdx.x < 64) decide_wave<true>((int)threadIdx.x, g.acc, dt.info, dt.ctl, g.epoch, dt.dp);
}
// a launch that finds its epoch gone still owes the host the trial's (dead) record
__device__ __forceinline__ void lm_tail_dead(const Graph& g, const DecideTail& dt) {
  if (dt.cnt != nullptr && blockIdx.x == 0 && threadIdx.x < 64) decide_wave((int)threadIdx.x, g.acc, dt.info, dt.ctl, g.epoch, dt.dp);
}

#ifndef PPO_LIN_WAVES
#define PPO_LIN_WAVES 3  // waves per SIMD the fused linearisation is compiled for (register budget 512 / this): 120 VGPRs, nothing spills,
                        // 4 waves per SIMD resident
#endif
enum { LIN_PLANE = 0, LIN_BBOX = 1, LIN_CORNER = 2, LIN_PTCUB = 3, LIN_POSE = 4, LIN_POINTS = 5, LIN_CUB3D = 6, LIN_PARTS = 7 };
struct LinPlan {
  int nb[LIN_PARTS];     // workgroups per part
  int order[LIN_PARTS];  // grid order of the parts
};
#ifdef PPO_LIN_MAXWAVES  // (A/B builds: cap the waves per SIMD from above)
#define PPO_LIN_OCC __attribute__((amdgpu_waves_per_eu(1, PPO_LIN_MAXWAVES)))
#else
#define PPO_LIN_OCC
#endif
template <bool USE_LDS, bool WITH_CORNERS, bool ANALYTIC = false>
__global__ __launch_bounds__(256, PPO_LIN_WAVES) PPO_LIN_OCC void k_linearize_all(Graph g, State st, LinPlan lp, PoseChunks pc, double delta, DecideTail tail) {
  extern __shared__ double smem[];
  if (lm_dead(g)) {
    lm_tail_dead(g, tail);
    return;
  }
  if (g.lin_tag_word != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *g.lin_tag_word = g.lin_tag;  // (deterministic + device LM control: this launch is live)
  int b = blockIdx.x, part = -1;
#pragma unroll
  for (int oi = 0; oi < LIN_PARTS; ++oi) {
    const int n = lp.nb[lp.order[oi]];
    if (part < 0) {
      if (b < n) part = lp.order[oi];
      else b -= n;
    }
  }
#ifdef PPO_LIN_ONLY
  if (part != PPO_LIN_ONLY) return;
#endif
  switch (part) {
    case LIN_PLANE:
      if (ANALYTIC) lin_plane_body_analytic(g, st, g.n_pl, b, smem);
      else lin_plane_body(g, st, g.n_pl, delta, b, smem);
      break;
    case LIN_BBOX:
      if (ANALYTIC) lin_cubproj_body_analytic<false>(g, st, g.n_bbox, b, smem);
      else lin_cubproj_body<false>(g, st, b, delta, smem);
      break;
    case LIN_CORNER:
      if (WITH_CORNERS) {  // compiled in only for windows that have corner edges
        if (ANALYTIC) lin_cubproj_body_analytic<true>(g, st, g.n_corner2d, b, smem);
        else lin_cubproj_body<true>(g, st, b, delta, smem);
      }
      break;
    case LIN_CUB3D:  // always numeric (the derivative of a minimum over four SE3 logarithms is not written out)
      if (WITH_CORNERS) lin_cub3d_body(g, st, g.n_corner2d + b, delta, smem);
      break;
    case LIN_PTCUB: lin_ptcub_body(g, st, b, delta, smem); break;
    case LIN_POSE: LIN_STAMP(g, 0); lin_poses_body(g, st, pc, b, smem); LIN_STAMP(g, 4); break;
    case LIN_POINTS: LIN_STAMP(g, 0); lin_points_body<USE_LDS>(g, st, b, lp.nb[LIN_POINTS], smem + 8, smem); LIN_STAMP(g, 4); break;
    default: break;
  }
  lm_tail(g, tail);
}
// residual-only pass of a trial state in ONE launch: numeric edge classes (thread per edge) | points (thread per point)
__global__ __launch_bounds__(256) void k_chi2_all(Graph g, State st, NumericPlan np, int nb_num, int nb_pts, DecideTail tail) {
  __shared__ double s_red[4];
  if (lm_dead(g)) {
    lm_tail_dead(g, tail);
    return;
  }
  int b = blockIdx.x;
  if (b >= nb_num) chi2_points_body(g, st, b - nb_num, nb_pts, s_red);
  else if (b < np.nb_plane) chi2_numeric_body<PlaneModel>(g, st, g.n_pl, false, b);
  else if ((b -= np.nb_plane) < np.nb_bbox) chi2_numeric_body<BboxModel>(g, st, g.n_bbox, false, b);
  else if ((b -= np.nb_bbox) < np.nb_corner) chi2_numeric_body<CornerModel>(g, st, g.n_corner, true, b);
  else chi2_numeric_body<PtCubModel>(g, st, g.n_ptcub, false, b - np.nb_corner);
  lm_tail(g, tail);
}

// Hands the accumulator block and the factorisation flag to the host through mapped, coherent host memory: payload first,
// then a sequence number the host spins on (a device-to-host copy engine transfer costs ~10 us of idle device per trial).
// One launch zeroes every accumulator range of a phase (instead of one memset node per range).
struct ZeroPlan {
  double* p[6];
  unsigned long long n[6];  // doubles
};
__device__ __forceinline__ void zero_ranges_body(const ZeroPlan& zp, size_t t0, size_t stride) {
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    double* p = zp.p[r];
    const size_t n = zp.n[r];
    if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {
      double2* p2 = reinterpret_cast<double2*>(p);
      for (size_t i = t0; i < n / 2; i += stride) p2[i] = make_double2(0.0, 0.0);
      if (t0 == 0 && (n & 1)) p[n - 1] = 0.0;
    } else {
      for (size_t i = t0; i < n; i += stride) p[i] = 0.0;
    }
  }
}
// (k_publish) Workgroup 0 hands the accumulator block and the factorisation flag to the host; when the next linearisation is
// queued right behind (speculation), the remaining workgroups clear its accumulation ranges in the same launch and
// workgroup 0 clears the accumulator block after copying it.
// The head of a pass on the device (round 6, device LM control): the hand-over launch behind the first linearisation of a pass also INITIALISES the control
// record -- lambda = tau * max diag H (computeLambdaInit, levenberg.cpp:166-180), chi2 of the estimate, _ni = 2, the epoch the first batch is queued under --
// so that the host queues that batch without waiting for the numbers (it learns them from the mapped block like everything else; the queue used to sit idle
// for ~24 us in front of k_ctl_set, twice per call).  The special cases of a head stay on the host: when a flag of the deterministic mode is up (a NaN chi2:
// frozen window; a term out of range: the handle leaves the fixed-point sums) the record is NOT written, its epoch stays the old one, and everything the
// host queued behind this launch finds itself dead; the host sees the flags with the first records it waits for (ppo_solver.hip: optimize_pipelined).
struct HeadInit {
  LmCtl* ctl;  // NULL: no head rides in this hand-over
  LmCtl m;     // the record to write: everything but lambda, chi_cur, ini_chi
  double tau;
};
__device__ __forceinline__ void head_init_record(const HeadInit& hi, double chi2, double maxdiag, double f_bad, double f_range) {
  if (hi.ctl == nullptr || f_bad != 0.0 || f_range != 0.0) return;
  LmCtl m = hi.m;
  m.chi_cur = m.ini_chi = chi2 + m.cubpl;
  m.lambda = hi.tau * maxdiag;
  *hi.ctl = m;
}
__global__ __launch_bounds__(64) void k_publish(double* acc, const int* __restrict__ info, double* host, unsigned long long seq, ZeroPlan zp, HeadInit hi) {
  const int t = threadIdx.x;
  if (blockIdx.x > 0) {
    zero_ranges_body(zp, (size_t)(blockIdx.x - 1) * 64 + t, (size_t)(gridDim.x - 1) * 64);
    return;
  }
  fold_acc_parts(acc, 0);  // (deterministic / sharded runs have folded already -- k_fold_acc -- and find zeros here)
  __syncthreads();
  double v = 0.0;
  if (t < ACC_N) { v = acc[t]; host[t] = v; }
  if (t == 0) head_init_record(hi, v, acc[ACC_MAXDIAG], acc[ACC_DETBAD], acc[ACC_DETRANGE]);
  __syncthreads();  // (thread 0 has read the flags)
  if (t == ACC_DETBAD || t == ACC_DETRANGE) acc[t] = 0.0;  // deterministic mode: "an addend did not fit" has been handed over; armed again for the next launches
  if (t == ACC_N) host[ACC_N] = (double)(info ? *info : 0);
  __threadfence_system();
  __syncthreads();
  if (t == 0) __hip_atomic_store(reinterpret_cast<unsigned long long*>(host + ACC_N + 1), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  if (gridDim.x > 1 && t < ACC_N) acc[t] = 0.0;
}

// Deterministic mode, one rank (round 6): the hand-over itself folds the INTEGER partial sums of chi2 and of the scale and converts them, and the other
// workgroups of its launch turn the ranges a linearisation in front of it summed into back into doubles (DecideFix; tag_word == NULL: unconditionally) --
// the k_fold_acc + two k_fix_to_double launches this mode used to put in front of every hand-over are gone.
__global__ __launch_bounds__(256) void k_publish_fix(double* acc, const int* __restrict__ info, double* host, unsigned long long seq, DecideFix df, HeadInit hi) {
  const int t = threadIdx.x;
  if (blockIdx.x > 0) {
    if (df.tag_word != nullptr && *df.tag_word != df.tag) return;
    const size_t t0 = (size_t)(blockIdx.x - 1) * 256 + t, stride = (size_t)(gridDim.x - 1) * 256;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      double* p = df.p[q];
      const size_t n = df.n[q];
      for (size_t i = t0; i < n; i += stride) p[i] = (double)__double_as_longlong(p[i]) * kFixInv;
    }
    return;
  }
  if (t < 64) fold_acc_parts(acc, 1);  // (lane 0 leaves the integer totals in acc[0..1] ...)
  if (t == 0) {                        // (... and converts them: chi2, scale)
    acc[ACC_CHI2] = (double)__double_as_longlong(acc[ACC_CHI2]) * kFixInv;
    acc[ACC_SCALE] = (double)__double_as_longlong(acc[ACC_SCALE]) * kFixInv;
  }
  __threadfence();
  __syncthreads();
  if (t < ACC_N) host[t] = acc[t];
  if (t == 0) head_init_record(hi, acc[ACC_CHI2], acc[ACC_MAXDIAG], acc[ACC_DETBAD], acc[ACC_DETRANGE]);
  __syncthreads();  // (thread 0 has read the flags)
  if (t == ACC_DETBAD || t == ACC_DETRANGE) acc[t] = 0.0;  // handed over; armed again for the next launches
  if (t == ACC_N) host[ACC_N] = (double)(info ? *info : 0);
  __threadfence_system();
  __syncthreads();
  if (t == 0) __hip_atomic_store(reinterpret_cast<unsigned long long*>(host + ACC_N + 1), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// the partial sums into acc[0..1] ahead of a reader other than k_publish (fixed-point conversion, sum over ranks)
__global__ __launch_bounds__(64) void k_fold_acc(double* acc, int det) { fold_acc_parts(acc, det); }

// reset of a solver in ONE launch: up to eight word-granular copies from the pristine device copies made at create, and the
// zero ranges.  (Seven hipMemcpyAsync / hipMemsetAsync calls took the runtime's blit path: ~15 us of queue idle each.)
struct CopyPlan {
  uint32_t* dst[8];
  const uint32_t* src[8];
  unsigned long long n[8];  // 32-bit words
};
// the small uploads of ppo_ba_create in ONE launch: every item is copied from page-locked host memory (read over the bus by the kernel itself) to its
// device block; 16-byte words, then the tail bytes.  One workgroup column per item (blockIdx.y), so that 60 items of a few KB do not queue behind
// each other's bus round trips.
struct GatherPlan {
  static constexpr int kItems = 56;
  void* dst[kItems];
  const void* src[kItems];
  unsigned int bytes[kItems];
};
__global__ __launch_bounds__(256) void k_gather_uploads(GatherPlan gp) {
  const int r = blockIdx.y;
  const size_t n = gp.bytes[r], n16 = n / 16;
  const uint4* s16 = static_cast<const uint4*>(gp.src[r]);
  uint4* d16 = static_cast<uint4*>(gp.dst[r]);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) d16[i] = s16[i];
  if (blockIdx.x == 0 && threadIdx.x < (n & 15)) static_cast<unsigned char*>(gp.dst[r])[n16 * 16 + threadIdx.x] = static_cast<const unsigned char*>(gp.src[r])[n16 * 16 + threadIdx.x];
}
__global__ __launch_bounds__(256) void k_reset(CopyPlan cp, ZeroPlan zp) {
  const size_t t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const size_t n = cp.n[r];
    for (size_t i = t0; i < n; i += stride) cp.dst[r][i] = cp.src[r][i];
  }
  zero_ranges_body(zp, t0, stride);
}
__global__ __launch_bounds__(256) void k_zero_ranges(ZeroPlan zp) {
  zero_ranges_body(zp, (size_t)blockIdx.x * blockDim.x + threadIdx.x, (size_t)gridDim.x * blockDim.x);
}
// deterministic mode: the accumulation ranges hold 2^-24 fixed-point integers after a kernel that summed into them -> doubles, in place
__global__ __launch_bounds__(256) void k_fix_to_double(ZeroPlan zp) {
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    double* p = zp.p[q];
    const size_t n = zp.n[q];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
      p[i] = (double)__double_as_longlong(p[i]) * kFixInv;
  }
}

// the same out of place: the reduced system summed into the fixed-point array -> doubles in S (consumers other than the native solver, which converts on load)
__global__ __launch_bounds__(256) void k_fix_to_double_copy(const double* __restrict__ src, double* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = (double)__double_as_longlong(src[i]) * kFixInv;
}

// max |H_jj| over every active diagonal (computeLambdaInit, levenberg.cpp:166-180)
__global__ void k_hb_diag(Graph g, double* out) {  // out[6 f + a] = Hpp_f[a][a]
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < g.n_pose_free * 6) out[i] = g.HB[(size_t)(i / 6) * 27 + packed_lower(i % 6, i % 6)];
}
// (grid-stride, one atomic per workgroup: with one atomic per wave of a 150 k-thread grid the 2 350 updates of the single
//  address serialised into 29 us)
__global__ __launch_bounds__(256) void k_maxdiag(Graph g) {
  __shared__ double s_m[4];
  const int nl_owned = (g.pt_last - g.pt_first) + (g.plane_last - g.plane_first);
  const int n = max(max(g.n_pose_free * 6, g.n_cuboids * 9), nl_owned * 3);
  // (g.lin_raw: H_pp | H_cc and the PLANES' D were summed as fixed-point integers and have not been converted yet; the points' D are plain stores)
  auto summed = [&](const double* p) { return g.lin_raw ? (double)__double_as_longlong(*p) * kFixInv : *p; };
  double m = 0.0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    if (i < g.n_pose_free * 6) m = fmax(m, fabs(g.hb_diag ? g.hb_diag[i] : summed(&g.HB[(size_t)(i / 6) * 27 + packed_lower(i % 6, i % 6)])));
    if (i < g.n_cuboids * 9 && g.owns_reduced) m = fmax(m, fabs(summed(&g.HC[(size_t)(i / 9) * 54 + packed_lower(i % 9, i % 9)])));
    if (i < nl_owned * 3) {
      const int k = i / 3, a = i % 3;
      const int l = k < (g.pt_last - g.pt_first) ? g.pt_first + k : g.n_points + g.plane_first + (k - (g.pt_last - g.pt_first));
      const bool fixed = l < g.n_points && g.point_fixed[l];
      const double* d = &g.D[(size_t)l * 6 + (a == 0 ? 0 : (a == 1 ? 3 : 5))];
      if (!fixed) m = fmax(m, fabs(l >= g.n_points ? summed(d) : *d));
    }
  }
  for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_down(m, off, 64));
  if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmax(fmax(s_m[0], s_m[1]), fmax(s_m[2], s_m[3]));
    if (m > 0.0) atomic_max_pos(&g.acc[ACC_MAXDIAG], m);
  }
}

// S(lower) += Hpp (+ lambda), Hcc (+ lambda), pose-cuboid blocks; bs = b.   S was zeroed beforehand.
__device__ __forceinline__ void assemble_body(const Graph& g, double lambda, int vb) {
  const int i = vb * blockDim.x + threadIdx.x;
  const int n = g.ld;
  const int nP = g.n_pose_free * 27, nC = g.n_cuboids * 54, nB = g.n_bbox * 54, nK = g.n_corner * 54;
  // sharded runs: every rank adds ITS partial pose blocks (the all-reduce of S sums them); lambda, the padding identity and
  // the cuboid terms come from the rank that owns the reduced terms
  if (!g.owns_reduced && i >= nP) return;
  if (i >= nP + nC + nB + nK) {  // identity on the padding so that the padded matrix stays positive definite
    const int k = g.n_red + (i - (nP + nC + nB + nK));
    if (k < g.ld) g.S[(size_t)k * n + k] = g.det ? __longlong_as_double((long long)kFixScale) : 1.0;  // (deterministic mode: S holds fixed-point integers until k_fix_to_double)
    return;
  }
  if (i < nP) {
    const int f = i / 27, k = i % 27;
    const int col0 = f * 6;  // free poses occupy the leading 6*n_pose_free columns in free-id order
    const double v = g.HB[i];
    if (k < 21) {
      int r = 0;
      while ((r + 1) * (r + 2) / 2 <= k) ++r;
      const int c = k - r * (r + 1) / 2;
      const double add = v + ((r == c && g.owns_reduced) ? lambda : 0.0);
      acc_add(g, &g.S[(size_t)(col0 + c) * n + col0 + r], add);
    } else {
      acc_add(g, &g.bs[col0 + (k - 21)], v);
    }
    return;
  }
  int j = i - nP;
  if (g.elim_cub) return;  // cuboid blocks enter through k_cub_schur instead
  if (j < nC) {
    if (!g.owns_reduced) return;
    const int cb = j / 54, k = j % 54;
    const int col0 = g.cub_col[cb];
    if (col0 < 0) return;
    const double v = g.HC[j];
    if (k < 45) {
      int r = 0;
      while ((r + 1) * (r + 2) / 2 <= k) ++r;
      const int c = k - r * (r + 1) / 2;
      acc_add(g, &g.S[(size_t)(col0 + c) * n + col0 + r], v + (r == c ? lambda : 0.0));
    } else {
      acc_add(g, &g.bs[col0 + (k - 45)], v);
    }
    return;
  }
  j -= nC;
  if (j < nB + nK) {
    if (!g.owns_reduced) return;
    const bool corner = j >= nB;
    const int jj = corner ? j - nB : j;
    const int e = jj / 54, k = jj % 54, a = k / 9, c = k % 9;
    const CubEdge& ed = corner ? g.corner[e] : g.bbox[e];
    if (!(ed.flags & F_ACTIVE)) return;
    const int pc = g.pose_col[ed.pose], cc = g.cub_col[ed.cuboid];
    if (pc < 0 || cc < 0) return;
    const double v = (corner ? g.Wpc_corner : g.Wpc_bbox)[(size_t)e * 54 + k];
    acc_add(g, &g.S[(size_t)(pc + a) * n + cc + c], v);  // row = cuboid column (cuboids come after poses)
  }
}

__device__ __forceinline__ void schur_prep_body(const Graph& g, double lambda, int vb) {
  const int nl_pts = g.pt_last - g.pt_first;
  const int k = vb * blockDim.x + threadIdx.x;
  if (k >= nl_pts + (g.plane_last - g.plane_first)) return;
  const int l = k < nl_pts ? g.pt_first + k : g.n_points + g.plane_first + (k - nl_pts);
  if (l < g.n_points && g.point_fixed[l]) return;
  const double d00 = g.D[(size_t)l * 6] + lambda, d01 = g.D[(size_t)l * 6 + 1], d02 = g.D[(size_t)l * 6 + 2];
  const double d11 = g.D[(size_t)l * 6 + 3] + lambda, d12 = g.D[(size_t)l * 6 + 4], d22 = g.D[(size_t)l * 6 + 5] + lambda;
  // D + lambda I = L L' (3x3), Linv = L^-1 (lower), Dinv = Linv' Linv
  const double l00 = sqrt(d00), i00 = 1.0 / l00;
  const double l10 = d01 * i00, l20 = d02 * i00;
  const double l11 = sqrt(d11 - l10 * l10), i11 = 1.0 / l11;
  const double l21 = (d12 - l20 * l10) * i11;
  const double l22 = sqrt(d22 - l20 * l20 - l21 * l21), i22 = 1.0 / l22;
  const double i10 = -l10 * i00 * i11;
  const double i21 = -l21 * i11 * i22;
  const double i20 = -(l20 * i00 + l21 * i10) * i22;
  double* I = &g.Dinv[(size_t)l * 6];
  const double v00 = i00 * i00 + i10 * i10 + i20 * i20, v01 = i10 * i11 + i20 * i21, v02 = i20 * i22;
  const double v11 = i11 * i11 + i21 * i21, v12 = i21 * i22, v22 = i22 * i22;
  I[0] = v00; I[1] = v01; I[2] = v02; I[3] = v11; I[4] = v12; I[5] = v22;
  const double b0 = g.bl[(size_t)l * 3], b1 = g.bl[(size_t)l * 3 + 1], b2 = g.bl[(size_t)l * 3 + 2];
  g.ul[(size_t)l * 3] = v00 * b0 + v01 * b1 + v02 * b2;  // (D + lambda I)^-1 b_l
  g.ul[(size_t)l * 3 + 1] = v01 * b0 + v11 * b1 + v12 * b2;
  g.ul[(size_t)l * 3 + 2] = v02 * b0 + v12 * b1 + v22 * b2;
}

// (PairChunks / schur_pairs_body: ppo_pairs.hpp -- shared with the fused launches of ppo_chol.hip)

// ---- cuboid elimination -------------------------------------------------------------------------------------------
// The reference keeps the 9-DoF cuboids in the dense system (they are not marginalised, Optimizer.cc:2172).  Here they
// are eliminated like landmarks: Dc = Hcc + lambda I = Lc Lc', S_pp -= sum_{a,b} (W_a Lc^-T)(W_b Lc^-T)', which is the
// same linear solve with a smaller dense factorisation (BASELINE.json north_star asks for exactly this).
__device__ __forceinline__ const CubEdge& cub_edge_of(const Graph& g, int code) { return (code & 1) ? g.corner[code >> 1] : g.bbox[code >> 1]; }
__device__ __forceinline__ const double* wpc_of(const Graph& g, int code) { return ((code & 1) ? g.Wpc_corner : g.Wpc_bbox) + (size_t)(code >> 1) * 54; }

// One cuboid per ROW OF 16 LANES (round 3; as one thread per cuboid walking the 9x9 Cholesky factor, its inverse and their product in
// registers this was a 3.7 us chain, the longest of stage 1's launch): lane r < 9 holds row r of Hcc + lambda I and row r of the identity;
// the right-looking elimination is the register factorisation of ppo_chol.hip in small -- the broadcast-multiply-adds are single
// v_fmac_f64 row_newbcast instructions, the identity rows come out as L^-T (the elimination is a right-multiplication by L^-T) -- and
// (Hcc + lambda I)^-1 = L^-T L^-1 is nine broadcast dot products per lane.  A pivot that is not positive gives NaN, which propagates to S
// and fails the dense factorisation as before.
template <int J, int C>
struct CubColUpdate {
  static __device__ __forceinline__ void run(double (&a)[9], double (&e)[9], double l, double le) {
    fmac_bcast<C, C == J + 1>(a[C], l, l);
    fmac_bcast<C, false>(e[C], l, le);
    if constexpr (C + 1 < 9) CubColUpdate<J, C + 1>::run(a, e, l, le);
  }
};
template <int J>
struct CubColSweep {
  static __device__ __forceinline__ void run(double (&a)[9], double (&e)[9]) {
    const double d = row_bcast<J>(a[J]);
    const double y0 = __builtin_amdgcn_rsq(d);  // (5e-8) + one third-order correction: 2.2e-16, see ppo_chol.hip
    const double e1 = __builtin_fma(-(d * y0), y0, 1.0);
    const double y = (d > 0.0) ? __builtin_fma(y0, e1 * __builtin_fma(0.375, e1, 0.5), y0) : __builtin_nan("");
    const double l = a[J] * y, le = e[J] * y;
    a[J] = l;
    e[J] = le;
    if constexpr (J + 1 < 9) {
      CubColUpdate<J, J + 1>::run(a, e, l, le);
      CubColSweep<J + 1>::run(a, e);
    }
  }
};
template <int B>
struct CubDinvRow {  // acc[b] = -sum_k e_b[k] e_mine[k]
  static __device__ __forceinline__ void run(const double (&e)[9], double (&acc)[9]) {
    acc[B] = 0.0;
#pragma unroll
    for (int k = 0; k < 9; ++k) fmac_bcast<B, false>(acc[B], e[k], e[k]);
    if constexpr (B + 1 < 9) CubDinvRow<B + 1>::run(e, acc);
  }
};
constexpr int kCubPerBlock = 16;  // 256 threads: 16 rows of 16 lanes
// the row of 16 lanes this thread belongs to factors cuboid c (c < 0: an idle row); s_di != nullptr: (Hcc + lambda I)^-1 also goes to LDS
__device__ __forceinline__ void cub_factor_row(const Graph& g, double lambda, int c, double* s_di) {
  const int r = threadIdx.x & 15;
  const bool live = c >= 0 && c < g.n_cuboids && g.cub_col[c] >= 0 && r < 9;
  double a[9], e[9];
  const double* hc = &g.HC[(size_t)(live ? c : 0) * 54];
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int rr = r < 9 ? r : 0;
    const double v = hc[rr >= k ? packed_lower(rr, k) : packed_lower(k, rr)] + (rr == k ? lambda : 0.0);
    a[k] = live ? v : (r == k ? 1.0 : 0.0);  // (idle lanes and rows carry the identity: no NaN traffic)
    e[k] = (r == k) ? 1.0 : 0.0;
  }
  CubColSweep<0>::run(a, e);
  // lane r now holds row r of L in a[0 .. r] and row r of L^-T in e[r .. 8] (zeros in front)
  double acc[9];
  CubDinvRow<0>::run(e, acc);
  if (!live) return;
  double* out = &g.cub_Linv[(size_t)c * 81];  // L^-1, row-major, lower: L^-1[i][r] = (L^-T)[r][i]
#pragma unroll
  for (int i = 0; i < 9; ++i) out[i * 9 + r] = (i >= r) ? e[i] : 0.0;
  double* di = &g.cub_Dinv[(size_t)c * 81];  // (Hcc + lambda I)^-1 = L^-T L^-1, full symmetric storage
#pragma unroll
  for (int b = 0; b < 9; ++b) {
    di[r * 9 + b] = -acc[b];
    if (s_di) s_di[r * 9 + b] = -acc[b];
  }
}
__device__ __forceinline__ void cub_prep_body(const Graph& g, double lambda, int vb) {
  cub_factor_row(g, lambda, vb * kCubPerBlock + (int)(threadIdx.x >> 4), nullptr);
}
// one (edge a >= edge b) pair of a cuboid, lane i < 6 of its 8 lanes; Di: (Hcc + lambda I)^-1 of the pair's cuboid (global memory or LDS)
struct CubPairOps {  // what a pair needs besides the cuboid's inverse: requested in one go (fused stage 1: BEFORE the inverse exists)
  int ca, cb, cola, colb;
  uint32_t fa, fb;
  double wa[9], zb[54];
  bool ok;
};
template <bool WITH_ZB>
__device__ __forceinline__ void cub_pair_load(const Graph& g, int pr, int i, bool valid, CubPairOps& o) {
  o.ok = false;
  if (!valid || i >= 6) return;
  o.ca = g.cp_a[pr]; o.cb = g.cp_b[pr];
  o.cola = g.cp_cola[pr]; o.colb = g.cp_colb[pr];
  if (o.cola < 0 || o.colb < 0) return;
  // (the activity flags are the one thing of a pair that changes between the passes: read next to the operands, applied at the end)
  o.fa = cub_edge_of(g, o.ca).flags; o.fb = cub_edge_of(g, o.cb).flags;
  // Every operand is requested up front (fully unrolled, independent loads): as loops of load -> multiply-add this body was a 10 us chain of
  // L2 round trips -- hidden behind the pair workgroups of a full reduction, but the whole length of a short launch
  const double* Wa = wpc_of(g, o.ca) + i * 9;
  const double* Zb = wpc_of(g, o.cb);
#pragma unroll
  for (int m = 0; m < 9; ++m) o.wa[m] = Wa[m];
  if constexpr (WITH_ZB) {
#pragma unroll
    for (int m = 0; m < 54; ++m) o.zb[m] = Zb[m];
  }
  o.ok = true;
}
// lane i < 6 of the pair's 8 lanes: row i of W_a (Hcc + lambda I)^-1 W_b' and of the right-hand side; Di: the cuboid's inverse (global memory or LDS)
// (DI_LDS: the inverse lies in LDS and is read where it is used -- as 81 register copies next to the pair's operands this body needed 337
//  registers and halved the occupancy of every other part of the fused stage-1 launch)
template <bool DI_LDS>
__device__ __forceinline__ void cub_pair_compute(const Graph& g, int pr, int i, const CubPairOps& o, const double* Di) {
  if (!o.ok) return;
  const int ca = o.ca, cb = o.cb, cola = o.cola, colb = o.colb;
  double za[9];
  double zb_late[DI_LDS ? 54 : 1];
  if constexpr (DI_LDS) {
    // (fused stage 1: edge b's block is requested here, not ahead of the factorisation -- 108 registers held across it were what pushed the
    //  launch to one wave per SIMD -- and lands while the row of W_a (Hcc + lambda I)^-1 is formed out of LDS)
    const double* Zb = wpc_of(g, cb);
#pragma unroll
    for (int m = 0; m < 54; ++m) zb_late[m] = Zb[m];
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      double v = 0.0;
#pragma unroll
      for (int m = 0; m < 9; ++m) v += o.wa[m] * Di[m * 9 + q];
      za[q] = v;
    }
  } else {
    double di[81];
#pragma unroll
    for (int m = 0; m < 81; ++m) di[m] = Di[m];
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      double v = 0.0;
#pragma unroll
      for (int m = 0; m < 9; ++m) v += o.wa[m] * di[m * 9 + q];
      za[q] = v;
    }
  }
  if (!(o.fa & F_ACTIVE) || !(o.fb & F_ACTIVE)) return;
  const double* zb = DI_LDS ? zb_late : o.zb;
  const bool self = ca == cb;
  const int n = g.ld;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    double m = 0.0;  // M[i][j] = (Za Zb')[i][j]: row index on pose a, column index on pose b
#pragma unroll
    for (int q = 0; q < 9; ++q) m += za[q] * zb[j * 9 + q];
    if (self) {
      if (i >= j) acc_add(g, &g.S[(size_t)(cola + j) * n + cola + i], -m);
    } else if (cola > colb) {
      acc_add(g, &g.S[(size_t)(colb + j) * n + cola + i], -m);
    } else if (cola < colb) {
      acc_add(g, &g.S[(size_t)(cola + i) * n + colb + j], -m);
    } else {  // two different edges on the same keyframe: M + M' lands in one diagonal block
      if (i >= j) acc_add(g, &g.S[(size_t)(cola + j) * n + cola + i], -m);
      if (j >= i) acc_add(g, &g.S[(size_t)(cola + i) * n + cola + j], -m);
    }
  }
  if (self) {
    const double* u = &g.HC[(size_t)g.cp_cub[pr] * 54 + 45];  // b_c
    double v = 0.0;
#pragma unroll
    for (int q = 0; q < 9; ++q) v += za[q] * u[q];
    acc_add(g, &g.bs[cola + i], -v);
  }
}

template <bool DI_LDS>
__device__ __forceinline__ void cub_pair_body(const Graph& g, int pr, int i, const double* Di) {
  CubPairOps o;
  cub_pair_load<!DI_LDS>(g, pr, i, true, o);
  cub_pair_compute<DI_LDS>(g, pr, i, o, Di);
}
__device__ __forceinline__ void cub_schur_body(const Graph& g, int vb) {
  const int t = vb * blockDim.x + threadIdx.x;
  const int pr = t >> 3;
  if (pr >= g.n_cub_pairs) return;
  cub_pair_body<false>(g, pr, t & 7, &g.cub_Dinv[(size_t)g.cp_cub[pr] * 81]);
}
// ONE workgroup per cuboid: its 9x9 block factored and inverted by the first row of 16 lanes, then all its edge pairs against the
// inverse in LDS -- preparation and use in one workgroup, so that the cuboid terms need no launch of their own behind stage 1
// (kCubFusedSplit workgroups per cuboid, each factoring the block for itself -- 1.5 us -- and taking every kCubFusedSplit-th round of 32 pairs:
//  a cuboid seen from ten keyframes has 55 pairs, as two rounds in one workgroup the chain was 16 us)
constexpr int kCubFusedSplit = 2;
__device__ __forceinline__ void cub_fused_body(const Graph& g, double lambda, int vb, double* s_di) {
  const int c = vb / kCubFusedSplit, part = vb % kCubFusedSplit;
  const int per_round = (int)(blockDim.x >> 3), i = threadIdx.x & 7;
  const bool live = g.cub_col[c] >= 0;
  // this thread's pair of the first round: its operands are requested BEFORE the block is factored (they do not depend on it)
  int pr = g.cp_off[c] + part * per_round + (int)(threadIdx.x >> 3);
  const int pr_end = g.cp_off[c + 1];
  CubPairOps o;
  cub_pair_load<false>(g, pr, i, live && pr < pr_end, o);
  if (threadIdx.x < 64) cub_factor_row(g, lambda, threadIdx.x < 16 ? c : -1, s_di);  // (every part writes the same cub_Linv / cub_Dinv: identical values)
  __syncthreads();
  if (!live) return;
  cub_pair_compute<true>(g, pr, i, o, s_di);
  for (pr += kCubFusedSplit * per_round; pr < pr_end; pr += kCubFusedSplit * per_round) cub_pair_body<true>(g, pr, i, s_di);
}

// ---- the reduction to the pose system in three launches (horizontal fusion of independent parts) ---------------------------
// stage 1: assemble Hpp + lambda I into S | invert the landmark blocks | invert the cuboid blocks
// stage 3: S -= W D^-1 W^T over pose pairs (wave per pair chunk) | over cuboid edge pairs
// (there is no stage 2 any more: it streamed Z = W L^-T to memory)
// The cuboid parts are tiny; as separate kernels on a side stream their fork/join events cost more than they hid.
struct ReducePlan {
  int nb_asm, nb_prep, nb_cubprep, nb_zero;
  int nb_pairs, nb_cubschur;
  int nb_cubfused, nb_pairs1;  // stage 1 of an overlapped reduction: one workgroup per cuboid (block + its pairs) and the pair chunks [0, chunk_hi)
                               // with their landmark blocks inverted on the fly -- no stage-3 launch at all
  int chunk_hi;  // the pair workgroups of stage 3 reduce the chunks [0, chunk_hi): all of them, or -- overlapped reduction -- those of the
                 // first column blocks of S (the fused launches of the dense factorisation take the rest, ppo_chol.hip)
};
inline PairCtx pair_ctx_host(const Graph& g) { return PairCtx{g.linR, g.F, g.W, g.Dinv, g.ul, g.S, g.bs, g.n_points, g.n_pt_slots, g.ld, g.D, g.bl, 0.0}; }
__device__ __forceinline__ PairCtx pair_ctx(const Graph& g, double lambda = 0.0) { return PairCtx{g.linR, g.F, g.W, g.Dinv, g.ul, g.S, g.bs, g.n_points, g.n_pt_slots, g.ld, g.D, g.bl, lambda}; }
// FUSED: stage 1 of an overlapped reduction, with the cuboid workgroups and the pair workgroups of the first column block(s) (a second
// instantiation: the pair code needs every register a wave can have, the plain stage 1 a third of them)
template <bool FUSED>
__global__ __launch_bounds__(256, 2) void k_reduce_stage1(Graph g, double lambda, ReducePlan rp, ZeroPlan zother, PairChunks pc) {
  __shared__ double s_buf[FUSED ? 4 : 1][FUSED ? 21 : 2][65];  // (FUSED: pair workgroups; the cuboid workgroups keep their inverse in its first 81 entries)
  if (lm_dead(g)) return;
  lambda = lm_lambda(g, lambda);
  int b = blockIdx.x;
  if (b == 0 && threadIdx.x < 2) g.acc[threadIdx.x] = 0.0;  // chi2 and scale of the trial this reduction leads to
  // (acc[ACC_DETBAD] is NOT cleared here: assemble_body of this very launch may raise it -- k_publish clears it once it is copied out)
  if (b == 0 && threadIdx.x >= 64 && threadIdx.x < 64 + kAccParts) {  // ... and their partial sums
    double* p = g.acc + ACC_N + (threadIdx.x - 64) * kAccPartStride;
    p[0] = 0.0;
    p[1] = 0.0;
  }
  // (the cuboid workgroup first: one thread per cuboid walks a 9x9 Cholesky + inverse, the longest chain of the launch)
  if constexpr (FUSED) {
    if (b < rp.nb_cubfused) { cub_fused_body(g, lambda, b, &s_buf[0][0][0]); return; }
    b -= rp.nb_cubfused;
    if (b < rp.nb_pairs1) {
      schur_pairs_body<4, true>(pair_ctx(g, lambda), pc, 0, rp.chunk_hi, b, rp.nb_pairs1, s_buf, [&](double* p, double v) { acc_add(g, p, v); });
      return;
    }
    b -= rp.nb_pairs1;
  }
  if (b < rp.nb_cubprep) { cub_prep_body(g, lambda, b); return; }
  b -= rp.nb_cubprep;
  if (b < rp.nb_asm) { assemble_body(g, lambda, b); return; }
  b -= rp.nb_asm;
  if (b < rp.nb_prep) { schur_prep_body(g, lambda, b); return; }
  b -= rp.nb_prep;
  // the linearisation buffer that is NOT current (ppo_solver.hip, LinBuf): cleared here for the linearisation of the trial state
  zero_ranges_body(zother, (size_t)b * 256 + threadIdx.x, (size_t)rp.nb_zero * 256);
}
__global__ __launch_bounds__(256) void k_reduce_stage3(Graph g, PairChunks pc, ReducePlan rp) {
  __shared__ double s_buf[4][21][65];
  if (lm_dead(g)) return;
  const int b = blockIdx.x;
  // the few cuboid workgroups first (a chain of dependent gathers: at the end of the grid they were a 5 us tail behind the pair
  // workgroups); their count is a multiple of 8, so the pair workgroups keep their workgroup -> XCD phase
  if (b < rp.nb_cubschur) { cub_schur_body(g, b); return; }
  schur_pairs_body<4, false>(pair_ctx(g), pc, 0, rp.chunk_hi, b - rp.nb_cubschur, rp.nb_pairs, s_buf, [&](double* p, double v) { acc_add(g, p, v); });
}
// single-part wrappers (measurement hooks)
__global__ __launch_bounds__(256) void k_schur_pairs(Graph g, PairChunks pc) {
  __shared__ double s_buf[4][21][65];
  schur_pairs_body<4, false>(pair_ctx(g), pc, 0, pc.n_chunks, blockIdx.x, gridDim.x, s_buf, [&](double* p, double v) { acc_add(g, p, v); });
}

// Poses and cuboids in ONE launch of 64-thread workgroups: workgroup c < n_cuboids back-substitutes cuboid c (when the cuboids
// were eliminated) and applies its update, the remaining workgroups update 64 poses each.
// (body: any workgroup size >= 64; workgroup vb < n_cuboids handles cuboid vb with its first 64 threads)
__device__ __forceinline__ void update_reduced_body(const Graph& g, const State& cur, const State& trial, double lambda, int vb, double* sp, double* st,
                                                    double* sy, double* sx) {
  const int lane = threadIdx.x;
  double sc = 0.0;
  if (vb < g.n_cuboids) {
    const int c = vb;
    const int col = g.cub_col[c];
    if (col >= 0 && g.elim_cub) {
      const int a = lane / 9;
      double part = 0.0;
      if (lane < 54) {
        // eight edges per trip, their loads independent of each other (as a loop of edge list -> edge -> pose -> column -> x, one edge after the
        // other, this was a 15 us chain -- the longest of the update launch); the products are added in list order as before
        const int q0 = g.cub_edge_off[c], q1 = g.cub_edge_off[c + 1];
        for (int q = q0; q < q1; q += 8) {
          int code[8], pcol[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int qq = min(q + u, q1 - 1);
            code[u] = g.cub_edges[qq];
            pcol[u] = g.cub_edge_col[qq];
          }
          double w[8], xv[8];
          uint32_t fl[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            fl[u] = cub_edge_of(g, code[u]).flags;
            w[u] = wpc_of(g, code[u])[lane];
            xv[u] = g.xr[max(pcol[u], 0) + a];
          }
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (q + u < q1 && (fl[u] & F_ACTIVE) && pcol[u] >= 0) part += w[u] * xv[u];
        }
        sp[lane] = part;
      }
      __syncthreads();
      if (lane < 9) {
        double t = g.HC[(size_t)c * 54 + 45 + lane];
        for (int k = 0; k < 6; ++k) t -= sp[k * 9 + lane];
        st[lane] = t;
      }
      __syncthreads();
      const double* Li = &g.cub_Linv[(size_t)c * 81];
      if (lane < 9) {
        double v = 0.0;
        for (int m = 0; m <= lane; ++m) v += Li[lane * 9 + m] * st[m];
        sy[lane] = v;
      }
      __syncthreads();
      if (lane < 9) {
        double v = 0.0;
        for (int k = lane; k < 9; ++k) v += Li[k * 9 + lane] * sy[k];
        g.xc[(size_t)c * 9 + lane] = v;
        sx[lane] = v;
      }
      __syncthreads();
    } else if (col >= 0) {
      if (lane < 9) sx[lane] = g.xr[col + lane];
      __syncthreads();
    }
    if (lane == 0) {
      Cub cb = load_cub(&cur.cub[10 * c]);
      if (col >= 0) {
        const double* fs = &g.cub_fixedscale[3 * c];
        cb = cuboid_boxplus(cb, sx, g.cub_flags[c], fs[0], fs[1], fs[2]);
        if (g.owns_reduced) {
          const double* b = &g.HC[(size_t)c * 54 + 45];
          for (int a = 0; a < 9; ++a) sc += sx[a] * (lambda * sx[a] + b[a]);
        }
      }
      double* o = &trial.cub[10 * c];
      o[0] = cb.T.q.x; o[1] = cb.T.q.y; o[2] = cb.T.q.z; o[3] = cb.T.q.w; o[4] = cb.T.t.x; o[5] = cb.T.t.y; o[6] = cb.T.t.z;
      o[7] = cb.s.x; o[8] = cb.s.y; o[9] = cb.s.z;
    }
  } else {
    const int i = (vb - g.n_cuboids) * (int)blockDim.x + lane;
    if (i < g.n_poses) {
      const int col = g.pose_col[i];
      Pose T = load_pose(&cur.pose_qt[7 * i]);
      if (col >= 0) {
        const double* x = &g.xr[col];
        T = pose_boxplus(T, x);
        const double* b = &g.HB[(size_t)g.pose_free_id[i] * 27 + 21];  // (sharded: this rank's part of b)
        const double lam = g.owns_reduced ? lambda : 0.0;
        for (int a = 0; a < 6; ++a) sc += x[a] * (lam * x[a] + b[a]);
      }
      store_pose(&trial.pose_qt[7 * i], &trial.pose_rt[12 * i], T);
    }
  }
  sc = wave_sum(sc);
  if ((lane & 63) == 0 && sc != 0.0) acc_add(g, acc_part(g, ACC_SCALE, threadIdx.x >> 6), sc);
}

// landmarks: x_l = Dinv (b_l - W' x_p) (block_solver.hpp:461-481), trial = cur [+] x_l, scale term.
// Points evaluate W' x_p = Jl' w (Jp x_p) from their own edges (streamed landmark-major) instead of reading W back;
// the few planes read their W blocks.
// LPP lanes share one landmark: they split its edges and meet in two shuffles (a point has ~5 edges and the whole graph only
// 50k points: one thread per point leaves most of the chip idle and every thread walking its edges one after the other).
template <bool USE_LDS, int LPP>
__device__ __forceinline__ void backsub_body(const Graph& g, const State& cur, const State& trial, double lambda, int vb, double* smem, double* s_red) {
  if (USE_LDS) {  // camera table: R, t, intrinsics of every keyframe
    stage_camera_table(smem, cur.pose_rt, g.pose_cam, g.n_poses);
    __syncthreads();
  }
  const int nl_pts = g.pt_last - g.pt_first;
  const int k = (vb * blockDim.x + threadIdx.x) / LPP, sub = threadIdx.x % LPP;
  double sc = 0.0;
  if (k < nl_pts + (g.plane_last - g.plane_first)) {
    const int l = k < nl_pts ? g.pt_first + k : g.n_points + g.plane_first + (k - nl_pts);
    const bool fixed = l < g.n_points && g.point_fixed[l];
    double xl[3] = {0, 0, 0};
    if (!fixed) {
      const double b0 = g.bl[(size_t)l * 3], b1 = g.bl[(size_t)l * 3 + 1], b2 = g.bl[(size_t)l * 3 + 2];
      double c0 = 0, c1 = 0, c2 = 0;
      if (l < g.n_points) {
        const Vec3 X = v3(cur.point[3 * l], cur.point[3 * l + 1], cur.point[3 * l + 2]);
        for (int e = g.mo_off[l] + sub; e < g.mo_off[l + 1]; e += LPP) {
          const uint32_t fl = g.mono_flags[e];
          if (!(fl & F_ACTIVE)) continue;
          const MonoEdge ed = g.mono[e];
          const int col = g.pose_col[ed.pose];
          if (col < 0) continue;
          const double* rt = USE_LDS ? &smem[ed.pose * kCamRow] : &cur.pose_rt[ed.pose * 12];
          const double* cm = USE_LDS ? &smem[ed.pose * kCamRow + 12] : &g.pose_cam[ed.pose * 5];
          Mat3 R;
#pragma unroll
          for (int q = 0; q < 9; ++q) R.a[q] = rt[q];
          double ev[2], Jl[6], Jp[12], zc;
          mono_eval(R, v3(rt[9], rt[10], rt[11]), X, cm[0], cm[1], cm[2], cm[3], ed.u, ed.v, ev, Jl, Jp, zc);
          const double chi2 = ev[0] * (ed.info * ev[0]) + ev[1] * (ed.info * ev[1]);
          double rho0, rho1;
          huber(chi2, (fl & F_ROBUST) ? ed.huber : 0.0, rho0, rho1);
          const double w = rho1 * ed.info;
          const double* x = &g.xr[col];
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            double tr = 0;
#pragma unroll
            for (int a = 0; a < 6; ++a) tr += Jp[6 * r + a] * x[a];
            tr *= w;
            c0 -= Jl[3 * r] * tr; c1 -= Jl[3 * r + 1] * tr; c2 -= Jl[3 * r + 2] * tr;
          }
        }
        for (int e = g.st_off[l] + sub; e < g.st_off[l + 1]; e += LPP) {
          const uint32_t fl = g.stereo_flags[e];
          if (!(fl & F_ACTIVE)) continue;
          const StereoEdge ed = g.stereo[e];
          const int col = g.pose_col[ed.pose];
          if (col < 0) continue;
          const double* rt = &cur.pose_rt[ed.pose * 12];
          const double* cm = &g.pose_cam[ed.pose * 5];
          Mat3 R;
#pragma unroll
          for (int q = 0; q < 9; ++q) R.a[q] = rt[q];
          double ev[3], Jl[9], Jp[18], zc;
          stereo_eval(R, v3(rt[9], rt[10], rt[11]), X, cm[0], cm[1], cm[2], cm[3], cm[4], ed.u, ed.v, ed.ur, ev, Jl, Jp, zc);
          double chi2 = 0;
#pragma unroll
          for (int r = 0; r < 3; ++r) chi2 += ev[r] * (ed.info * ev[r]);
          double rho0, rho1;
          huber(chi2, (fl & F_ROBUST) ? ed.huber : 0.0, rho0, rho1);
          const double w = rho1 * ed.info;
          const double* x = &g.xr[col];
#pragma unroll
          for (int r = 0; r < 3; ++r) {
            double tr = 0;
#pragma unroll
            for (int a = 0; a < 6; ++a) tr += Jp[6 * r + a] * x[a];
            tr *= w;
            c0 -= Jl[3 * r] * tr; c1 -= Jl[3 * r + 1] * tr; c2 -= Jl[3 * r + 2] * tr;
          }
        }
      } else {
        const int j = l - g.n_points;
        for (int q = g.pl_slot_off[j] + sub; q < g.pl_slot_off[j + 1]; q += LPP) {
          const int s = g.pl_slots[q];
          const double* w = &g.W[(size_t)(s - g.n_pt_slots) * 18];
          const double* x = &g.xr[g.slot_col[s]];
          for (int a = 0; a < 6; ++a) {
            c0 -= w[a * 3] * x[a]; c1 -= w[a * 3 + 1] * x[a]; c2 -= w[a * 3 + 2] * x[a];
          }
        }
      }
#pragma unroll
      for (int off = 1; off < LPP; off <<= 1) {
        c0 += __shfl_xor(c0, off, 64); c1 += __shfl_xor(c1, off, 64); c2 += __shfl_xor(c2, off, 64);
      }
      c0 += b0; c1 += b1; c2 += b2;
      const double* I = &g.Dinv[(size_t)l * 6];
      xl[0] = I[0] * c0 + I[1] * c1 + I[2] * c2;
      xl[1] = I[1] * c0 + I[3] * c1 + I[4] * c2;
      xl[2] = I[2] * c0 + I[4] * c1 + I[5] * c2;
      if (sub == 0) sc = xl[0] * (lambda * xl[0] + b0) + xl[1] * (lambda * xl[1] + b1) + xl[2] * (lambda * xl[2] + b2);
    }
    if (sub != 0) {
    } else if (l < g.n_points) {
      for (int a = 0; a < 3; ++a) trial.point[3 * l + a] = cur.point[3 * l + a] + xl[a];
    } else {
      const int j = l - g.n_points;
      const PlaneC p = plane_boxplus(load_plane(&cur.plane[4 * j]), xl);
      trial.plane[4 * j] = p.nx; trial.plane[4 * j + 1] = p.ny; trial.plane[4 * j + 2] = p.nz; trial.plane[4 * j + 3] = p.d;
    }
  }
  block_atomic_add(g, sc, ACC_SCALE, s_red);
}
template <bool USE_LDS, int LPP>
__global__ __launch_bounds__(256) void k_backsub(Graph g, State cur, State trial, double lambda) {
  extern __shared__ double smem[];
  __shared__ double s_red[4];
  backsub_body<USE_LDS, LPP>(g, cur, trial, lambda, blockIdx.x, smem, s_red);
}

// Sharded runs exchange only what the factorisation reads: the lower-triangle 64x64 tiles of S (and b), packed
// contiguously for ONE all-reduce (half the bytes of the full column-major matrix).  dir = 0: S -> pack, 1: pack -> S.
__global__ __launch_bounds__(256) void k_pack_lower_tiles(double* __restrict__ S, double* __restrict__ bs, double* __restrict__ pack, int ld, int dir) {
  int b = blockIdx.x, bi = 0;
  const int nb = ld / 64, ntiles = nb * (nb + 1) / 2;
  if (b == ntiles) {  // the right-hand side rides behind the tiles
    for (int i = threadIdx.x; i < ld; i += 256) {
      if (dir == 0) pack[(size_t)ntiles * 4096 + i] = bs[i];
      else if (dir == 2) pack[(size_t)ntiles * 4096 + i] = (double)__double_as_longlong(bs[i]) * kFixInv;
      else bs[i] = pack[(size_t)ntiles * 4096 + i];
    }
    return;
  }
  while ((bi + 1) * (bi + 2) / 2 <= b) ++bi;
  const int bj = b - bi * (bi + 1) / 2;
  double2* tile = reinterpret_cast<double2*>(&S[(size_t)(bj * 64) * ld + bi * 64]);
  double2* dst = reinterpret_cast<double2*>(pack + (size_t)b * 4096);
  for (int i = threadIdx.x; i < 64 * 32; i += 256) {
    const int c = i >> 5, r2 = i & 31;
    if (dir == 0) dst[i] = tile[(size_t)c * (ld / 2) + r2];
    else if (dir == 2) {  // deterministic mode: S / bs are the fixed-point arrays -- converted while they are packed (no conversion launch in front of the collective)
      const double2 v = tile[(size_t)c * (ld / 2) + r2];
      dst[i] = make_double2((double)__double_as_longlong(v.x) * kFixInv, (double)__double_as_longlong(v.y) * kFixInv);
    } else tile[(size_t)c * (ld / 2) + r2] = dst[i];
  }
}

// The whole update of a trial in ONE launch (horizontal fusion; none of the parts reads another's output): cuboid
// back-substitution + update | pose updates | landmark back-substitution + update | and, since the factor in S is dead once
// the backward substitution has run, the zeroing of S (lower-triangle 64x64 tiles) and b for the NEXT reduction.
#ifndef PPO_UPD_LPP
#define PPO_UPD_LPP 4  // lanes per landmark in the update launch
#endif
struct UpdatePlan {
  int nb_red, nb_lm, nb_zero;  // workgroups of the reduced part (cuboids + poses), landmarks, S tiles
  double* S;
  double* bs;
  int ld;
#ifdef PPO_BA_TEST_HOOKS
  // round 6, timing experiment PPO_BA_FUSE_BOUND=1 (results stay right): the first fuse_lin_blocks landmark workgroups ALSO run the landmark side of the
  // linearisation on the trial state behind their back-substitution -- into the linearisation buffer nobody reads (fD / fbl), without the chi2 atomics --,
  // i.e. what a fused "update + landmark-major linearisation" launch would make them do.  The duration of THIS launch next to the duration of the fused
  // linearisation without its landmark part (PPO_BA_LIN_PARTS=15) bounds what that fusion can gain.
  int fuse_lin_blocks;
  double *fD, *fbl;
#endif
};
template <bool USE_LDS>
__global__ __launch_bounds__(256) void k_update_all(Graph g, State cur, State trial, double lambda, UpdatePlan up) {
  extern __shared__ double smem[];
  __shared__ double s_red[4];
  __shared__ double sp[54], st[9], sy[9], sx[9];
  if (lm_dead(g)) return;
  lambda = lm_lambda(g, lambda);
  int b = blockIdx.x;
  if (b < up.nb_red) { update_reduced_body(g, cur, trial, lambda, b, sp, st, sy, sx); return; }
  b -= up.nb_red;
  if (b < up.nb_lm) {
    backsub_body<USE_LDS, PPO_UPD_LPP>(g, cur, trial, lambda, b, smem, s_red);
#ifdef PPO_BA_TEST_HOOKS
    if (b < up.fuse_lin_blocks) {
      __syncthreads();
      Graph g2 = g;
      g2.D = up.fD; g2.bl = up.fbl;
      g2.dbg |= 4;  // (no chi2 atomics: the real linearisation behind this launch owns the sums)
      lin_points_body<USE_LDS>(g2, trial, b, up.fuse_lin_blocks, smem, s_red);
    }
#endif
    return;
  }
  b -= up.nb_lm;
  // zero one 64x64 tile of the lower triangle of S (tile index b -> (bi >= bj)); the last workgroup also clears b
  int bi = 0;
  while ((bi + 1) * (bi + 2) / 2 <= b) ++bi;
  const int bj = b - bi * (bi + 1) / 2;
  if (bi * 64 < up.ld) {
    double2* base = reinterpret_cast<double2*>(&up.S[(size_t)(bj * 64) * up.ld + bi * 64]);
    for (int i = threadIdx.x; i < 64 * 32; i += 256) {
      const int c = i >> 5, r2 = i & 31;
      base[(size_t)c * (up.ld / 2) + r2] = make_double2(0.0, 0.0);
    }
  }
  if (b == up.nb_zero - 1)
    for (int i = threadIdx.x; i < up.ld; i += 256) up.bs[i] = 0.0;
}

// Pose-major static copies of the reprojection edges, expanded ON THE DEVICE from the landmark-major records and the index list the
// host's bucket pass produced (ppo_ba_create): 4 bytes per edge cross the bus instead of a second 48 / 64-byte record.
__global__ __launch_bounds__(256) void k_expand_pose_major(const MonoEdge* __restrict__ mono, const int32_t* __restrict__ mono_src, MonoEdgeP* __restrict__ monoP, int n_mono,
                                                           const StereoEdge* __restrict__ stereo, const int32_t* __restrict__ stereo_src, StereoEdgeP* __restrict__ stereoP, int n_stereo) {
  const int stride = gridDim.x * blockDim.x;
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < n_mono; q += stride) {
    const int k = mono_src[q];
    const MonoEdge m = mono[k];
    monoP[q] = MonoEdgeP{m.u, m.v, m.info, m.huber, m.point, k, m.slot, 0};
  }
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < n_stereo; q += stride) {
    const int k = stereo_src[q];
    const StereoEdge m = stereo[k];
    StereoEdgeP e{};
    e.u = m.u; e.v = m.v; e.ur = m.ur; e.info = m.info; e.huber = m.huber; e.point = m.point; e.eid = k; e.slot = m.slot;
    stereoP[q] = e;
  }
}

__global__ void k_pose_rt(State st, int n_poses) {  // derive rotation matrices after an upload
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_poses) store_pose(&st.pose_qt[7 * i], &st.pose_rt[12 * i], load_pose(&st.pose_qt[7 * i]));
}

}  // namespace ppo
