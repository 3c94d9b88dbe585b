// ppo_pose.hip -- pose-only optimisation (include/ppo_pose.h): Optimizer::PoseOptimization of the reference
// (src/Optimizer.cc:247-459) as ONE kernel launch.
//
// The problem is tiny (one 6-DoF pose, a few hundred to a few thousand unary reprojection edges) and strictly sequential
// on the outside: 4 rounds x <= 10 Levenberg-Marquardt iterations x >= 1 trial, each needing a reduction over all edges.
// Driving that from the host would cost a launch and a round trip per trial (~10 us each against ~1 us of arithmetic), so
// the whole protocol runs on the device: one workgroup per frame (256 or 512 threads) holds the pose, the 6x6 system and the LM
// state redundantly in the registers of every thread, edges are dealt round-robin to the threads, and the only shared
// traffic is the reduction (an LDS transpose for the 28 values of the normal equations, wave shuffles for the single chi2 of
// a trial -- a fixed summation order, so the result is bit-reproducible from run to run).  Independent frames go to different workgroups of the same launch.
//
// Restated per thread, in the reference's order (G2O = Thirdparty/g2o/g2o):
//   residuals / Jacobians   G2O/types/types_six_dof_expmap.h:236-240, 267-271, .cpp:298-338, 367-398
//   quadratic form + Huber  G2O/core/base_unary_edge.hpp:43-72, robust_kernel_impl.cpp:78-91
//   6x6 solve               G2O/solvers/linear_solver_dense.h:65-113 (Cholesky here; same gate: reject when not positive)
//   LM                      G2O/core/optimization_algorithm_levenberg.cpp:61-189
//   outer loop              G2O/core/sparse_optimizer.cpp:354-419
//   rounds + classification src/Optimizer.cc:375-450
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>

#include "../../include/ppo_pose.h"
#include "ppo_math.hpp"
#include "ppo_pose_math.hpp"
#include "ppo_pool.hpp"

namespace ppo {

struct PoseOnlyArgs {
  const double* pose_qt;
  const double* cam;
  const int32_t* off;
  const double* xw;
  const double* obs;
  const double* invs;
  double* out_pose;       // [F][7]
  double* out_chi2;       // [E]
  double* stored;         // [E] chi2 of the stored _error of an active edge (what e->chi2() returns without a recompute)
  uint8_t* level;         // [E]
  int32_t* n_inliers;     // [F]
  ppo_pose_stats* stats;  // [F]
  int rounds, iters, robust_rounds, min_corr, min_edges, max_trials;
  float chi2_mono, chi2_stereo;
  double hub_mono, hub_stereo, tau, lower, upper;
};


// Sum of one value over the workgroup, the same result in every thread; order: xor-butterfly inside a wave, then the waves.
template <int NT>
__device__ __forceinline__ double block_sum1(double x, double* sbuf) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m, 64);
  if ((threadIdx.x & 63) == 0) sbuf[threadIdx.x >> 6] = x;
  __syncthreads();
  double r = sbuf[0];
#pragma unroll
  for (int w = 1; w < NT / 64; ++w) r += sbuf[w];
  __syncthreads();
  return r;
}

// Sum of the 28 values of the normal equations (21 + 6 + chi2) over the workgroup through an LDS transpose: every thread
// stores its 28 partials as one row (stride 29 doubles: the column reads below are conflict-free), thread (seg, v) adds 32
// rows of column v, thread v adds the 8 segment sums, everybody reads the 28 totals back.  ~1 k cycles; the shuffle
// butterfly this replaces (28 values x 6 steps x 2 ds_bpermute) took 8 k of the 11 k cycles of an LM iteration.
constexpr int kAccN = 28, kAccStride = 29;
template <int NT>
__device__ __forceinline__ void block_sum_acc(double (&v)[kAccN], double* sP, double* sQ, double* sR) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int k = 0; k < kAccN; ++k) sP[tid * kAccStride + k] = v[k];
  __syncthreads();
  {
    const int col = tid & 31, seg = tid >> 5;
    if (col < kAccN) {
      double s = 0.0;
#pragma unroll 8
      for (int j = 0; j < 32; ++j) s += sP[(seg * 32 + j) * kAccStride + col];
      sQ[seg * 32 + col] = s;
    }
  }
  __syncthreads();
  if (tid < kAccN) {
    double s = sQ[tid];
#pragma unroll
    for (int g = 1; g < NT / 32; ++g) s += sQ[g * 32 + tid];
    sR[tid] = s;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kAccN; ++k) v[k] = sR[k];
}

__device__ __forceinline__ PoseEdge load_pose_edge(const PoseOnlyArgs& a, int e) {
  PoseEdge E;
  E.X = v3(a.xw[3 * (size_t)e], a.xw[3 * (size_t)e + 1], a.xw[3 * (size_t)e + 2]);
  E.u = a.obs[3 * (size_t)e];
  E.v = a.obs[3 * (size_t)e + 1];
  E.ur = a.obs[3 * (size_t)e + 2];
  E.invs = a.invs[e];
  E.stereo = !(E.ur < 0);  // Optimizer.cc:294 `if(pFrame->mvuRight[i]<0)` is the monocular branch
  return E;
}
// (H + lambda I) x = b with H packed lower (row r, column c <= r at r(r+1)/2 + c); false when a pivot is not positive.
// Every thread walks this chain alone, so it is written for latency: one reciprocal square root per pivot, no division.
__device__ __forceinline__ bool solve6(const double* H, double lambda, const double* b, double* x) {
  double L[21], inv[6];
  bool ok = true;
#pragma unroll
  for (int r = 0; r < 6; ++r) {
#pragma unroll
    for (int c = 0; c <= r; ++c) {
      double s = H[r * (r + 1) / 2 + c] + (r == c ? lambda : 0.0);
#pragma unroll
      for (int k = 0; k < c; ++k) s -= L[r * (r + 1) / 2 + k] * L[c * (c + 1) / 2 + k];
      if (r == c) {
        ok = ok && (s > 0.0);
        inv[r] = rsqrt(s);
        L[r * (r + 1) / 2 + c] = s * inv[r];
      } else {
        L[r * (r + 1) / 2 + c] = s * inv[c];
      }
    }
  }
  if (!ok) return false;
  double y[6];
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    double s = b[r];
#pragma unroll
    for (int k = 0; k < r; ++k) s -= L[r * (r + 1) / 2 + k] * y[k];
    y[r] = s * inv[r];
  }
#pragma unroll
  for (int r = 5; r >= 0; --r) {
    double s = y[r];
#pragma unroll
    for (int k = r + 1; k < 6; ++k) s -= L[k * (k + 1) / 2 + r] * x[k];
    x[r] = s * inv[r];
  }
  return true;
}

// exp(u) * T (VertexSE3Expmap::oplusImpl) on the same latency diet: the rotation of exp(u) directly as a quaternion
// (sin(th/2)/th * omega, cos(th/2)), sin th / cos th from the half angle, one reciprocal for the normalisation.  Equal to
// compose(exp_se3(u), T) up to rounding; below g2o's small-angle threshold (se3quat.h:229-271 switches formula there) the
// shared routine is used so that the branch is reproduced exactly.
__device__ __forceinline__ Pose pose_boxplus_fast(const Pose& T, const double* u) {
  const Vec3 om = v3(u[0], u[1], u[2]), up = v3(u[3], u[4], u[5]);
  const double th2 = dot3(om, om);
  if (th2 < 1e-10) return compose(exp_se3(u), T);
  const double th = sqrt(th2), ith = 1.0 / th;
  double sh, ch;
  sincos(0.5 * th, &sh, &ch);
  const double sn = 2.0 * sh * ch, vers = 2.0 * sh * sh;  // sin th, 1 - cos th
  const double k = sh * ith;
  Quat qe = Quat{k * om.x, k * om.y, k * om.z, ch};       // unit by construction, w >= 0 for th < pi
  if (qe.w < 0) qe = Quat{-qe.x, -qe.y, -qe.z, -qe.w};
  const double ith2 = ith * ith;
  const double bb = vers * ith2, cc = (th - sn) * ith2 * ith;  // V = I + b W + c W^2
  const Vec3 wu = cross3(om, up);
  const Vec3 wwu = cross3(om, wu);
  Pose r;
  r.t = up + wu * bb + wwu * cc + qrot(qe, T.t);
  Quat q = qmul(qe, T.q);
  if (q.w < 0) q = Quat{-q.x, -q.y, -q.z, -q.w};
  const double in = 1.0 / sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  r.q = Quat{q.x * in, q.y * in, q.z * in, q.w * in};
  return r;
}

// NT threads per frame: 256 suit the usual few hundred observations (fewer waves to synchronise), 512 pay off above ~700
template <int NT>
__global__ __launch_bounds__(NT) void k_pose_only(PoseOnlyArgs a) {
  constexpr int kPoseThreads = NT;
  __shared__ double sbuf[NT / 64];
  __shared__ double sP[NT * kAccStride], sQ[(NT / 32) * 32], sR[32];
  const int f = blockIdx.x, tid = threadIdx.x;
  const int e0 = a.off[f], e1 = a.off[f + 1], n = e1 - e0;
  double cam[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) cam[i] = a.cam[5 * f + i];
  const double* q0 = &a.pose_qt[7 * f];
  Pose T0;
  T0.q = canon(Quat{q0[0], q0[1], q0[2], q0[3]});  // SE3Quat(R, t) normalises (se3quat.h:58-60)
  T0.t = v3(q0[4], q0[5], q0[6]);
  for (int e = e0 + tid; e < e1; e += kPoseThreads) {  // Optimizer.cc:297, 331: every correspondence starts as an inlier
    a.level[e] = 0;
    a.out_chi2[e] = 0.0;
    a.stored[e] = 0.0;
  }
  ppo_pose_stats st;
  st.rounds_run = 0; st.lm_iterations = 0; st.lm_trials = 0; st.n_bad = 0;
  st.chi2_initial = 0; st.chi2_final = 0; st.lambda_final = 0;
  if (n < a.min_corr) {  // Optimizer.cc:372-373: return 0, the frame keeps its pose
    if (tid == 0) {
      for (int i = 0; i < 7; ++i) a.out_pose[7 * f + i] = q0[i];
      a.n_inliers[f] = 0;
      a.stats[f] = st;
    }
    return;
  }
  Pose T = T0;
  double lambda = 0.0, ni = 2.0;
  double x[6] = {0, 0, 0, 0, 0, 0};
  int nBad = 0;
  for (int it = 0; it < a.rounds; ++it) {
    T = T0;                                   // :385
    const bool robust = it < a.robust_rounds;  // `if(it==2) e->setRobustKernel(0)` takes effect from the next round on
    double cnt = 0.0;
    for (int e = e0 + tid; e < e1; e += kPoseThreads) cnt += a.level[e] == 0 ? 1.0 : 0.0;
    cnt = block_sum1<NT>(cnt, sbuf);
    if (cnt > 0.0) {  // initializeOptimization(0) found active edges; otherwise optimize() returns at once (sparse_optimizer.cpp:356-360)
      int lm_bad = 0;
      bool ok = true;
      for (int i = 0; i < a.iters && ok; ++i) {
        // ---- computeActiveErrors + activeRobustChi2 + buildSystem at T (optimization_algorithm_levenberg.cpp:76-84) ----
        double acc[28];
#pragma unroll
        for (int k = 0; k < 28; ++k) acc[k] = 0.0;
        for (int e = e0 + tid; e < e1; e += kPoseThreads) {
          if (a.level[e]) continue;
          const PoseEdge E = load_pose_edge(a, e);
          a.stored[e] = pose_obs_accumulate(T, cam, E, robust ? (E.stereo ? a.hub_stereo : a.hub_mono) : 0.0, acc);
        }
        block_sum_acc<NT>(acc, sP, sQ, sR);
        double currentChi = acc[27];
        const double iniChi = currentChi;
        double tempChi = currentChi;
        if (it == 0 && i == 0) st.chi2_initial = currentChi;
        if (i == 0) {  // computeLambdaInit, :166-180
          double maxDiagonal = 0.0;
#pragma unroll
          for (int c = 0; c < 6; ++c) maxDiagonal = fmax(fabs(acc[c * (c + 1) / 2 + c]), maxDiagonal);
          lambda = a.tau * maxDiagonal;
          ni = 2.0;
          lm_bad = 0;
        }
        double rho = 0.0;
        int qmax = 0;
        do {
          const bool ok2 = solve6(acc, lambda, &acc[21], x);  // on failure x keeps the previous solution, as g2o's _x does
          const Pose Tn = pose_boxplus_fast(T, x);             // VertexSE3Expmap::oplusImpl
          double c1 = 0.0;
          for (int e = e0 + tid; e < e1; e += kPoseThreads) {
            if (a.level[e]) continue;
            const PoseEdge E = load_pose_edge(a, e);
            double er[3];
            Vec3 pc;
            const double chi2 = pose_edge_chi2(Tn, cam, E, er, pc);
            a.stored[e] = chi2;
            double rho0, rho1;
            huber(chi2, robust ? (E.stereo ? a.hub_stereo : a.hub_mono) : 0.0, rho0, rho1);
            c1 += rho0;
          }
          c1 = block_sum1<NT>(c1, sbuf);
          tempChi = ok2 ? c1 : 1.7976931348623157e308;
          rho = currentChi - tempChi;
          double scale = 0.0;  // computeScale, :182-189
#pragma unroll
          for (int c = 0; c < 6; ++c) scale += x[c] * (lambda * x[c] + acc[21 + c]);
          scale += 1e-3;
          rho /= scale;
          if (rho > 0 && isfinite(tempChi)) {
            const double tr = 2 * rho - 1;
            double alpha = 1. - tr * tr * tr;  // pow(2 rho - 1, 3)
            alpha = fmin(alpha, a.upper);
            const double scaleFactor = fmax(a.lower, alpha);
            lambda *= scaleFactor;
            ni = 2.0;
            currentChi = tempChi;
            T = Tn;
          } else {
            lambda *= ni;
            ni *= 2.0;
          }
          qmax++;
          st.lm_trials++;
        } while (rho < 0 && qmax < a.max_trials);
        st.lm_iterations++;
        st.chi2_final = currentChi;
        if (qmax == a.max_trials || rho == 0) { ok = false; continue; }  // Terminate
        if ((iniChi - currentChi) * 1e3 < iniChi) lm_bad++;
        else lm_bad = 0;
        if (lm_bad >= 3) ok = false;
      }
    }
    // ---- classification, Optimizer.cc:389-446 ----
    double bad = 0.0;
    for (int e = e0 + tid; e < e1; e += kPoseThreads) {
      const PoseEdge E = load_pose_edge(a, e);
      double c;
      if (a.level[e]) {  // `if(pFrame->mvbOutlier[idx]) e->computeError()`: it did not take part in this round
        double er[3];
        Vec3 pc;
        c = pose_edge_chi2(T, cam, E, er, pc);
      } else {
        c = a.stored[e];
      }
      const float chi2 = (float)c;
      const bool out = chi2 > (E.stereo ? a.chi2_stereo : a.chi2_mono);
      a.level[e] = out ? 1 : 0;
      a.out_chi2[e] = c;
      bad += out ? 1.0 : 0.0;
    }
    bad = block_sum1<NT>(bad, sbuf);
    nBad = (int)bad;
    st.rounds_run = it + 1;
    if (n < a.min_edges) break;  // :448-449
  }
  if (tid == 0) {
    double* o = &a.out_pose[7 * f];
    o[0] = T.q.x; o[1] = T.q.y; o[2] = T.q.z; o[3] = T.q.w; o[4] = T.t.x; o[5] = T.t.y; o[6] = T.t.z;
    a.n_inliers[f] = n - nBad;  // :458
    st.n_bad = nBad;
    st.lambda_final = lambda;
    a.stats[f] = st;
  }
}

}  // namespace ppo

using namespace ppo;

extern "C" {

void ppo_pose_default_options(ppo_pose_options* o) {
  if (!o) return;
  std::memset(o, 0, sizeof(*o));
  o->abi_version = PPO_POSE_ABI_VERSION;
  o->device_id = 0;
  o->rounds = 4;                 // Optimizer.cc:377-382
  o->iters_per_round = 10;
  o->robust_rounds = 3;          // :415, 444
  o->min_correspondences = 3;    // :372
  o->min_edges = 10;             // :448
  o->chi2_mono = 5.991f;
  o->chi2_stereo = 7.815f;
  o->huber_mono = (float)std::sqrt(5.991);    // `const float deltaMono = sqrt(5.991)`, :281
  o->huber_stereo = (float)std::sqrt(7.815);
  o->lm_tau = 1e-5;              // optimization_algorithm_levenberg.cpp:47-51
  o->lm_max_trials = 10;
  o->lm_lower = 1. / 3.;
  o->lm_upper = 2. / 3.;
}

int ppo_pose_optimize(const ppo_pose_problem* p, const ppo_pose_options* o, ppo_pose_result* r) try {
  if (!p || !o || !r || !r->pose_qt) return PPO_BA_ERR_ARG;
  if (o->abi_version != PPO_POSE_ABI_VERSION) return PPO_BA_ERR_ARG;
  if (p->n_frames < 0 || o->rounds < 0 || o->rounds > PPO_POSE_MAX_ROUNDS || o->iters_per_round < 0 || o->lm_max_trials < 1) return PPO_BA_ERR_ARG;
  const int F = p->n_frames;
  if (F == 0) return PPO_BA_OK;
  if (!p->pose_qt || !p->cam || !p->edge_offsets) return PPO_BA_ERR_ARG;
  if (p->edge_offsets[0] != 0) return PPO_BA_ERR_ARG;
  for (int f = 0; f < F; ++f)
    if (p->edge_offsets[f + 1] < p->edge_offsets[f]) return PPO_BA_ERR_ARG;
  const size_t E = (size_t)p->edge_offsets[F];
  if (E && (!p->xw || !p->obs || !p->inv_sigma2)) return PPO_BA_ERR_ARG;
  auto finite = [](const double* v, size_t n) {
    for (size_t i = 0; i < n; ++i)
      if (!std::isfinite(v[i])) return false;
    return true;
  };
  if (!finite(p->pose_qt, (size_t)7 * F) || !finite(p->cam, (size_t)5 * F) || !finite(p->xw, 3 * E) || !finite(p->obs, 3 * E) || !finite(p->inv_sigma2, E))
    return PPO_BA_ERR_NONFINITE;

  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    std::fprintf(stderr, "[ppo_ba] no HIP device: libppo_ba has no CPU path\n");
    return PPO_BA_ERR_HIP;
  }
  if (o->device_id < 0 || o->device_id >= ndev) return PPO_BA_ERR_ARG;
  if (hipSetDevice(o->device_id) != hipSuccess) return PPO_BA_ERR_HIP;
  DevCtx ctx;
  if (!g_ctx_pool.get(o->device_id, &ctx)) return PPO_BA_ERR_HIP;

  // one packed block each way: [pose 7F | cam 5F | xw 3E | obs 3E | invs E | offsets F+1 (int32)] in,
  // [pose 7F | chi2 E | stats F | n_inliers F (int32) | level E (bytes)] out; `stored` is device scratch
  auto up8 = [](size_t b) { return (b + 7) & ~size_t(7); };
  const size_t in_d = (size_t)12 * F + 7 * E;
  const size_t in_bytes = in_d * 8 + up8((size_t)(F + 1) * 4);
  const size_t out_bytes = ((size_t)7 * F + E) * 8 + up8((size_t)F * sizeof(ppo_pose_stats)) + up8((size_t)F * 4) + up8(E);
  Stage stage;
  int rc = PPO_BA_OK;
  char* h_in = static_cast<char*>(stage.take(in_bytes));
  char* h_out = static_cast<char*>(stage.take(out_bytes));
  size_t cap_in = 0, cap_out = 0, cap_st = 0;
  char* d_in = static_cast<char*>(g_pool.get(in_bytes, o->device_id, &cap_in));
  char* d_out = static_cast<char*>(g_pool.get(out_bytes, o->device_id, &cap_out));
  double* d_stored = static_cast<double*>(g_pool.get(std::max<size_t>(8, E * 8), o->device_id, &cap_st));
  if (!h_in || !h_out || !d_in || !d_out || !d_stored) rc = PPO_BA_ERR_NOMEM;
  if (rc == PPO_BA_OK) {
    double* hd = reinterpret_cast<double*>(h_in);
    std::memcpy(hd, p->pose_qt, (size_t)7 * F * 8);
    std::memcpy(hd + 7 * F, p->cam, (size_t)5 * F * 8);
    if (E) {
      std::memcpy(hd + 12 * F, p->xw, 3 * E * 8);
      std::memcpy(hd + 12 * F + 3 * E, p->obs, 3 * E * 8);
      std::memcpy(hd + 12 * F + 6 * E, p->inv_sigma2, E * 8);
    }
    std::memcpy(hd + in_d, p->edge_offsets, (size_t)(F + 1) * 4);
    PoseOnlyArgs a;
    const double* dd = reinterpret_cast<const double*>(d_in);
    a.pose_qt = dd; a.cam = dd + 7 * F; a.xw = dd + 12 * F; a.obs = dd + 12 * F + 3 * E; a.invs = dd + 12 * F + 6 * E;
    a.off = reinterpret_cast<const int32_t*>(dd + in_d);
    char* q = d_out;
    a.out_pose = reinterpret_cast<double*>(q); q += (size_t)7 * F * 8;
    a.out_chi2 = reinterpret_cast<double*>(q); q += E * 8;
    a.stats = reinterpret_cast<ppo_pose_stats*>(q); q += up8((size_t)F * sizeof(ppo_pose_stats));
    a.n_inliers = reinterpret_cast<int32_t*>(q); q += up8((size_t)F * 4);
    a.level = reinterpret_cast<uint8_t*>(q);
    a.stored = d_stored;
    a.rounds = o->rounds; a.iters = o->iters_per_round; a.robust_rounds = o->robust_rounds; a.min_corr = o->min_correspondences;
    a.min_edges = o->min_edges; a.max_trials = o->lm_max_trials;
    a.chi2_mono = o->chi2_mono; a.chi2_stereo = o->chi2_stereo;
    a.hub_mono = (double)o->huber_mono; a.hub_stereo = (double)o->huber_stereo;
    a.tau = o->lm_tau; a.lower = o->lm_lower; a.upper = o->lm_upper;
    bool ok = hipMemcpyAsync(d_in, h_in, in_bytes, hipMemcpyHostToDevice, ctx.stream) == hipSuccess;
    if (ok) {
      int max_n = 0;
      for (int f = 0; f < F; ++f) max_n = std::max(max_n, p->edge_offsets[f + 1] - p->edge_offsets[f]);
      if (max_n > 700) hipLaunchKernelGGL(k_pose_only<512>, dim3(F), dim3(512), 0, ctx.stream, a);
      else hipLaunchKernelGGL(k_pose_only<256>, dim3(F), dim3(256), 0, ctx.stream, a);
      ok = hipGetLastError() == hipSuccess;
    }
    ok = ok && hipMemcpyAsync(h_out, d_out, out_bytes, hipMemcpyDeviceToHost, ctx.stream) == hipSuccess;
    ok = ok && hipStreamSynchronize(ctx.stream) == hipSuccess;
    if (!ok) {
      std::fprintf(stderr, "[ppo_ba] ppo_pose_optimize: %s\n", hipGetErrorString(hipGetLastError()));
      rc = PPO_BA_ERR_HIP;
      (void)hipStreamSynchronize(ctx.stream);
    } else {
      const char* s = h_out;
      std::memcpy(r->pose_qt, s, (size_t)7 * F * 8); s += (size_t)7 * F * 8;
      if (r->chi2 && E) std::memcpy(r->chi2, s, E * 8);
      s += E * 8;
      if (r->stats) std::memcpy(r->stats, s, (size_t)F * sizeof(ppo_pose_stats));
      s += up8((size_t)F * sizeof(ppo_pose_stats));
      if (r->n_inliers) std::memcpy(r->n_inliers, s, (size_t)F * 4);
      s += up8((size_t)F * 4);
      if (r->outlier && E) std::memcpy(r->outlier, s, E);
    }
  }
  if (d_in) g_pool.put(d_in, cap_in, o->device_id);
  if (d_out) g_pool.put(d_out, cap_out, o->device_id);
  if (d_stored) g_pool.put(d_stored, cap_st, o->device_id);
  stage.release();
  g_ctx_pool.put(ctx);
  return rc;
} catch (const std::bad_alloc&) {
  return PPO_BA_ERR_NOMEM;
}

}  // extern "C"
