// ppo_oracle.cpp -- TEST INFRASTRUCTURE ONLY (see oracle/README.md): single-threaded CPU restatement of
// the reference's local bundle adjustment, used as parity checker and as bench.py's cpu_baseline leg.
// Nothing in the product path (libppo_ba.so, the host adapter) links, includes or calls this file.
//
// Parity status: the reference cannot be compiled here (Eigen3/OpenCV/PCL/Pangolin absent) and ships no tests for this path,
// so this restatement is pinned by golden vectors produced by the reference's OWN prebuilt binaries, run in the build container:
//   Thirdparty/g2o/lib/libg2o.so  (oracle/pin_g2o_prebuilt.py, pin_g2o_lm.py, pin_g2o_pose.py)
//     SE3 exp, pose oplus, Huber, projections, the unary OnlyPose edges, the two binary reprojection edges through computeError /
//     linearizeOplus / numeric linearizeOplusXi,Xj / constructQuadraticForm -- BIT FOR BIT; the LM loop and the PoseOptimization protocol;
//   lib/libORB_SLAM2.so  (oracle/orbslam_loader.py + pin_orbslam_prebuilt.py, pin_orbslam_lm.py; round 4)
//     Plane3D / VertexPlane, VertexCuboid, exptwist_norollpitch and all eight custom edge classes (residual, chi2, numeric Jacobians,
//     quadratic form), and the complete 5 + classify + 10 optimisation of eleven windows with the reference's own BlockSolverX (Schur
//     complement) + LinearSolverDense (Eigen LDLT): same trial schedule and outlier levels, lambda bit-identical through pass 1 and the
//     first reduced solve to 6e-13 on points-only windows (tests/test_orbslam_pin.py, tests/test_orbslam_lm_pin.py);
// and by analytic known-answer tests.  Nothing on the path is unpinned (oracle/README.md).
//
// What is restated, in the reference's own order (paths relative to the reference checkout, G2O =
// Thirdparty/g2o/g2o):
//   graph activation / index mapping   G2O/core/sparse_optimizer.cpp:166-267
//   structure                          G2O/core/block_solver.hpp:143-295
//   residuals                          types_six_dof_expmap.{h,cpp}, include/G2O_Plane3D.h, src/g2o_cuboid.cc
//   numeric Jacobians (delta 1e-9)     G2O/core/base_binary_edge.hpp:217-320, base_unary_edge.hpp:82-123
//   quadratic form + Huber             G2O/core/base_binary_edge.hpp:55-120, base_unary_edge.hpp:43-72,
//                                      base_edge.h:96-102, robust_kernel_impl.cpp:78-91
//   Schur complement + back-subst.     G2O/core/block_solver.hpp:354-486
//   dense LDLT with positivity gate    G2O/solvers/linear_solver_dense.h:65-113 (Eigen::LDLT restated)
//   Levenberg-Marquardt                G2O/core/optimization_algorithm_levenberg.cpp:61-189
//   outer loop                         G2O/core/sparse_optimizer.cpp:354-419
//   two-pass protocol + classification src/Optimizer.cc:2723-2838
//   pose-only optimisation (row N4)    src/Optimizer.cc:247-459 on the same engine, with the unary OnlyPose edges of
//                                      types_six_dof_expmap.{h:227-290,cpp:298-398} (pinned against the prebuilt libg2o.so)
#include "../include/ppo_ba.h"
#include "../include/ppo_pose.h"
#include "oracle_math.hpp"

#ifdef _OPENMP
#include <omp.h>
#endif
#include <algorithm>
#include <chrono>
#include <cfloat>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <vector>

using namespace orc;

namespace {

enum VType { V_POSE = 0, V_POINT = 1, V_PLANE = 2, V_CUBOID = 3, V_NONE = -1 };
enum EType { E_PLANE = 0, E_MONO, E_STEREO, E_BBOX, E_CORNER, E_PTCUB, E_CUBPL, E_CUB3D, E_NTYPES };
static const int kVDim[4] = {6, 3, 3, 9};
static const int kMaxDim = 16;

struct EdgeRef {
  int type, idx;
};

struct Oracle {
  ppo_ba_options opt;
  // ---- problem copy ----
  int n_poses = 0, n_points = 0, n_planes = 0, n_cuboids = 0;
  std::vector<SE3> pose0;
  std::vector<uint8_t> pose_fixed;
  std::vector<double> pose_cam;
  std::vector<V3> point0;
  std::vector<uint8_t> point_fixed;
  std::vector<Plane> plane0;
  std::vector<Cuboid> cuboid0;
  std::vector<uint8_t> cuboid_flags;
  std::vector<double> cuboid_fixedscale;
  int n_edges[E_NTYPES] = {0};
  std::vector<int> e_v0[E_NTYPES], e_v1[E_NTYPES];
  std::vector<double> e_meas[E_NTYPES], e_info[E_NTYPES], e_huber[E_NTYPES];
  std::vector<uint8_t> pl_kind;
  std::vector<Plane> pl_meas_plane;
  std::vector<double> bbox_K, corner_K;
  std::vector<Cuboid> cub3d_meas;  // EdgeSE3Cuboid measurement (camera frame)
  std::vector<int> ptcub_off;
  std::vector<double> ptcub_pts, ptcub_margin, ptcub_prior;
  // ---- state ----
  std::vector<SE3> pose;
  std::vector<V3> point;
  std::vector<Plane> plane;
  std::vector<Cuboid> cuboid;
  std::vector<uint8_t> level[E_NTYPES], robust[E_NTYPES];
  std::vector<double> err[E_NTYPES];  // stored _error, kMaxDim per edge
  // ---- active sets / index mapping (rebuilt per pass) ----
  std::vector<EdgeRef> active;
  std::vector<int> hidx[4];   // per vertex: -1 or block index (reduced for pose/cuboid, landmark index else)
  std::vector<int> col0[4];   // column of the block inside its partition
  int n_red_blocks = 0, n_red = 0, n_lm = 0;
  std::vector<int> red_col0, red_dim;  // per reduced block
  std::vector<int> lm_type, lm_vidx;   // per landmark block
  // ---- linear system ----
  std::vector<double> Hpp;  // n_red x n_red, column-major, upper blocks filled
  std::vector<double> Hll;  // n_lm x 9 (row-major 3x3)
  std::vector<double> b;    // n_red + 3 n_lm
  std::vector<int> hpl_start;                 // per landmark: first block slot
  std::vector<int> hpl_rb;                    // per slot: reduced block id (sorted inside a landmark)
  std::vector<double> hpl;                    // per slot: 6x3 row-major (pose rows)
  std::vector<int> edge_slot[E_NTYPES];       // per landmark-pose edge: slot (or -1)
  std::vector<double> S, bs, x, coeff, Dinv;  // Schur workspace; x is n_red + 3 n_lm
  std::vector<double> diag_backup_p, diag_backup_l;
  bool do_schur = false;
  int variant = 0;  // test aid (oracle_ba_set_variant): 1 = edges and landmarks summed in REVERSE order (nothing else changes);
                    // 2 = the long sums (pose blocks of Hpp, b, the Schur complement S and its right-hand side) accumulated with a
                    // compensation term each (Neumaier's two-sum: the result a wider accumulator would give, rounded once at the end)
  std::vector<double> Hpp_c, b_c, S_c, coeff_c;  // compensation terms of variant 2
#ifdef _OPENMP
  std::vector<std::vector<double>> omp_S, omp_c, omp_Hd, omp_bp;  // per-thread accumulators of the OpenMP build, kept across calls: eight fresh
                                                                    // 21 MB vectors per solve are eight threads page-faulting at once
#endif
  bool only_pose = false;  // the unary Edge*SE3ProjectXYZOnlyPose flavour of the reprojection edges (pose-only optimisation)
  // ---- LM ----
  double lambda = -1, ni = 2;
  int nBad = 0;
  // ---- bookkeeping ----
  ppo_ba_stats stats;
  bool ran = false;
};

int edge_dim(const Oracle& o, int type, int idx) {
  switch (type) {
    case E_PLANE: return o.pl_kind[idx] == 0 ? 3 : 2;
    case E_MONO: return 2;
    case E_STEREO: return 3;
    case E_BBOX: return 4;
    case E_CORNER: return 16;
    case E_PTCUB: return 3;
    case E_CUBPL: return 3;
    case E_CUB3D: return 9;
  }
  return 0;
}
// vertex slots, SURVEY.md appendix A "Edge vertex slots" (Optimizer.cc:2237-2238,2373-2374,2445-2446,2644,2691-2692)
void edge_vertices(const Oracle& o, int type, int idx, int& t0, int& i0, int& t1, int& i1) {
  i0 = o.e_v0[type][idx];
  i1 = o.e_v1[type].empty() ? -1 : o.e_v1[type][idx];
  t0 = V_POSE;
  t1 = V_NONE;
  switch (type) {
    case E_PLANE: t0 = V_PLANE; t1 = V_POSE; break;
    case E_MONO: case E_STEREO: t0 = V_POINT; t1 = V_POSE; break;
    case E_BBOX: case E_CORNER: case E_CUB3D: t0 = V_POSE; t1 = V_CUBOID; break;
    case E_PTCUB: t0 = V_CUBOID; t1 = V_NONE; i1 = -1; break;
    case E_CUBPL: t0 = V_CUBOID; t1 = V_PLANE; break;
  }
}
bool vertex_fixed(const Oracle& o, int t, int i) {
  if (t == V_POSE) return o.pose_fixed[i] != 0;
  if (t == V_POINT) return o.point_fixed[i] != 0;
  return false;
}
const void* vertex_ptr(const Oracle& o, int t, int i) {
  switch (t) {
    case V_POSE: return &o.pose[i];
    case V_POINT: return &o.point[i];
    case V_PLANE: return &o.plane[i];
    case V_CUBOID: return &o.cuboid[i];
  }
  return nullptr;
}

// computeError() of every edge class; a/b are the estimates of vertex slot 0 / 1.
void compute_error(const Oracle& o, int type, int idx, const void* a, const void* bptr, double* e) {
  switch (type) {
    case E_MONO: {  // types_six_dof_expmap.h:174-179, .cpp:173-179
      const V3& X = *static_cast<const V3*>(a);
      const SE3& T = *static_cast<const SE3*>(bptr);
      const double* cam = &o.pose_cam[5 * o.e_v1[type][idx]];
      const double* obs = &o.e_meas[type][2 * idx];
      const V3 pc = se3_map(T, X);
      const double px = pc[0] / pc[2], py = pc[1] / pc[2];
      e[0] = obs[0] - (px * cam[0] + cam[2]);
      e[1] = obs[1] - (py * cam[1] + cam[3]);
      break;
    }
    case E_STEREO: {  // types_six_dof_expmap.h:206-211, .cpp:182-189 (invz and bf are float there)
      const V3& X = *static_cast<const V3*>(a);
      const SE3& T = *static_cast<const SE3*>(bptr);
      const double* cam = &o.pose_cam[5 * o.e_v1[type][idx]];
      const double* obs = &o.e_meas[type][3 * idx];
      const V3 pc = se3_map(T, X);
      const float bf = (float)cam[4];
      const float invz = (float)(1.0 / pc[2]);  // 1.0f / double: a double division rounded to float (divsd + cvtsd2ss in the prebuilt libg2o.so)
      const double r0 = pc[0] * invz * cam[0] + cam[2];
      const double r1 = pc[1] * invz * cam[1] + cam[3];
      // EdgeStereoSE3ProjectXYZOnlyPose keeps bf as a double member (types_six_dof_expmap.h:288, .cpp:331-338)
      const double r2 = o.only_pose ? r0 - cam[4] * invz : r0 - (double)(bf * invz);
      e[0] = obs[0] - r0;
      e[1] = obs[1] - r1;
      e[2] = obs[2] - r2;
      break;
    }
    case E_PLANE: {  // G2O_Plane3D.h:181-192, 220-231, 279-290
      const Plane& pl = *static_cast<const Plane*>(a);
      const SE3& T = *static_cast<const SE3*>(bptr);
      const Plane local = plane_transform(T, pl);
      const Plane& m = o.pl_meas_plane[idx];
      if (o.pl_kind[idx] == 0) plane_ominus(local, m, e);
      else if (o.pl_kind[idx] == 1) plane_ominus_ver(local, m, e);
      else plane_ominus_par(local, m, e);
      break;
    }
    case E_BBOX: {  // src/g2o_cuboid.cc:70-81
      const SE3& T = *static_cast<const SE3*>(a);
      const Cuboid& c = *static_cast<const Cuboid*>(bptr);
      double bb[4];
      cuboid_project_bbox(c, T, &o.bbox_K[9 * idx], bb);
      for (int i = 0; i < 4; ++i) e[i] = bb[i] - o.e_meas[type][4 * idx + i];
      break;
    }
    case E_CORNER: {  // src/g2o_cuboid.cc:103-120
      const SE3& T = *static_cast<const SE3*>(a);
      const Cuboid& c = *static_cast<const Cuboid*>(bptr);
      double uv[8][2];
      cuboid_project_corners(c, T, &o.corner_K[9 * idx], uv);
      for (int k = 0; k < 8; ++k) {
        e[2 * k] = uv[k][0] - o.e_meas[type][16 * idx + 2 * k];
        e[2 * k + 1] = uv[k][1] - o.e_meas[type][16 * idx + 2 * k + 1];
      }
      break;
    }
    case E_PTCUB: {  // src/g2o_cuboid.cc:132-160
      const Cuboid& c = *static_cast<const Cuboid*>(a);
      V3 acc = mk(0, 0, 0);
      const int p0 = o.ptcub_off[idx], p1 = o.ptcub_off[idx + 1];
      for (int p = p0; p < p1; ++p) {
        const V3 pe = cuboid_point_boundary_error(c, mk(o.ptcub_pts[3 * p], o.ptcub_pts[3 * p + 1], o.ptcub_pts[3 * p + 2]),
                                                  o.ptcub_margin[idx]);
        for (int i = 0; i < 3; ++i) acc[i] += std::fabs(pe[i]);
      }
      if (p1 > p0)
        for (int i = 0; i < 3; ++i) acc[i] = acc[i] / (double)(p1 - p0);
      for (int i = 0; i < 3; ++i) acc[i] = acc[i] / c.scale[i];
      double prior_weight = 0.2;
      V3 prior = c.scale;
      const double* ph = &o.ptcub_prior[3 * idx];
      if (ph[0] > 0) {
        prior_weight = 50.0;
        for (int i = 0; i < 3; ++i) prior[i] = std::fabs((c.scale[i] - ph[i]) / ph[i]);
      }
      for (int i = 0; i < 3; ++i) e[i] = 1.0 * acc[i] + prior_weight * prior[i];
      break;
    }
    case E_CUBPL: {  // G2O_Plane3D.h:470-473: _error = _measurement
      for (int i = 0; i < 3; ++i) e[i] = o.e_meas[type][3 * idx + i];
      break;
    }
    case E_CUB3D: {  // EdgeSE3Cuboid::computeError, include/g2o_cuboid.h:332-341 (Optimizer::LocalBACameraPointCuboids2D, OPT:1767-1800)
      const SE3& T = *static_cast<const SE3*>(a);
      const Cuboid& c = *static_cast<const Cuboid*>(bptr);
      cuboid3d_error(T, c, o.cub3d_meas[idx], e);
      break;
    }
  }
}

// oplus on a copy of one vertex estimate (push / oplus / ... / pop of base_vertex.h:96-97)
struct VTemp {
  SE3 pose;
  V3 point;
  Plane plane;
  Cuboid cuboid;
};
const void* oplus_copy(const Oracle& o, int t, int i, const double* upd, VTemp& tmp) {
  switch (t) {
    case V_POSE: tmp.pose = pose_oplus(o.pose[i], upd); return &tmp.pose;
    case V_POINT: tmp.point = add(o.point[i], mk(upd[0], upd[1], upd[2])); return &tmp.point;  // types_sba.h:52-56
    case V_PLANE: tmp.plane = plane_oplus(o.plane[i], upd); return &tmp.plane;
    case V_CUBOID:
      tmp.cuboid = cuboid_oplus(o.cuboid[i], upd, o.cuboid_flags[i], &o.cuboid_fixedscale[3 * i]);
      return &tmp.cuboid;
  }
  return nullptr;
}

void linearize_numeric(const Oracle& o, int type, int idx, double* J0, double* J1);

// linearizeOplus: analytic for the two reprojection edges, central differences otherwise.
// J0 is dim x d0 (row-major), J1 is dim x d1.
void linearize(const Oracle& o, int type, int idx, double* J0, double* J1) {
  int t0, i0, t1, i1;
  edge_vertices(o, type, idx, t0, i0, t1, i1);
  if (type == E_MONO || type == E_STEREO) {
    // types_six_dof_expmap.cpp:135-171 and :220-266
    const V3& X = o.point[i0];
    const SE3& T = o.pose[i1];
    const double* cam = &o.pose_cam[5 * i1];
    const double fx = cam[0], fy = cam[1], bf = cam[4];
    const V3 pc = se3_map(T, X);
    const M3 R = quat_to_matrix(T.r);
    const double x = pc[0], y = pc[1], z = pc[2], z_2 = z * z;
    if (type == E_MONO) {
      double tmp[2][3] = {{fx, 0, -x / z * fx}, {0, fy, -y / z * fy}};
      for (int r = 0; r < 2; ++r)
        for (int c = 0; c < 3; ++c) {
          const double s = -1. / z;
          J0[r * 3 + c] = (s * tmp[r][0]) * R.m[0][c] + (s * tmp[r][1]) * R.m[1][c] + (s * tmp[r][2]) * R.m[2][c];
        }
    } else {
      for (int c = 0; c < 3; ++c) {
        J0[0 * 3 + c] = -fx * R.m[0][c] / z + fx * x * R.m[2][c] / z_2;
        J0[1 * 3 + c] = -fy * R.m[1][c] / z + fy * y * R.m[2][c] / z_2;
        J0[2 * 3 + c] = J0[0 * 3 + c] - bf * R.m[2][c] / z_2;
      }
    }
    if (o.only_pose) {  // types_six_dof_expmap.cpp:298-320, 367-398: the same matrix written with 1/z and 1/z^2
      const double invz = 1.0 / z, invz_2 = invz * invz;
      J1[0] = x * y * invz_2 * fx;
      J1[1] = -(1 + (x * x * invz_2)) * fx;
      J1[2] = y * invz * fx;
      J1[3] = -invz * fx;
      J1[4] = 0;
      J1[5] = x * invz_2 * fx;
      J1[6] = (1 + y * y * invz_2) * fy;
      J1[7] = -x * y * invz_2 * fy;
      J1[8] = -x * invz * fy;
      J1[9] = 0;
      J1[10] = -invz * fy;
      J1[11] = y * invz_2 * fy;
      if (type == E_STEREO) {
        J1[12] = J1[0] - bf * y * invz_2;
        J1[13] = J1[1] + bf * x * invz_2;
        J1[14] = J1[2];
        J1[15] = J1[3];
        J1[16] = 0;
        J1[17] = J1[5] - bf * invz_2;
      }
      return;
    }
    J1[0] = x * y / z_2 * fx;
    J1[1] = -(1 + (x * x / z_2)) * fx;
    J1[2] = y / z * fx;
    J1[3] = -1. / z * fx;
    J1[4] = 0;
    J1[5] = x / z_2 * fx;
    J1[6] = (1 + y * y / z_2) * fy;
    J1[7] = -x * y / z_2 * fy;
    J1[8] = -x / z * fy;
    J1[9] = 0;
    J1[10] = -1. / z * fy;
    J1[11] = y / z_2 * fy;
    if (type == E_STEREO) {
      J1[12] = J1[0] - bf * y / z_2;
      J1[13] = J1[1] + bf * x / z_2;
      J1[14] = J1[2];
      J1[15] = J1[3];
      J1[16] = 0;
      J1[17] = J1[5] - bf / z_2;
    }
    return;
  }
  linearize_numeric(o, type, idx, J0, J1);
}

// BaseBinaryEdge::linearizeOplusXi / Xj, base_binary_edge.hpp:217-320 (unary: base_unary_edge.hpp:82-123): central differences
// with delta = 1e-9 through the vertex oplus, one column at a time, the other vertex untouched; nothing is written for a
// fixed vertex (:223-226, :275-278).  Usable on every edge class (the reprojection edges override it with the analytic form
// above; the pinning vectors of the prebuilt libg2o.so drive this routine on them, oracle_binary_edge_pin below).
void linearize_numeric(const Oracle& o, int type, int idx, double* J0, double* J1) {
  int t0, i0, t1, i1;
  edge_vertices(o, type, idx, t0, i0, t1, i1);
  const int dim = edge_dim(o, type, idx);
  const double delta = o.opt.numeric_delta;
  const double scalar = 1.0 / (2 * delta);
  VTemp tmp;
  double ep[kMaxDim], em[kMaxDim];
  for (int side = 0; side < 2; ++side) {
    const int t = side == 0 ? t0 : t1, i = side == 0 ? i0 : i1;
    double* J = side == 0 ? J0 : J1;
    if (t == V_NONE || vertex_fixed(o, t, i)) continue;
    const int d = kVDim[t];
    double addv[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int c = 0; c < d; ++c) {
      addv[c] = delta;
      const void* vp = oplus_copy(o, t, i, addv, tmp);
      compute_error(o, type, idx, side == 0 ? vp : vertex_ptr(o, t0, i0), side == 1 ? vp : (t1 == V_NONE ? nullptr : vertex_ptr(o, t1, i1)), ep);
      addv[c] = -delta;
      vp = oplus_copy(o, t, i, addv, tmp);
      compute_error(o, type, idx, side == 0 ? vp : vertex_ptr(o, t0, i0), side == 1 ? vp : (t1 == V_NONE ? nullptr : vertex_ptr(o, t1, i1)), em);
      addv[c] = 0.0;
      for (int r = 0; r < dim; ++r) J[r * d + c] = scalar * (ep[r] - em[r]);
    }
  }
}

double edge_chi2(const Oracle& o, int type, int idx) {  // base_edge.h:58-61 with a diagonal information matrix
  const int dim = edge_dim(o, type, idx);
  const double* e = &o.err[type][(size_t)kMaxDim * idx];
  double s = 0;
  if (type == E_MONO || type == E_STEREO) {
    const double w = o.e_info[type][idx];
    for (int i = 0; i < dim; ++i) s += e[i] * (w * e[i]);
  } else {
    const int stride = (type == E_BBOX) ? 4 : (type == E_CORNER ? 16 : (type == E_CUB3D ? 9 : 3));
    const double* w = &o.e_info[type][(size_t)stride * idx];
    for (int i = 0; i < dim; ++i) s += e[i] * (w[i] * e[i]);
  }
  return s;
}
double info_at(const Oracle& o, int type, int idx, int i) {
  if (type == E_MONO || type == E_STEREO) return o.e_info[type][idx];
  const int stride = (type == E_BBOX) ? 4 : (type == E_CORNER ? 16 : (type == E_CUB3D ? 9 : 3));
  return o.e_info[type][(size_t)stride * idx + i];
}

// SparseOptimizer::initializeOptimization(level) + buildIndexMapping, sparse_optimizer.cpp:166-267,
// followed by BlockSolver::buildStructure, block_solver.hpp:143-295.
void initialize_optimization(Oracle& o, int lvl) {
  o.active.clear();
  std::vector<uint8_t> vact[4];
  vact[V_POSE].assign(o.n_poses, 0);
  vact[V_POINT].assign(o.n_points, 0);
  vact[V_PLANE].assign(o.n_planes, 0);
  vact[V_CUBOID].assign(o.n_cuboids, 0);
  for (int type = 0; type < E_NTYPES; ++type)  // addEdge order, SURVEY.md appendix A "Edge evaluation order"
    for (int idx = 0; idx < o.n_edges[type]; ++idx) {
      if (o.level[type][idx] != lvl) continue;
      int t0, i0, t1, i1;
      edge_vertices(o, type, idx, t0, i0, t1, i1);
      const bool all_fixed = vertex_fixed(o, t0, i0) && (t1 == V_NONE || vertex_fixed(o, t1, i1));
      if (all_fixed) continue;
      o.active.push_back(EdgeRef{type, idx});
      vact[t0][i0] = 1;
      if (t1 != V_NONE) vact[t1][i1] = 1;
    }
  for (int t = 0; t < 4; ++t) {
    const int n = (int)vact[t].size();
    o.hidx[t].assign(n, -1);
    o.col0[t].assign(n, -1);
  }
  // non-marginalised first (poses, cuboids), then marginalised (planes, points)
  o.red_col0.clear(); o.red_dim.clear();
  o.n_red = 0;
  for (int t : {V_POSE, V_CUBOID})
    for (int i = 0; i < (int)vact[t].size(); ++i)
      if (vact[t][i] && !vertex_fixed(o, t, i)) {
        o.hidx[t][i] = (int)o.red_col0.size();
        o.col0[t][i] = o.n_red;
        o.red_col0.push_back(o.n_red);
        o.red_dim.push_back(kVDim[t]);
        o.n_red += kVDim[t];
      }
  o.n_red_blocks = (int)o.red_col0.size();
  o.lm_type.clear(); o.lm_vidx.clear();
  for (int t : {V_PLANE, V_POINT})
    for (int i = 0; i < (int)vact[t].size(); ++i)
      if (vact[t][i] && !vertex_fixed(o, t, i)) {
        o.hidx[t][i] = (int)o.lm_type.size();
        o.col0[t][i] = 3 * (int)o.lm_type.size();
        o.lm_type.push_back(t);
        o.lm_vidx.push_back(i);
      }
  o.n_lm = (int)o.lm_type.size();
  o.do_schur = o.n_lm > 0;  // optimization_algorithm_with_hessian.cpp:50-73
  // Hpl structure: per landmark the sorted set of reduced blocks it touches
  std::vector<std::vector<int>> touch(o.n_lm);
  for (const EdgeRef& er : o.active) {
    if (er.type != E_PLANE && er.type != E_MONO && er.type != E_STEREO) continue;
    int t0, i0, t1, i1;
    edge_vertices(o, er.type, er.idx, t0, i0, t1, i1);
    const int l = o.hidx[t0][i0], rb = o.hidx[t1][i1];
    if (l >= 0 && rb >= 0) touch[l].push_back(rb);
  }
  o.hpl_start.assign(o.n_lm + 1, 0);
  o.hpl_rb.clear();
  for (int l = 0; l < o.n_lm; ++l) {
    std::sort(touch[l].begin(), touch[l].end());
    touch[l].erase(std::unique(touch[l].begin(), touch[l].end()), touch[l].end());
    o.hpl_start[l] = (int)o.hpl_rb.size();
    o.hpl_rb.insert(o.hpl_rb.end(), touch[l].begin(), touch[l].end());
  }
  o.hpl_start[o.n_lm] = (int)o.hpl_rb.size();
  o.hpl.assign(o.hpl_rb.size() * 18, 0.0);
  for (int type : {E_PLANE, E_MONO, E_STEREO}) o.edge_slot[type].assign(o.n_edges[type], -1);
  for (const EdgeRef& er : o.active) {
    if (er.type != E_PLANE && er.type != E_MONO && er.type != E_STEREO) continue;
    int t0, i0, t1, i1;
    edge_vertices(o, er.type, er.idx, t0, i0, t1, i1);
    const int l = o.hidx[t0][i0], rb = o.hidx[t1][i1];
    if (l < 0 || rb < 0) continue;
    const int* first = &o.hpl_rb[o.hpl_start[l]];
    const int* last = &o.hpl_rb[0] + o.hpl_start[l + 1];
    o.edge_slot[er.type][er.idx] = (int)(std::lower_bound(first, last, rb) - &o.hpl_rb[0]);
  }
  o.Hpp.assign((size_t)o.n_red * o.n_red, 0.0);
  o.Hll.assign((size_t)o.n_lm * 9, 0.0);
  o.b.assign((size_t)o.n_red + 3 * (size_t)o.n_lm, 0.0);
  o.x.assign(o.b.size(), 0.0);
  o.S.assign(o.do_schur ? o.Hpp.size() : 0, 0.0);
  o.bs.assign(o.n_red, 0.0);
  o.coeff.assign(o.b.size(), 0.0);
  o.Dinv.assign((size_t)o.n_lm * 9, 0.0);
}

// SparseOptimizer::computeActiveErrors, sparse_optimizer.cpp:61-76
void compute_active_errors(Oracle& o) {
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
  for (size_t k = 0; k < o.active.size(); ++k) {
    const EdgeRef& er = o.active[k];
    int t0, i0, t1, i1;
    edge_vertices(o, er.type, er.idx, t0, i0, t1, i1);
    compute_error(o, er.type, er.idx, vertex_ptr(o, t0, i0), t1 == V_NONE ? nullptr : vertex_ptr(o, t1, i1),
                  &o.err[er.type][(size_t)kMaxDim * er.idx]);
  }
}
// SparseOptimizer::activeRobustChi2, sparse_optimizer.cpp:100-114
double active_robust_chi2(const Oracle& o) {
  double chi = 0.0, rho[3];
  const size_t na = o.active.size();
  for (size_t k = 0; k < na; ++k) {
    const EdgeRef& er = o.active[o.variant == 1 ? na - 1 - k : k];
    const double c = edge_chi2(o, er.type, er.idx);
    const double hub = o.e_huber[er.type][er.idx];
    if (o.robust[er.type][er.idx] && hub > 0) {
      huber_robustify(c, hub, rho);
      chi += rho[0];
    } else {
      chi += c;
    }
  }
  return chi;
}

inline double& HPP(Oracle& o, int r, int c) { return o.Hpp[(size_t)c * o.n_red + r]; }
// s += v, with the rounding error collected in c when the compensated variant is on
inline void acc_comp(bool on, double& s, double& c, double v) {
  if (!on) { s += v; return; }
  const double t = s + v;
  c += (std::fabs(s) >= std::fabs(v)) ? (s - t) + v : (v - t) + s;
  s = t;
}

// BlockSolver::buildSystem, block_solver.hpp:502-560 with constructQuadraticForm of
// base_binary_edge.hpp:55-120 / base_unary_edge.hpp:43-72.
#ifdef _OPENMP
// libppo_oracle_omp.so ONLY (secondary "best CPU" line of bench.py): the same per-edge arithmetic as build_system below, the edge
// loop split over the threads.  Reduced-variable blocks (pose / cuboid diagonal blocks, b) go to per-thread compact accumulators that
// are summed at the end; landmark blocks are added with atomics (a landmark's edges are neighbours in the list: contention is rare);
// the few pose-cuboid coupling blocks too.  The sums are taken in another order than the serial build: same values to rounding.
void build_system_omp(Oracle& o) {
  std::fill(o.Hpp.begin(), o.Hpp.end(), 0.0);
  std::fill(o.Hll.begin(), o.Hll.end(), 0.0);
  std::fill(o.hpl.begin(), o.hpl.end(), 0.0);
  std::fill(o.b.begin(), o.b.end(), 0.0);
  const int nb = o.n_red_blocks, n = o.n_red;
  const size_t na = o.active.size();
  const int nt = omp_get_max_threads();
  std::vector<std::vector<double>>&Hd = o.omp_Hd, &bp = o.omp_bp;
  if ((int)Hd.size() < nt) { Hd.resize(nt); bp.resize(nt); }
#pragma omp parallel num_threads(nt)
  {
    const int t = omp_get_thread_num();
    Hd[t].assign((size_t)nb * 81, 0.0);
    bp[t].assign(n, 0.0);
    double J0[kMaxDim * 9], J1[kMaxDim * 9];
#pragma omp for schedule(static)
    for (size_t k = 0; k < na; ++k) {
      const EdgeRef& er = o.active[k];
      int t0, i0, t1, i1;
      edge_vertices(o, er.type, er.idx, t0, i0, t1, i1);
      const int dim = edge_dim(o, er.type, er.idx);
      linearize(o, er.type, er.idx, J0, J1);
      const double* e = &o.err[er.type][(size_t)kMaxDim * er.idx];
      double rho1 = 1.0;
      const double hub = o.e_huber[er.type][er.idx];
      if (o.robust[er.type][er.idx] && hub > 0) {
        double rho[3];
        huber_robustify(edge_chi2(o, er.type, er.idx), hub, rho);
        rho1 = rho[1];
      }
      double w[kMaxDim], omega_r[kMaxDim];
      for (int r = 0; r < dim; ++r) {
        const double om = info_at(o, er.type, er.idx, r);
        w[r] = rho1 * om;
        omega_r[r] = -(om * e[r]) * rho1;
      }
      const bool free0 = !vertex_fixed(o, t0, i0) && o.hidx[t0][i0] >= 0;
      const bool free1 = t1 != V_NONE && !vertex_fixed(o, t1, i1) && o.hidx[t1][i1] >= 0;
      const int d0 = kVDim[t0], d1 = t1 == V_NONE ? 0 : kVDim[t1];
      auto add_diag = [&](int vt, int vi, const double* J, int d) {
        if (vt == V_POINT || vt == V_PLANE) {
          double* D = &o.Hll[(size_t)9 * o.hidx[vt][vi]];
          double* bl = &o.b[(size_t)o.n_red + o.col0[vt][vi]];
          for (int a = 0; a < 3; ++a) {
            double sb = 0;
            for (int r = 0; r < dim; ++r) sb += J[r * d + a] * omega_r[r];
#pragma omp atomic
            bl[a] += sb;
            for (int c = 0; c < 3; ++c) {
              double sacc = 0;
              for (int r = 0; r < dim; ++r) sacc += J[r * d + a] * w[r] * J[r * d + c];
#pragma omp atomic
              D[a * 3 + c] += sacc;
            }
          }
        } else {
          double* H = &Hd[t][(size_t)81 * o.hidx[vt][vi]];
          double* bb = &bp[t][o.col0[vt][vi]];
          for (int a = 0; a < d; ++a) {
            double sb = 0;
            for (int r = 0; r < dim; ++r) sb += J[r * d + a] * omega_r[r];
            bb[a] += sb;
            for (int c = 0; c < d; ++c) {
              double sacc = 0;
              for (int r = 0; r < dim; ++r) sacc += J[r * d + a] * w[r] * J[r * d + c];
              H[a * 9 + c] += sacc;
            }
          }
        }
      };
      if (free0) add_diag(t0, i0, J0, d0);
      if (free1) add_diag(t1, i1, J1, d1);
      if (free0 && free1) {
        if (er.type == E_BBOX || er.type == E_CORNER || er.type == E_CUB3D) {
          const int r0 = o.col0[t0][i0], c0 = o.col0[t1][i1];
          for (int a = 0; a < d0; ++a)
            for (int c = 0; c < d1; ++c) {
              double sacc = 0;
              for (int r = 0; r < dim; ++r) sacc += J0[r * d0 + a] * w[r] * J1[r * d1 + c];
#pragma omp atomic
              HPP(o, r0 + a, c0 + c) += sacc;
            }
        } else if (er.type == E_PLANE || er.type == E_MONO || er.type == E_STEREO) {
          double* blk = &o.hpl[(size_t)18 * o.edge_slot[er.type][er.idx]];
          for (int a = 0; a < 6; ++a)
            for (int c = 0; c < 3; ++c) {
              double sacc = 0;
              for (int r = 0; r < dim; ++r) sacc += J1[r * 6 + a] * w[r] * J0[r * 3 + c];
              blk[a * 3 + c] += sacc;
            }
        }
      }
    }
  }
  for (int t = 0; t < nt; ++t) {
    for (int blk = 0; blk < nb; ++blk) {
      const int c0 = o.red_col0[blk], d = o.red_dim[blk];
      for (int a = 0; a < d; ++a)
        for (int c = 0; c < d; ++c) HPP(o, c0 + a, c0 + c) += Hd[t][(size_t)81 * blk + a * 9 + c];
    }
    for (int i = 0; i < n; ++i) o.b[i] += bp[t][i];
  }
}
#endif

void build_system(Oracle& o) {
#ifdef _OPENMP
  if (o.variant == 0) { build_system_omp(o); return; }
#endif
  std::fill(o.Hpp.begin(), o.Hpp.end(), 0.0);
  std::fill(o.Hll.begin(), o.Hll.end(), 0.0);
  std::fill(o.hpl.begin(), o.hpl.end(), 0.0);
  std::fill(o.b.begin(), o.b.end(), 0.0);
  const bool comp = o.variant == 2;
  if (comp) { o.Hpp_c.assign(o.Hpp.size(), 0.0); o.b_c.assign(o.b.size(), 0.0); }
  double dummy = 0.0;
  double J0[kMaxDim * 9], J1[kMaxDim * 9];
  const size_t na = o.active.size();
  for (size_t k = 0; k < na; ++k) {
    const EdgeRef& er = o.active[o.variant == 1 ? na - 1 - k : k];
    int t0, i0, t1, i1;
    edge_vertices(o, er.type, er.idx, t0, i0, t1, i1);
    const int dim = edge_dim(o, er.type, er.idx);
    linearize(o, er.type, er.idx, J0, J1);
    const double* e = &o.err[er.type][(size_t)kMaxDim * er.idx];
    // omega_r = -rho1 * Omega * e ; weightedOmega = rho1 * Omega (base_edge.h:96-102: rho2 term dropped)
    double rho1 = 1.0;
    const double hub = o.e_huber[er.type][er.idx];
    if (o.robust[er.type][er.idx] && hub > 0) {
      double rho[3];
      huber_robustify(edge_chi2(o, er.type, er.idx), hub, rho);
      rho1 = rho[1];
    }
    double w[kMaxDim], omega_r[kMaxDim];
    for (int r = 0; r < dim; ++r) {
      const double om = info_at(o, er.type, er.idx, r);
      w[r] = rho1 * om;
      omega_r[r] = -(om * e[r]) * rho1;
    }
    const bool free0 = !vertex_fixed(o, t0, i0) && o.hidx[t0][i0] >= 0;
    const bool free1 = t1 != V_NONE && !vertex_fixed(o, t1, i1) && o.hidx[t1][i1] >= 0;
    const int d0 = kVDim[t0], d1 = t1 == V_NONE ? 0 : kVDim[t1];
    auto add_diag = [&](int t, int i, const double* J, int d) {
      const bool marg = (t == V_POINT || t == V_PLANE);
      if (marg) {
        double* D = &o.Hll[(size_t)9 * o.hidx[t][i]];
        double* bl = &o.b[(size_t)o.n_red + o.col0[t][i]];
        for (int a = 0; a < 3; ++a) {
          double sb = 0;
          for (int r = 0; r < dim; ++r) sb += J[r * d + a] * omega_r[r];
          bl[a] += sb;
          for (int c = 0; c < 3; ++c) {
            double s = 0;
            for (int r = 0; r < dim; ++r) s += J[r * d + a] * w[r] * J[r * d + c];
            D[a * 3 + c] += s;
          }
        }
      } else {
        const int c0 = o.col0[t][i];
        // A unary edge groups its right-hand side differently (base_unary_edge.hpp:62,65): b -= ((rho1 A') Omega) e, the 6 x D
        // product formed first -- against b += B' ((-Omega e) rho1) of a binary edge (base_binary_edge.hpp:74,99-111).  Same
        // value, different rounding; the golden vectors of the prebuilt libg2o.so pin both.
        const bool unary = t1 == V_NONE || o.only_pose;
        for (int a = 0; a < d; ++a) {
          double sb = 0;
          if (unary) {
            for (int r = 0; r < dim; ++r) sb += e[r] * ((rho1 * J[r * d + a]) * info_at(o, er.type, er.idx, r));
            sb = -sb;
          } else {
            for (int r = 0; r < dim; ++r) sb += J[r * d + a] * omega_r[r];
          }
          acc_comp(comp, o.b[c0 + a], comp ? o.b_c[c0 + a] : dummy, sb);
          for (int c = 0; c < d; ++c) {
            double s = 0;
            for (int r = 0; r < dim; ++r) s += J[r * d + a] * w[r] * J[r * d + c];
            acc_comp(comp, HPP(o, c0 + a, c0 + c), comp ? o.Hpp_c[(size_t)(c0 + c) * o.n_red + c0 + a] : dummy, s);
          }
        }
      }
    };
    if (free0) add_diag(t0, i0, J0, d0);
    if (free1) add_diag(t1, i1, J1, d1);
    if (free0 && free1) {
      if (er.type == E_BBOX || er.type == E_CORNER || er.type == E_CUB3D) {  // pose (slot 0) x cuboid (slot 1): upper block of Hpp
        const int r0 = o.col0[t0][i0], c0 = o.col0[t1][i1];
        for (int a = 0; a < d0; ++a)
          for (int c = 0; c < d1; ++c) {
            double s = 0;
            for (int r = 0; r < dim; ++r) s += J0[r * d0 + a] * w[r] * J1[r * d1 + c];
            HPP(o, r0 + a, c0 + c) += s;
          }
      } else if (er.type == E_PLANE || er.type == E_MONO || er.type == E_STEREO) {
        // landmark (slot 0) x pose (slot 1): Hpl(pose, landmark) += B' W A  (block_solver.hpp:240-244)
        double* blk = &o.hpl[(size_t)18 * o.edge_slot[er.type][er.idx]];
        for (int a = 0; a < 6; ++a)
          for (int c = 0; c < 3; ++c) {
            double s = 0;
            for (int r = 0; r < dim; ++r) s += J1[r * 6 + a] * w[r] * J0[r * 3 + c];
            blk[a * 3 + c] += s;
          }
      }
      // E_CUBPL: cuboid x plane block exists structurally in g2o but its Jacobians are identically zero
      // (constant error differentiated numerically), so nothing is accumulated.
    }
  }
  if (comp) {
    for (size_t i = 0; i < o.Hpp.size(); ++i) o.Hpp[i] += o.Hpp_c[i];
    for (size_t i = 0; i < o.b.size(); ++i) o.b[i] += o.b_c[i];
  }
}

// PartialPivLU-based inverse of a 3x3 block: MatrixXd::inverse() on a dynamic matrix, block_solver.hpp:389
bool inverse3(const double* A, double* inv) {
  double lu[3][3];
  int perm[3] = {0, 1, 2};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) lu[i][j] = A[i * 3 + j];
  for (int k = 0; k < 3; ++k) {
    int piv = k;
    double best = std::fabs(lu[k][k]);
    for (int i = k + 1; i < 3; ++i)
      if (std::fabs(lu[i][k]) > best) { best = std::fabs(lu[i][k]); piv = i; }
    if (piv != k) {
      for (int j = 0; j < 3; ++j) std::swap(lu[k][j], lu[piv][j]);
      std::swap(perm[k], perm[piv]);
    }
    if (lu[k][k] != 0.0)
      for (int i = k + 1; i < 3; ++i) lu[i][k] /= lu[k][k];
    for (int i = k + 1; i < 3; ++i)
      for (int j = k + 1; j < 3; ++j) lu[i][j] -= lu[i][k] * lu[k][j];
  }
  for (int c = 0; c < 3; ++c) {  // solve L U x = P e_c
    double y[3];
    for (int i = 0; i < 3; ++i) {
      double s = (perm[i] == c) ? 1.0 : 0.0;
      for (int j = 0; j < i; ++j) s -= lu[i][j] * y[j];
      y[i] = s;
    }
    for (int i = 2; i >= 0; --i) {
      double s = y[i];
      for (int j = i + 1; j < 3; ++j) s -= lu[i][j] * inv[j * 3 + c];
      inv[i * 3 + c] = s / lu[i][i];
    }
  }
  return true;
}

// Eigen::LDLT<MatrixXd>::compute + isPositive + solve as used by linear_solver_dense.h:104-112.
// Restated from Eigen 3.3's published algorithm: left-looking unblocked L D L^T on the lower triangle with
// symmetric pivoting on the largest remaining |diagonal|, sign tracking for isPositive(), and a solve that
// treats |d_i| <= DBL_MIN as zero.  A is n x n column-major and is overwritten.
bool ldlt_solve(std::vector<double>& A, int n, const double* rhs, double* xout) {
  std::vector<int> tr(n);
  std::vector<double> temp(n);
  enum { ZeroSign, PositiveSemiDef, NegativeSemiDef, Indefinite } sign = ZeroSign;
  auto M = [&](int r, int c) -> double& { return A[(size_t)c * n + r]; };
  bool all_zero_diag = false;
  for (int k = 0; k < n && !all_zero_diag; ++k) {
    int big = k;
    double best = std::fabs(M(k, k));
    for (int i = k + 1; i < n; ++i)
      if (std::fabs(M(i, i)) > best) { best = std::fabs(M(i, i)); big = i; }
    tr[k] = big;
    if (k != big) {
      const int s = n - big - 1;
      for (int j = 0; j < k; ++j) std::swap(M(k, j), M(big, j));
      for (int i = 0; i < s; ++i) std::swap(M(big + 1 + i, k), M(big + 1 + i, big));
      std::swap(M(k, k), M(big, big));
      for (int i = k + 1; i < big; ++i) std::swap(M(i, k), M(big, i));
    }
    const int rs = n - k - 1;
    if (k > 0) {
      for (int j = 0; j < k; ++j) temp[j] = M(j, j) * M(k, j);
      double s = 0;
      for (int j = 0; j < k; ++j) s += M(k, j) * temp[j];
      M(k, k) -= s;
      if (rs > 0) {
        double* colk = &A[(size_t)k * n + k + 1];
        for (int j = 0; j < k; ++j) {
          const double tj = temp[j];
          if (tj == 0.0) continue;
          double* colj = &A[(size_t)j * n + k + 1];
          for (int i = 0; i < rs; ++i) colk[i] -= colj[i] * tj;
        }
      }
    }
    const double akk = M(k, k);
    const bool pivot_ok = std::fabs(akk) > 0.0;
    if (k == 0 && !pivot_ok) {
      for (int j = 0; j < n; ++j) tr[j] = j;
      all_zero_diag = true;
      break;
    }
    if (rs > 0 && pivot_ok) {
      double* colk = &A[(size_t)k * n + k + 1];
      for (int i = 0; i < rs; ++i) colk[i] /= akk;
    }
    if (sign == PositiveSemiDef) {
      if (akk < 0) sign = Indefinite;
    } else if (sign == NegativeSemiDef) {
      if (akk > 0) sign = Indefinite;
    } else if (sign == ZeroSign) {
      if (akk > 0) sign = PositiveSemiDef;
      else if (akk < 0) sign = NegativeSemiDef;
    }
  }
  const bool positive = (sign == PositiveSemiDef || sign == ZeroSign);
  if (!positive) return false;
  std::vector<double> y(rhs, rhs + n);
  for (int k = 0; k < n; ++k) std::swap(y[k], y[tr[k]]);
  for (int j = 0; j < n; ++j) {  // L y = P b
    const double yj = y[j];
    const double* col = &A[(size_t)j * n];
    for (int i = j + 1; i < n; ++i) y[i] -= col[i] * yj;
  }
  const double tol = std::numeric_limits<double>::min();
  for (int i = 0; i < n; ++i) {
    const double d = M(i, i);
    y[i] = std::fabs(d) > tol ? y[i] / d : 0.0;
  }
  for (int j = n - 1; j >= 0; --j) {  // L' z = y
    const double* col = &A[(size_t)j * n];
    double s = y[j];
    for (int i = j + 1; i < n; ++i) s -= col[i] * y[i];
    y[j] = s;
  }
  for (int k = n - 1; k >= 0; --k) std::swap(y[k], y[tr[k]]);
  std::copy(y.begin(), y.end(), xout);
  return true;
}

// LinearSolverDense::solve, linear_solver_dense.h:65-113: mirror the upper blocks, factor, gate on isPositive

#ifdef _OPENMP
// libppo_oracle_omp.so ONLY (bench.py's secondary "best CPU" line; never the checker): what a multi-core CPU implementation of this
// path would do with the dense solve -- a right-looking blocked Cholesky, 64-wide panels, the panel solve and the trailing update
// spread over the threads (two fork-joins per panel instead of one per column), column-major so that every inner loop is a
// contiguous axpy.  Not Eigen's pivoted LDLT: same solution to rounding, "not positive" = a non-positive pivot.
bool blocked_llt_solve_omp(std::vector<double>& A, int n, const double* rhs, double* xout) {
  constexpr int NB = 64;
  bool ok = true;
  for (int k0 = 0; k0 < n && ok; k0 += NB) {
    const int kb = std::min(NB, n - k0);
    for (int k = k0; k < k0 + kb; ++k) {  // diagonal block, unblocked
      double* colk = &A[(size_t)k * n];
      const double d = colk[k];
      if (!(d > 0.0)) { ok = false; break; }
      const double l = std::sqrt(d), il = 1.0 / l;
      colk[k] = l;
      for (int i = k + 1; i < k0 + kb; ++i) colk[i] *= il;
      for (int j = k + 1; j < k0 + kb; ++j) {
        const double s = colk[j];
        double* colj = &A[(size_t)j * n];
        for (int i = j; i < k0 + kb; ++i) colj[i] -= colk[i] * s;
      }
    }
    if (!ok) break;
    const int r0 = k0 + kb, rs = n - r0;
    if (rs <= 0) continue;
    // panel solve: rows below the diagonal block, X L_kk' = B, row slices of 64 per task
#pragma omp parallel for schedule(static)
    for (int ib = 0; ib < (rs + NB - 1) / NB; ++ib) {
      const int i0 = r0 + ib * NB, i1 = std::min(n, i0 + NB);
      for (int k = k0; k < k0 + kb; ++k) {
        double* colk = &A[(size_t)k * n];
        const double il = 1.0 / colk[k];
        for (int i = i0; i < i1; ++i) colk[i] *= il;
        for (int j = k + 1; j < k0 + kb; ++j) {
          const double s = colk[j];
          double* colj = &A[(size_t)j * n];
          for (int i = i0; i < i1; ++i) colj[i] -= colk[i] * s;
        }
      }
    }
    // trailing update, lower triangle, one column block per task
#pragma omp parallel for schedule(dynamic, 1)
    for (int jb = 0; jb < (rs + NB - 1) / NB; ++jb) {
      const int j0 = r0 + jb * NB, j1 = std::min(n, j0 + NB);
      for (int j = j0; j < j1; ++j) {
        double* colj = &A[(size_t)j * n];
        for (int k = k0; k < k0 + kb; ++k) {
          const double* colk = &A[(size_t)k * n];
          const double s = colk[j];
          for (int i = j; i < n; ++i) colj[i] -= colk[i] * s;
        }
      }
    }
  }
  if (!ok) return false;
  std::vector<double> y(rhs, rhs + n);
  for (int j = 0; j < n; ++j) {
    const double* col = &A[(size_t)j * n];
    const double yj = (y[j] /= col[j]);
    for (int i = j + 1; i < n; ++i) y[i] -= col[i] * yj;
  }
  for (int j = n - 1; j >= 0; --j) {
    const double* col = &A[(size_t)j * n];
    double sacc = y[j];
    for (int i = j + 1; i < n; ++i) sacc -= col[i] * y[i];
    y[j] = sacc / col[j];
  }
  std::copy(y.begin(), y.end(), xout);
  return true;
}
#endif

bool dense_solve(Oracle& o, std::vector<double>& Aupper, const double* rhs, double* xout) {
  const int n = o.n_red;
#ifdef _OPENMP
  static thread_local std::vector<double> H;  // (kept: see Oracle::omp_S)
  H = Aupper;
#else
  std::vector<double> H(Aupper);
#endif
  for (int c = 0; c < n; ++c)
    for (int r = 0; r < c; ++r) H[(size_t)r * n + c] = H[(size_t)c * n + r];  // lower(c,r) = upper(r,c)
#ifdef _OPENMP
  return blocked_llt_solve_omp(H, n, rhs, xout);
#else
  return ldlt_solve(H, n, rhs, xout);
#endif
}

// BlockSolver::setLambda / restoreDiagonal, block_solver.hpp:564-604
void set_lambda(Oracle& o, double lambda) {
  o.diag_backup_p.resize(o.n_red);
  o.diag_backup_l.resize((size_t)3 * o.n_lm);
  for (int i = 0; i < o.n_red; ++i) {
    o.diag_backup_p[i] = HPP(o, i, i);
    HPP(o, i, i) += lambda;
  }
  for (int l = 0; l < o.n_lm; ++l)
    for (int a = 0; a < 3; ++a) {
      o.diag_backup_l[3 * l + a] = o.Hll[(size_t)9 * l + 4 * a];
      o.Hll[(size_t)9 * l + 4 * a] += lambda;
    }
}
void restore_diagonal(Oracle& o) {
  for (int i = 0; i < o.n_red; ++i) HPP(o, i, i) = o.diag_backup_p[i];
  for (int l = 0; l < o.n_lm; ++l)
    for (int a = 0; a < 3; ++a) o.Hll[(size_t)9 * l + 4 * a] = o.diag_backup_l[3 * l + a];
}

// BlockSolver::solve, block_solver.hpp:354-486
bool block_solve(Oracle& o) {
  const int n = o.n_red;
  if (!o.do_schur) return dense_solve(o, o.Hpp, o.b.data(), o.x.data());
  o.S = o.Hpp;
  std::fill(o.coeff.begin(), o.coeff.begin() + n, 0.0);
#ifdef _OPENMP
  if (o.variant == 0) {
    // libppo_oracle_omp.so ONLY: the landmark loop split over the threads, each with its own copy of S and of the right-hand side's
    // Schur term, summed at the end (same per-landmark arithmetic; another summation order)
    const int nt = omp_get_max_threads();
    std::vector<std::vector<double>>&St = o.omp_S, &ct = o.omp_c;
    if ((int)St.size() < nt) { St.resize(nt); ct.resize(nt); }
#pragma omp parallel num_threads(nt)
    {
      const int t = omp_get_thread_num();
      St[t].assign((size_t)n * n, 0.0);  // (same size as last time: a fill, no allocation)
      ct[t].assign(n, 0.0);
      double* S = St[t].data();
#pragma omp for schedule(static)
      for (int l = 0; l < o.n_lm; ++l) {
        double* Dinv = &o.Dinv[(size_t)9 * l];
        inverse3(&o.Hll[(size_t)9 * l], Dinv);
        const double* bl = &o.b[(size_t)n + 3 * l];
        double db[3];
        for (int a = 0; a < 3; ++a) db[a] = Dinv[a * 3] * bl[0] + Dinv[a * 3 + 1] * bl[1] + Dinv[a * 3 + 2] * bl[2];
        for (int s1 = o.hpl_start[l]; s1 < o.hpl_start[l + 1]; ++s1) {
          const double* Bi = &o.hpl[(size_t)18 * s1];
          const int r0 = o.red_col0[o.hpl_rb[s1]];
          double BDinv[18];
          for (int a = 0; a < 6; ++a)
            for (int c = 0; c < 3; ++c) BDinv[a * 3 + c] = Bi[a * 3] * Dinv[c] + Bi[a * 3 + 1] * Dinv[3 + c] + Bi[a * 3 + 2] * Dinv[6 + c];
          for (int a = 0; a < 6; ++a) ct[t][r0 + a] += Bi[a * 3] * db[0] + Bi[a * 3 + 1] * db[1] + Bi[a * 3 + 2] * db[2];
          for (int s2 = s1; s2 < o.hpl_start[l + 1]; ++s2) {
            const double* Bj = &o.hpl[(size_t)18 * s2];
            const int c0 = o.red_col0[o.hpl_rb[s2]];
            for (int a = 0; a < 6; ++a)
              for (int c = 0; c < 6; ++c)
                S[(size_t)(c0 + c) * n + r0 + a] -= BDinv[a * 3] * Bj[c * 3] + BDinv[a * 3 + 1] * Bj[c * 3 + 1] + BDinv[a * 3 + 2] * Bj[c * 3 + 2];
          }
        }
      }
#pragma omp for schedule(static)
      for (int c = 0; c < n; ++c)
        for (int tt = 0; tt < nt; ++tt) {
          const double* src = &St[tt][(size_t)c * n];
          double* dst = &o.S[(size_t)c * n];
          for (int r = 0; r <= c; ++r) dst[r] += src[r];
        }
    }
    for (int tt = 0; tt < nt; ++tt)
      for (int i = 0; i < n; ++i) o.coeff[i] += ct[tt][i];
  } else
#endif
  {
  const bool comp = o.variant == 2;
  if (comp) { o.S_c.assign(o.S.size(), 0.0); o.coeff_c.assign(o.coeff.size(), 0.0); }
  double dummy = 0.0;
  for (int lk = 0; lk < o.n_lm; ++lk) {
    const int l = o.variant == 1 ? o.n_lm - 1 - lk : lk;
    double* Dinv = &o.Dinv[(size_t)9 * l];
    inverse3(&o.Hll[(size_t)9 * l], Dinv);
    const double* bl = &o.b[(size_t)n + 3 * l];
    double db[3];
    for (int a = 0; a < 3; ++a) db[a] = Dinv[a * 3] * bl[0] + Dinv[a * 3 + 1] * bl[1] + Dinv[a * 3 + 2] * bl[2];
    for (int s1 = o.hpl_start[l]; s1 < o.hpl_start[l + 1]; ++s1) {
      const double* Bi = &o.hpl[(size_t)18 * s1];
      const int r0 = o.red_col0[o.hpl_rb[s1]];
      double BDinv[18];
      for (int a = 0; a < 6; ++a)
        for (int c = 0; c < 3; ++c)
          BDinv[a * 3 + c] = Bi[a * 3] * Dinv[c] + Bi[a * 3 + 1] * Dinv[3 + c] + Bi[a * 3 + 2] * Dinv[6 + c];
      for (int a = 0; a < 6; ++a) acc_comp(comp, o.coeff[r0 + a], comp ? o.coeff_c[r0 + a] : dummy, Bi[a * 3] * db[0] + Bi[a * 3 + 1] * db[1] + Bi[a * 3 + 2] * db[2]);
      for (int s2 = s1; s2 < o.hpl_start[l + 1]; ++s2) {
        const double* Bj = &o.hpl[(size_t)18 * s2];
        const int c0 = o.red_col0[o.hpl_rb[s2]];
        for (int a = 0; a < 6; ++a)
          for (int c = 0; c < 6; ++c)
            acc_comp(comp, o.S[(size_t)(c0 + c) * n + r0 + a], comp ? o.S_c[(size_t)(c0 + c) * n + r0 + a] : dummy,
                     -(BDinv[a * 3] * Bj[c * 3] + BDinv[a * 3 + 1] * Bj[c * 3 + 1] + BDinv[a * 3 + 2] * Bj[c * 3 + 2]));
      }
    }
  }
  if (comp) {
    for (size_t i = 0; i < o.S.size(); ++i) o.S[i] += o.S_c[i];
    for (int i = 0; i < n; ++i) o.coeff[i] += o.coeff_c[i];
  }
  }
  for (int i = 0; i < n; ++i) o.bs[i] = o.b[i] - o.coeff[i];
  if (!dense_solve(o, o.S, o.bs.data(), o.x.data())) return false;
  for (int l = 0; l < o.n_lm; ++l) {  // x_l = Dinv (b_l - B' x_p), :461-481
    double cl[3] = {o.b[(size_t)n + 3 * l], o.b[(size_t)n + 3 * l + 1], o.b[(size_t)n + 3 * l + 2]};
    for (int s1 = o.hpl_start[l]; s1 < o.hpl_start[l + 1]; ++s1) {
      const double* Bi = &o.hpl[(size_t)18 * s1];
      const int r0 = o.red_col0[o.hpl_rb[s1]];
      for (int c = 0; c < 3; ++c)
        for (int a = 0; a < 6; ++a) cl[c] += Bi[a * 3 + c] * (-o.x[r0 + a]);
    }
    const double* Dinv = &o.Dinv[(size_t)9 * l];
    for (int a = 0; a < 3; ++a) o.x[(size_t)n + 3 * l + a] = Dinv[a * 3] * cl[0] + Dinv[a * 3 + 1] * cl[1] + Dinv[a * 3 + 2] * cl[2];
  }
  return true;
}

// SparseOptimizer::update, sparse_optimizer.cpp:422-435
void apply_update(Oracle& o) {
  for (int i = 0; i < o.n_poses; ++i)
    if (o.hidx[V_POSE][i] >= 0) o.pose[i] = pose_oplus(o.pose[i], &o.x[o.col0[V_POSE][i]]);
  for (int i = 0; i < o.n_cuboids; ++i)
    if (o.hidx[V_CUBOID][i] >= 0)
      o.cuboid[i] = cuboid_oplus(o.cuboid[i], &o.x[o.col0[V_CUBOID][i]], o.cuboid_flags[i], &o.cuboid_fixedscale[3 * i]);
  for (int i = 0; i < o.n_planes; ++i)
    if (o.hidx[V_PLANE][i] >= 0) o.plane[i] = plane_oplus(o.plane[i], &o.x[(size_t)o.n_red + o.col0[V_PLANE][i]]);
  for (int i = 0; i < o.n_points; ++i)
    if (o.hidx[V_POINT][i] >= 0) {
      const double* u = &o.x[(size_t)o.n_red + o.col0[V_POINT][i]];
      o.point[i] = add(o.point[i], mk(u[0], u[1], u[2]));
    }
}

bool stop_requested(const volatile uint8_t* f) { return f && *f; }

// OptimizationAlgorithmLevenberg::solve, optimization_algorithm_levenberg.cpp:61-164.
// returns 0 OK, 1 Terminate
int lm_solve(Oracle& o, int iteration, const volatile uint8_t* stop, ppo_ba_iter_stat& st) {
  compute_active_errors(o);
  double currentChi = active_robust_chi2(o);
  double tempChi = currentChi;
  const double iniChi = currentChi;
  st.chi2_before = currentChi;
  build_system(o);
  if (iteration == 0) {  // computeLambdaInit, :166-180
    double maxDiagonal = 0.;
    for (int i = 0; i < o.n_red; ++i) maxDiagonal = std::max(std::fabs(HPP(o, i, i)), maxDiagonal);
    for (int l = 0; l < o.n_lm; ++l)
      for (int a = 0; a < 3; ++a) maxDiagonal = std::max(std::fabs(o.Hll[(size_t)9 * l + 4 * a]), maxDiagonal);
    o.lambda = o.opt.lm_tau * maxDiagonal;
    o.ni = 2;
    o.nBad = 0;
  }
  double rho = 0;
  int qmax = 0;
  std::vector<SE3> pose_bak;
  std::vector<V3> point_bak;
  std::vector<Plane> plane_bak;
  std::vector<Cuboid> cuboid_bak;
  do {
    pose_bak = o.pose; point_bak = o.point; plane_bak = o.plane; cuboid_bak = o.cuboid;  // push()
    set_lambda(o, o.lambda);
    const bool ok2 = block_solve(o);
    apply_update(o);
    restore_diagonal(o);
    compute_active_errors(o);
    tempChi = active_robust_chi2(o);
    if (!ok2) tempChi = std::numeric_limits<double>::max();
    rho = (currentChi - tempChi);
    double scale = 0.;  // computeScale, :182-189
    for (size_t j = 0; j < o.x.size(); ++j) scale += o.x[j] * (o.lambda * o.x[j] + o.b[j]);
    scale += 1e-3;
    rho /= scale;
    st.lambda = o.lambda;
    if (rho > 0 && std::isfinite(tempChi)) {
      double alpha = 1. - std::pow((2 * rho - 1), 3);
      alpha = std::min(alpha, o.opt.lm_upper);
      const double scaleFactor = std::max(o.opt.lm_lower, alpha);
      o.lambda *= scaleFactor;
      o.ni = 2;
      currentChi = tempChi;
      st.accepted = 1;
    } else {
      o.lambda *= o.ni;
      o.ni *= 2;
      o.pose.swap(pose_bak); o.point.swap(point_bak); o.plane.swap(plane_bak); o.cuboid.swap(cuboid_bak);  // pop()
      st.accepted = 0;
    }
    qmax++;
    o.stats.n_trials++;
  } while (rho < 0 && qmax < o.opt.lm_max_trials && !stop_requested(stop));
  st.trials = qmax;
  st.chi2_after = currentChi;
  if (qmax == o.opt.lm_max_trials || rho == 0) return 1;
  if ((iniChi - currentChi) * 1e3 < iniChi) o.nBad++;
  else o.nBad = 0;
  if (o.nBad >= 3) return 1;
  return 0;
}

// SparseOptimizer::optimize, sparse_optimizer.cpp:354-419
void optimize(Oracle& o, int pass, int iterations, const volatile uint8_t* stop) {
  if (o.n_red == 0) return;  // "0 vertices to optimize"
  bool ok = true;
  for (int i = 0; i < iterations && !stop_requested(stop) && ok; ++i) {
    ppo_ba_iter_stat st{};
    st.pass = pass;
    st.iteration = i;
    const int res = lm_solve(o, i, stop, st);
    st.result = res;
    ok = (res == 0);
    if (o.stats.n_iters < PPO_BA_MAX_ITER_STATS) o.stats.iters[o.stats.n_iters] = st;
    o.stats.n_iters++;
  }
}

bool depth_positive(const Oracle& o, int type, int idx) {  // types_six_dof_expmap.h:181-185, 213-217
  const V3 pc = se3_map(o.pose[o.e_v1[type][idx]], o.point[o.e_v0[type][idx]]);
  return pc[2] > 0.0;
}
double err_norm(const Oracle& o, int type, int idx) {
  const int dim = edge_dim(o, type, idx);
  const double* e = &o.err[type][(size_t)kMaxDim * idx];
  double s = 0;
  for (int i = 0; i < dim; ++i) s += e[i] * e[i];
  return std::sqrt(s);
}

// Optimizer.cc:2736-2833
void classify_outliers(Oracle& o) {
  for (int i = 0; i < o.n_edges[E_MONO]; ++i) {
    if (edge_chi2(o, E_MONO, i) > o.opt.chi2_mono || !depth_positive(o, E_MONO, i)) o.level[E_MONO][i] = 1;
    o.robust[E_MONO][i] = 0;
  }
  for (int i = 0; i < o.n_edges[E_STEREO]; ++i) {
    if (edge_chi2(o, E_STEREO, i) > o.opt.chi2_stereo || !depth_positive(o, E_STEREO, i)) o.level[E_STEREO][i] = 1;
    o.robust[E_STEREO][i] = 0;
  }
  for (int i = 0; i < o.n_edges[E_BBOX]; ++i)
    if (err_norm(o, E_BBOX, i) > o.opt.th_bbox_norm) o.level[E_BBOX][i] = 1;
  for (int i = 0; i < o.n_edges[E_CORNER]; ++i)
    if (err_norm(o, E_CORNER, i) > o.opt.th_corner_norm) o.level[E_CORNER][i] = 1;
  for (int i = 0; i < o.n_edges[E_CUB3D]; ++i)  // Optimizer.cc:1875-1882 (LocalBACameraPointCuboids2D); the kernel stays
    if (err_norm(o, E_CUB3D, i) > o.opt.th_cub3d_norm) o.level[E_CUB3D][i] = 1;
  for (int i = 0; i < o.n_edges[E_PLANE]; ++i) {
    const double th = o.pl_kind[i] == 0 ? o.opt.chi2_plane : o.opt.chi2_plane_vp;
    if (edge_chi2(o, E_PLANE, i) > th) o.level[E_PLANE][i] = 1;
    o.robust[E_PLANE][i] = 0;
  }
  // point-cuboid: no filter (:2816-2824); cuboid-plane: vpEdgesCuboidPlane is never filled (:2660, 2826-2833)
}

void reset_state(Oracle& o) {
  o.pose = o.pose0; o.point = o.point0; o.plane = o.plane0; o.cuboid = o.cuboid0;
  for (int t = 0; t < E_NTYPES; ++t) {
    o.level[t].assign(o.n_edges[t], 0);
    o.robust[t].assign(o.n_edges[t], 1);
    o.err[t].assign((size_t)kMaxDim * o.n_edges[t], 0.0);
  }
  o.ran = false;
}

template <class T>
void copy_in(std::vector<T>& dst, const T* src, size_t n, T fill = T()) {
  if (src) dst.assign(src, src + n);
  else dst.assign(n, fill);
}

}  // namespace

extern "C" {

typedef struct Oracle* oracle_handle;

int oracle_ba_create(const ppo_ba_problem* p, const ppo_ba_options* opt, oracle_handle* out) {
  if (!p || !opt || !out) return PPO_BA_ERR_ARG;
  Oracle* o = new Oracle();
  o->opt = *opt;
  o->n_poses = p->n_poses; o->n_points = p->n_points; o->n_planes = p->n_planes; o->n_cuboids = p->n_cuboids;
  o->pose0.resize(p->n_poses);
  for (int i = 0; i < p->n_poses; ++i) {
    const double* q = &p->pose_qt[7 * i];
    // Converter::toSE3Quat builds SE3Quat(R,t) whose ctor normalises (se3quat.h:58-60)
    o->pose0[i] = se3_make(Quat{q[0], q[1], q[2], q[3]}, mk(q[4], q[5], q[6]));
  }
  copy_in(o->pose_fixed, p->pose_fixed, (size_t)p->n_poses, (uint8_t)0);
  copy_in(o->pose_cam, p->pose_cam, (size_t)5 * p->n_poses, 0.0);
  o->point0.resize(p->n_points);
  for (int i = 0; i < p->n_points; ++i) o->point0[i] = mk(p->point_xyz[3 * i], p->point_xyz[3 * i + 1], p->point_xyz[3 * i + 2]);
  copy_in(o->point_fixed, p->point_fixed, (size_t)p->n_points, (uint8_t)0);
  o->plane0.resize(p->n_planes);
  for (int i = 0; i < p->n_planes; ++i) o->plane0[i] = plane_from_cv(&p->plane_coef[4 * i]);
  o->cuboid0.resize(p->n_cuboids);
  for (int i = 0; i < p->n_cuboids; ++i) {
    const double* q = &p->cuboid_qt[7 * i];
    o->cuboid0[i].pose = se3_make(Quat{q[0], q[1], q[2], q[3]}, mk(q[4], q[5], q[6]));
    o->cuboid0[i].scale = mk(p->cuboid_scale[3 * i], p->cuboid_scale[3 * i + 1], p->cuboid_scale[3 * i + 2]);
  }
  copy_in(o->cuboid_flags, p->cuboid_flags, (size_t)p->n_cuboids, (uint8_t)(PPO_CUBOID_FIX_ROLLPITCH | PPO_CUBOID_FIX_HEIGHT));
  copy_in(o->cuboid_fixedscale, p->cuboid_fixedscale, (size_t)3 * p->n_cuboids, 0.0);

  auto set_edges = [&](int t, int n, const int32_t* v0, const int32_t* v1, const double* meas, int mdim, const double* info,
                       int idim, const double* huber) {
    o->n_edges[t] = n;
    copy_in(o->e_v0[t], v0, (size_t)n, 0);
    if (v1) copy_in(o->e_v1[t], v1, (size_t)n, 0);
    copy_in(o->e_meas[t], meas, (size_t)mdim * n, 0.0);
    copy_in(o->e_info[t], info, (size_t)idim * n, 1.0);
    copy_in(o->e_huber[t], huber, (size_t)n, 0.0);
  };
  set_edges(E_MONO, p->n_mono, p->mono_point, p->mono_pose, p->mono_obs, 2, p->mono_info, 1, p->mono_huber);
  set_edges(E_STEREO, p->n_stereo, p->stereo_point, p->stereo_pose, p->stereo_obs, 3, p->stereo_info, 1, p->stereo_huber);
  set_edges(E_PLANE, p->n_plane_edges, p->pl_plane, p->pl_pose, p->pl_meas, 4, p->pl_info, 3, p->pl_huber);
  copy_in(o->pl_kind, p->pl_kind, (size_t)p->n_plane_edges, (uint8_t)0);
  o->pl_meas_plane.resize(p->n_plane_edges);
  for (int i = 0; i < p->n_plane_edges; ++i) o->pl_meas_plane[i] = plane_from_cv(&p->pl_meas[4 * i]);
  set_edges(E_BBOX, p->n_bbox, p->bbox_pose, p->bbox_cuboid, p->bbox_meas, 4, p->bbox_info, 4, p->bbox_huber);
  copy_in(o->bbox_K, p->bbox_K, (size_t)9 * p->n_bbox, 0.0);
  set_edges(E_CORNER, p->n_corner, p->corner_pose, p->corner_cuboid, p->corner_meas, 16, p->corner_info, 16, p->corner_huber);
  copy_in(o->corner_K, p->corner_K, (size_t)9 * p->n_corner, 0.0);
  set_edges(E_PTCUB, p->n_ptcub, p->ptcub_cuboid, nullptr, nullptr, 0, p->ptcub_info, 3, p->ptcub_huber);
  copy_in(o->ptcub_off, p->ptcub_offsets, (size_t)p->n_ptcub + 1, 0);
  copy_in(o->ptcub_pts, p->ptcub_points, (size_t)3 * (p->n_ptcub ? o->ptcub_off[p->n_ptcub] : 0), 0.0);
  copy_in(o->ptcub_margin, p->ptcub_margin_ratio, (size_t)p->n_ptcub, 1.0);
  copy_in(o->ptcub_prior, p->ptcub_prior_half, (size_t)3 * p->n_ptcub, 0.0);
  set_edges(E_CUBPL, p->n_cubpl, p->cubpl_cuboid, p->cubpl_plane, p->cubpl_meas, 3, p->cubpl_info, 3, p->cubpl_huber);
  set_edges(E_CUB3D, p->n_cub3d, p->cub3d_pose, p->cub3d_cuboid, p->cub3d_meas, 10, p->cub3d_info, 9, p->cub3d_huber);
  o->cub3d_meas.resize(p->n_cub3d);
  for (int i = 0; i < p->n_cub3d; ++i) {  // g2o::cuboid measurement as MapCuboid::cuboid_local_meas holds it (values as given)
    const double* m = &p->cub3d_meas[10 * i];
    o->cub3d_meas[i].pose.r = Quat{m[0], m[1], m[2], m[3]};
    o->cub3d_meas[i].pose.t = mk(m[4], m[5], m[6]);
    o->cub3d_meas[i].scale = mk(m[7], m[8], m[9]);
  }
  reset_state(*o);
  *out = o;
  return PPO_BA_OK;
}

int oracle_ba_reset(oracle_handle o) {
  if (!o) return PPO_BA_ERR_ARG;
  reset_state(*o);
  return PPO_BA_OK;
}

// Optimizer.cc:2723-2838
int oracle_ba_run(oracle_handle o, const volatile uint8_t* stop, ppo_ba_stats* stats) {
  if (!o) return PPO_BA_ERR_ARG;
  std::memset(&o->stats, 0, sizeof(o->stats));
  const auto t0 = std::chrono::steady_clock::now();
  int rc = PPO_BA_OK;
  if (stop_requested(stop)) return PPO_BA_ABORTED;
  initialize_optimization(*o, 0);
  o->stats.n_reduced = o->n_red;
  o->stats.n_active_edges_pass1 = (int)o->active.size();
  optimize(*o, 0, o->opt.iters_pass1, stop);
  const auto t1 = std::chrono::steady_clock::now();
  bool more = !stop_requested(stop) && o->opt.iters_pass2 > 0;
  if (stop_requested(stop)) rc = PPO_BA_ABORTED;
  auto t2 = t1;
  if (more) {
    classify_outliers(*o);
    t2 = std::chrono::steady_clock::now();
    initialize_optimization(*o, 0);
    o->stats.n_active_edges_pass2 = (int)o->active.size();
    optimize(*o, 1, o->opt.iters_pass2, stop);
    if (stop_requested(stop)) rc = PPO_BA_ABORTED;
  }
  const auto t3 = std::chrono::steady_clock::now();
  auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  o->stats.wall_ms_total = ms(t0, t3);
  o->stats.wall_ms_pass1 = ms(t0, t1);
  o->stats.wall_ms_classify = ms(t1, t2);
  o->stats.wall_ms_pass2 = ms(t2, t3);
  o->ran = true;
  if (stats) *stats = o->stats;
  return rc;
}

int oracle_ba_read(oracle_handle o, ppo_ba_result* r) {
  if (!o || !r) return PPO_BA_ERR_ARG;
  auto put_se3 = [](double* dst, const SE3& T) {
    dst[0] = T.r.x; dst[1] = T.r.y; dst[2] = T.r.z; dst[3] = T.r.w; dst[4] = T.t[0]; dst[5] = T.t[1]; dst[6] = T.t[2];
  };
  if (r->pose_qt) for (int i = 0; i < o->n_poses; ++i) put_se3(&r->pose_qt[7 * i], o->pose[i]);
  if (r->point_xyz) for (int i = 0; i < o->n_points; ++i) for (int a = 0; a < 3; ++a) r->point_xyz[3 * i + a] = o->point[i][a];
  if (r->plane_coef) for (int i = 0; i < o->n_planes; ++i) for (int a = 0; a < 4; ++a) r->plane_coef[4 * i + a] = o->plane[i].c[a];
  if (r->cuboid_qt) for (int i = 0; i < o->n_cuboids; ++i) put_se3(&r->cuboid_qt[7 * i], o->cuboid[i].pose);
  if (r->cuboid_scale) for (int i = 0; i < o->n_cuboids; ++i) for (int a = 0; a < 3; ++a) r->cuboid_scale[3 * i + a] = o->cuboid[i].scale[a];
  auto put_chi = [&](int t, double* chi, uint8_t* lvl) {
    for (int i = 0; i < o->n_edges[t]; ++i) {
      if (chi) chi[i] = edge_chi2(*o, t, i);
      if (lvl) lvl[i] = o->level[t][i];
    }
  };
  put_chi(E_MONO, r->mono_chi2, r->mono_level);
  put_chi(E_STEREO, r->stereo_chi2, r->stereo_level);
  put_chi(E_PLANE, r->pl_chi2, r->pl_level);
  put_chi(E_BBOX, r->bbox_chi2, r->bbox_level);
  put_chi(E_CORNER, r->corner_chi2, r->corner_level);
  put_chi(E_PTCUB, r->ptcub_chi2, nullptr);
  put_chi(E_CUBPL, r->cubpl_chi2, nullptr);
  put_chi(E_CUB3D, r->cub3d_chi2, r->cub3d_level);
  if (r->cub3d_errnorm) for (int i = 0; i < o->n_edges[E_CUB3D]; ++i) r->cub3d_errnorm[i] = err_norm(*o, E_CUB3D, i);
  if (r->mono_depth_positive) for (int i = 0; i < o->n_edges[E_MONO]; ++i) r->mono_depth_positive[i] = depth_positive(*o, E_MONO, i);
  if (r->stereo_depth_positive) for (int i = 0; i < o->n_edges[E_STEREO]; ++i) r->stereo_depth_positive[i] = depth_positive(*o, E_STEREO, i);
  if (r->bbox_errnorm) for (int i = 0; i < o->n_edges[E_BBOX]; ++i) r->bbox_errnorm[i] = err_norm(*o, E_BBOX, i);
  if (r->corner_errnorm) for (int i = 0; i < o->n_edges[E_CORNER]; ++i) r->corner_errnorm[i] = err_norm(*o, E_CORNER, i);
  return PPO_BA_OK;
}

// Test aid: the oracle's CURRENT objective -- the active edge set, levels and robust kernels as its last
// initializeOptimization / classification left them (after oracle_ba_run: the pass-2 cost) -- evaluated at a FOREIGN state (any
// pointer may be NULL: that vertex class keeps the oracle's own state).  computeActiveErrors + activeRobustChi2
// (sparse_optimizer.cpp:61-76, 100-114) on a copy: the oracle's state and stored errors are untouched.
// tests/test_parity_gpu.py: "the device's final state is an equally good minimiser of the REFERENCE cost".
int oracle_ba_chi2_at(oracle_handle o, const double* pose_qt, const double* point_xyz, const double* plane_coef, const double* cuboid_qt,
                      const double* cuboid_scale, double* chi2) {
  if (!o || !chi2) return PPO_BA_ERR_ARG;
  const std::vector<SE3> pose = o->pose;
  const std::vector<V3> point = o->point;
  const std::vector<Plane> plane = o->plane;
  const std::vector<Cuboid> cuboid = o->cuboid;
  std::vector<double> err[E_NTYPES];
  for (int t = 0; t < E_NTYPES; ++t) err[t] = o->err[t];
  auto get_se3 = [](const double* q) { SE3 T; T.r = Quat{q[0], q[1], q[2], q[3]}; T.t = mk(q[4], q[5], q[6]); return T; };
  if (pose_qt) for (int i = 0; i < o->n_poses; ++i) o->pose[i] = get_se3(&pose_qt[7 * i]);
  if (point_xyz) for (int i = 0; i < o->n_points; ++i) o->point[i] = mk(point_xyz[3 * i], point_xyz[3 * i + 1], point_xyz[3 * i + 2]);
  if (plane_coef) for (int i = 0; i < o->n_planes; ++i) for (int a = 0; a < 4; ++a) o->plane[i].c[a] = plane_coef[4 * i + a];
  if (cuboid_qt) for (int i = 0; i < o->n_cuboids; ++i) o->cuboid[i].pose = get_se3(&cuboid_qt[7 * i]);
  if (cuboid_scale) for (int i = 0; i < o->n_cuboids; ++i) o->cuboid[i].scale = mk(cuboid_scale[3 * i], cuboid_scale[3 * i + 1], cuboid_scale[3 * i + 2]);
  compute_active_errors(*o);
  *chi2 = active_robust_chi2(*o);
  o->pose = pose; o->point = point; o->plane = plane; o->cuboid = cuboid;
  for (int t = 0; t < E_NTYPES; ++t) o->err[t] = err[t];
  return PPO_BA_OK;
}

void oracle_ba_destroy(oracle_handle o) { delete o; }

// OpenMP build: number of threads of the following calls (the environment variable is read when libgomp is FIRST loaded -- in a process
// that imported PyTorch that happened long before this library, with every visible CPU); the plain build ignores it.  Returns the value in use.
int oracle_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
  return omp_get_max_threads();
#else
  (void)n;
  return 1;
#endif
}

// Test aid for tests/test_oracle_sensitivity.py: 1 = sum the edges (chi2, H, b) and the landmarks (Schur complement) in reverse
// order.  The algorithm, every formula and every input are unchanged; only the order of the floating-point additions differs --
// the freedom g2o itself has (std::map<KeyFrame*, ...> pointer order, Optimizer.cc:2350).
int oracle_ba_set_variant(oracle_handle o, int variant) {
  if (!o || variant < 0 || variant > 2) return PPO_BA_ERR_ARG;
  o->variant = variant;
  return PPO_BA_OK;
}

// ---- step-level hooks for the parity tests ------------------------------------------------------
// One linearisation of the current state at level 0: errors, robust chi2, block system.
int oracle_ba_linearize(oracle_handle o, double* robust_chi2) {
  if (!o) return PPO_BA_ERR_ARG;
  initialize_optimization(*o, 0);
  compute_active_errors(*o);
  if (robust_chi2) *robust_chi2 = active_robust_chi2(*o);
  build_system(*o);
  o->stats.n_reduced = o->n_red;
  return PPO_BA_OK;
}
// Reduced system for a given lambda after oracle_ba_linearize: S (n x n column-major, full symmetric),
// b_schur and the solution x_p.  Landmark range [lm_first, lm_last) restricts the Schur sum (for the
// sharding tests: partial systems of disjoint ranges sum to the full one when add_hpp is set on one rank).
int oracle_ba_reduced(oracle_handle o, double lambda, int lm_first, int lm_last, int add_hpp, double* S, double* bsch, int32_t* n_out) {
  if (!o) return PPO_BA_ERR_ARG;
  const int n = o->n_red;
  if (n_out) *n_out = n;
  if (lm_last < 0 || lm_last > o->n_lm) lm_last = o->n_lm;
  std::vector<double> Sm((size_t)n * n, 0.0), bb(n, 0.0);
  if (add_hpp) {
    for (int c = 0; c < n; ++c)
      for (int r = 0; r <= c; ++r) Sm[(size_t)c * n + r] = o->Hpp[(size_t)c * n + r];
    for (int i = 0; i < n; ++i) { Sm[(size_t)i * n + i] += lambda; bb[i] = o->b[i]; }
  }
  for (int l = lm_first; l < lm_last; ++l) {
    double D[9], Dinv[9];
    for (int k = 0; k < 9; ++k) D[k] = o->Hll[(size_t)9 * l + k];
    D[0] += lambda; D[4] += lambda; D[8] += lambda;
    inverse3(D, Dinv);
    const double* bl = &o->b[(size_t)n + 3 * l];
    double db[3];
    for (int a = 0; a < 3; ++a) db[a] = Dinv[a * 3] * bl[0] + Dinv[a * 3 + 1] * bl[1] + Dinv[a * 3 + 2] * bl[2];
    for (int s1 = o->hpl_start[l]; s1 < o->hpl_start[l + 1]; ++s1) {
      const double* Bi = &o->hpl[(size_t)18 * s1];
      const int r0 = o->red_col0[o->hpl_rb[s1]];
      double BDinv[18];
      for (int a = 0; a < 6; ++a)
        for (int c = 0; c < 3; ++c) BDinv[a * 3 + c] = Bi[a * 3] * Dinv[c] + Bi[a * 3 + 1] * Dinv[3 + c] + Bi[a * 3 + 2] * Dinv[6 + c];
      for (int a = 0; a < 6; ++a) bb[r0 + a] -= Bi[a * 3] * db[0] + Bi[a * 3 + 1] * db[1] + Bi[a * 3 + 2] * db[2];
      for (int s2 = s1; s2 < o->hpl_start[l + 1]; ++s2) {
        const double* Bj = &o->hpl[(size_t)18 * s2];
        const int c0 = o->red_col0[o->hpl_rb[s2]];
        for (int a = 0; a < 6; ++a)
          for (int c = 0; c < 6; ++c)
            Sm[(size_t)(c0 + c) * n + r0 + a] -= BDinv[a * 3] * Bj[c * 3] + BDinv[a * 3 + 1] * Bj[c * 3 + 1] + BDinv[a * 3 + 2] * Bj[c * 3 + 2];
      }
    }
  }
  for (int c = 0; c < n; ++c)
    for (int r = 0; r < c; ++r) Sm[(size_t)r * n + c] = Sm[(size_t)c * n + r];
  if (S) std::copy(Sm.begin(), Sm.end(), S);
  if (bsch) std::copy(bb.begin(), bb.end(), bsch);
  return PPO_BA_OK;
}
int oracle_ba_num_landmarks(oracle_handle o) { return o ? o->n_lm : -1; }
int oracle_has_cuboid3d(void) { return 1; }  // the 9-D camera-object edge (E_CUB3D) is implemented

// ---- hooks of the reference-binary pins (oracle/pin_orbslam_prebuilt.py, tests/test_orbslam_pin.py) ----------------------------
// Overwrite the CURRENT estimates with the given numbers, bit for bit (no quaternion / plane normalisation: the pinning script
// writes its values straight into the reference's vertex objects too).  Any pointer may be NULL.
int oracle_ba_set_state(oracle_handle o, const double* pose_qt, const double* point_xyz, const double* plane_coef, const double* cuboid_qt,
                        const double* cuboid_scale) {
  if (!o) return PPO_BA_ERR_ARG;
  auto get_se3 = [](const double* q) { SE3 T; T.r = Quat{q[0], q[1], q[2], q[3]}; T.t = mk(q[4], q[5], q[6]); return T; };
  if (pose_qt) for (int i = 0; i < o->n_poses; ++i) o->pose[i] = get_se3(&pose_qt[7 * i]);
  if (point_xyz) for (int i = 0; i < o->n_points; ++i) o->point[i] = mk(point_xyz[3 * i], point_xyz[3 * i + 1], point_xyz[3 * i + 2]);
  if (plane_coef) for (int i = 0; i < o->n_planes; ++i) for (int a = 0; a < 4; ++a) o->plane[i].c[a] = plane_coef[4 * i + a];
  if (cuboid_qt) for (int i = 0; i < o->n_cuboids; ++i) o->cuboid[i].pose = get_se3(&cuboid_qt[7 * i]);
  if (cuboid_scale) for (int i = 0; i < o->n_cuboids; ++i) o->cuboid[i].scale = mk(cuboid_scale[3 * i], cuboid_scale[3 * i + 1], cuboid_scale[3 * i + 2]);
  return PPO_BA_OK;
}
// The measurement plane of plane edge `idx` as the oracle holds it (after its Converter::toPlane3D / Plane3D(v) restatement).
int oracle_ba_plane_meas(oracle_handle o, int idx, double out[4]) {
  if (!o || idx < 0 || idx >= o->n_edges[E_PLANE]) return PPO_BA_ERR_ARG;
  std::memcpy(out, o->pl_meas_plane[idx].c, 4 * sizeof(double));
  return PPO_BA_OK;
}
// One block of the system assembled by the last oracle_ba_linearize: H(a, b) row-major (rows of vertex a, columns of vertex b) and
// the right-hand side of vertex a.  Vertex types: 0 pose, 1 point, 2 plane, 3 cuboid.  tb < 0: the diagonal block of a.
// Returns the number of rows, 0 when a vertex is not in the system (fixed / inactive) or the pair has no block.
int oracle_ba_block(oracle_handle o, int ta, int ia, int tb, int ib, double* H, double* bvec) {
  if (!o || ta < 0 || ta > 3) return PPO_BA_ERR_ARG;
  if (tb < 0) { tb = ta; ib = ia; }
  if (ia < 0 || ia >= (int)o->hidx[ta].size() || ib < 0 || ib >= (int)o->hidx[tb].size()) return PPO_BA_ERR_ARG;
  if (o->hidx[ta][ia] < 0 || o->hidx[tb][ib] < 0) return 0;
  const int da = kVDim[ta], db = kVDim[tb];
  const bool la = (ta == V_POINT || ta == V_PLANE), lb = (tb == V_POINT || tb == V_PLANE);
  if (bvec)
    for (int r = 0; r < da; ++r) bvec[r] = la ? o->b[(size_t)o->n_red + o->col0[ta][ia] + r] : o->b[o->col0[ta][ia] + r];
  if (!H) return da;
  if (!la && !lb) {  // reduced partition: upper blocks are stored
    const int r0 = o->col0[ta][ia], c0 = o->col0[tb][ib];
    for (int r = 0; r < da; ++r)  // a diagonal block is filled completely, an off-diagonal one only above the diagonal
      for (int c = 0; c < db; ++c) H[r * db + c] = (r0 <= c0) ? HPP(*o, r0 + r, c0 + c) : HPP(*o, c0 + c, r0 + r);
    return da;
  }
  if (la && lb) {
    if (ta != tb || ia != ib) return 0;
    std::copy(&o->Hll[(size_t)9 * o->hidx[ta][ia]], &o->Hll[(size_t)9 * o->hidx[ta][ia]] + 9, H);
    return 3;
  }
  // landmark x reduced block (either order): hpl slots hold the 6 x 3 block (pose rows)
  const int l = la ? o->hidx[ta][ia] : o->hidx[tb][ib];
  const int rb = la ? o->hidx[tb][ib] : o->hidx[ta][ia];
  const int dred = la ? db : da;
  if (dred != 6) { std::fill(H, H + da * db, 0.0); return da; }  // cuboid x plane: structurally present in g2o, identically zero
  for (int s = o->hpl_start[l]; s < o->hpl_start[l + 1]; ++s)
    if (o->hpl_rb[s] == rb) {
      const double* blk = &o->hpl[(size_t)18 * s];
      for (int a = 0; a < 6; ++a)
        for (int c = 0; c < 3; ++c) {
          if (la) H[c * 6 + a] = blk[a * 3 + c];
          else H[a * 3 + c] = blk[a * 3 + c];
        }
      return da;
    }
  return 0;
}
// dense symmetric solve through the restated LDLT (column-major full matrix in, lower triangle used)
int oracle_ldlt_solve(const double* A, int n, const double* rhs, double* x) {
  std::vector<double> M(A, A + (size_t)n * n);
  return ldlt_solve(M, n, rhs, x) ? 1 : 0;
}
// per-edge residual and Jacobians at the current state.  type: 0 plane,1 mono,2 stereo,3 bbox,4 corner,5 ptcub,6 cubpl
int oracle_edge_eval(oracle_handle o, int type, int idx, double* e, double* J0, double* J1) {
  if (!o || type < 0 || type >= E_NTYPES || idx < 0 || idx >= o->n_edges[type]) return PPO_BA_ERR_ARG;
  int t0, i0, t1, i1;
  edge_vertices(*o, type, idx, t0, i0, t1, i1);
  double ee[kMaxDim], j0[kMaxDim * 9] = {0}, j1[kMaxDim * 9] = {0};
  compute_error(*o, type, idx, vertex_ptr(*o, t0, i0), t1 == V_NONE ? nullptr : vertex_ptr(*o, t1, i1), ee);
  linearize(*o, type, idx, j0, j1);
  const int dim = edge_dim(*o, type, idx);
  if (e) std::copy(ee, ee + dim, e);
  if (J0) std::copy(j0, j0 + dim * kVDim[t0], J0);
  if (J1 && t1 != V_NONE) std::copy(j1, j1 + dim * kVDim[t1], J1);
  return dim;
}

// ---- leaf functions for the known-answer / golden tests ------------------------------------------
void oracle_se3_exp(const double upd[6], double qt[7]) {
  const SE3 T = se3_exp(upd);
  qt[0] = T.r.x; qt[1] = T.r.y; qt[2] = T.r.z; qt[3] = T.r.w; qt[4] = T.t[0]; qt[5] = T.t[1]; qt[6] = T.t[2];
}
void oracle_pose_oplus(const double qt[7], const double upd[6], double out[7]) {
  const SE3 T = pose_oplus(se3_make(Quat{qt[0], qt[1], qt[2], qt[3]}, mk(qt[4], qt[5], qt[6])), upd);
  out[0] = T.r.x; out[1] = T.r.y; out[2] = T.r.z; out[3] = T.r.w; out[4] = T.t[0]; out[5] = T.t[1]; out[6] = T.t[2];
}
void oracle_plane_oplus(const double c[4], const double upd[3], double out[4]) {
  const Plane p = plane_oplus(plane_from_vector(c), upd);
  std::memcpy(out, p.c, sizeof(p.c));
}
// Plane3D(const Vector4D&) (via_cv = 0) or Converter::toPlane3D (via_cv = 1: sign flip before the normalisation)
void oracle_plane_normalize(const double v[4], int via_cv, double out[4]) {
  const Plane p = via_cv ? plane_from_cv(v) : plane_from_vector(v);
  std::memcpy(out, p.c, sizeof(p.c));
}
void oracle_exptwist_norollpitch(const double upd[6], double qt[7]) {
  const SE3 T = exptwist_norollpitch(upd);
  qt[0] = T.r.x; qt[1] = T.r.y; qt[2] = T.r.z; qt[3] = T.r.w; qt[4] = T.t[0]; qt[5] = T.t[1]; qt[6] = T.t[2];
}
void oracle_plane_ominus(const double a[4], const double b[4], int kind, double* e) {
  const Plane pa = plane_from_vector(a), pb = plane_from_vector(b);
  if (kind == 0) plane_ominus(pa, pb, e);
  else if (kind == 1) plane_ominus_ver(pa, pb, e);
  else plane_ominus_par(pa, pb, e);
}
void oracle_plane_transform(const double qt[7], const double c[4], double out[4]) {
  const Plane p = plane_transform(se3_make(Quat{qt[0], qt[1], qt[2], qt[3]}, mk(qt[4], qt[5], qt[6])), plane_from_vector(c));
  std::memcpy(out, p.c, sizeof(p.c));
}
void oracle_cuboid_oplus(const double qt[7], const double scale[3], const double upd[9], unsigned flags, const double* fixedscale,
                         double out_qt[7], double out_scale[3]) {
  Cuboid c;
  c.pose = se3_make(Quat{qt[0], qt[1], qt[2], qt[3]}, mk(qt[4], qt[5], qt[6]));
  c.scale = mk(scale[0], scale[1], scale[2]);
  const Cuboid n = cuboid_oplus(c, upd, flags, fixedscale);
  out_qt[0] = n.pose.r.x; out_qt[1] = n.pose.r.y; out_qt[2] = n.pose.r.z; out_qt[3] = n.pose.r.w;
  out_qt[4] = n.pose.t[0]; out_qt[5] = n.pose.t[1]; out_qt[6] = n.pose.t[2];
  for (int i = 0; i < 3; ++i) out_scale[i] = n.scale[i];
}
void oracle_cuboid_project(const double cqt[7], const double scale[3], const double pose_qt[7], const double K[9], double uv[16], double bbox[4]) {
  Cuboid c;
  c.pose = se3_make(Quat{cqt[0], cqt[1], cqt[2], cqt[3]}, mk(cqt[4], cqt[5], cqt[6]));
  c.scale = mk(scale[0], scale[1], scale[2]);
  const SE3 T = se3_make(Quat{pose_qt[0], pose_qt[1], pose_qt[2], pose_qt[3]}, mk(pose_qt[4], pose_qt[5], pose_qt[6]));
  double c2[8][2];
  cuboid_project_corners(c, T, K, c2);
  for (int k = 0; k < 8; ++k) { uv[2 * k] = c2[k][0]; uv[2 * k + 1] = c2[k][1]; }
  cuboid_project_bbox(c, T, K, bbox);
}
void oracle_huber(double e, double delta, double rho[3]) { huber_robustify(e, delta, rho); }

// ---- Optimizer::PoseOptimization, src/Optimizer.cc:247-459 ---------------------------------------------------------------
// One free pose, fixed map points, unary reprojection edges; four rounds of optimize(10) from the same initial pose with a
// re-classification of every observation after each round.  Runs on the engine above: a fixed point vertex contributes no
// block, which is what the unary edges amount to; `only_pose` selects their residual / Jacobian flavour.
static thread_local ppo_ba_stats* t_pose_trace = nullptr;  // oracle_pose_optimize_trace: LM statistics of the LAST frame of the call
int oracle_pose_optimize(const ppo_pose_problem* p, const ppo_pose_options* po, ppo_pose_result* r) {
  if (!p || !po || !r || !r->pose_qt) return PPO_BA_ERR_ARG;
  if (po->rounds < 0 || po->rounds > PPO_POSE_MAX_ROUNDS) return PPO_BA_ERR_ARG;
  for (int f = 0; f < p->n_frames; ++f) {
    const int e0 = p->edge_offsets[f], e1 = p->edge_offsets[f + 1], n = e1 - e0;
    ppo_pose_stats st{};
    for (int e = e0; e < e1; ++e) {  // OPT:297, 331: pFrame->mvbOutlier[i] = false
      if (r->outlier) r->outlier[e] = 0;
      if (r->chi2) r->chi2[e] = 0.0;
    }
    std::copy(&p->pose_qt[7 * f], &p->pose_qt[7 * f] + 7, &r->pose_qt[7 * f]);
    if (r->n_inliers) r->n_inliers[f] = 0;
    if (n < po->min_correspondences) {  // OPT:372-373
      if (r->stats) r->stats[f] = st;
      continue;
    }
    std::vector<double> pts((size_t)3 * n), m_obs, s_obs, m_info, s_info, m_hub, s_hub;
    std::vector<uint8_t> pfix(n, 1);
    std::vector<int32_t> m_pt, m_pose, s_pt, s_pose, m_edge, s_edge;
    for (int k = 0; k < n; ++k) {
      const int e = e0 + k;
      for (int c = 0; c < 3; ++c) pts[3 * k + c] = p->xw[3 * (size_t)e + c];
      const double* ob = &p->obs[3 * (size_t)e];
      if (ob[2] < 0) {  // OPT:294
        m_pt.push_back(k); m_pose.push_back(0); m_edge.push_back(e);
        m_obs.push_back(ob[0]); m_obs.push_back(ob[1]);
        m_info.push_back(p->inv_sigma2[e]); m_hub.push_back((double)po->huber_mono);
      } else {
        s_pt.push_back(k); s_pose.push_back(0); s_edge.push_back(e);
        s_obs.push_back(ob[0]); s_obs.push_back(ob[1]); s_obs.push_back(ob[2]);
        s_info.push_back(p->inv_sigma2[e]); s_hub.push_back((double)po->huber_stereo);
      }
    }
    ppo_ba_problem P{};
    const uint8_t free_pose = 0;
    P.n_poses = 1; P.pose_qt = &p->pose_qt[7 * f]; P.pose_fixed = &free_pose; P.pose_cam = &p->cam[5 * f];
    P.n_points = n; P.point_xyz = pts.data(); P.point_fixed = pfix.data();
    P.n_mono = (int32_t)m_pt.size(); P.mono_point = m_pt.data(); P.mono_pose = m_pose.data(); P.mono_obs = m_obs.data();
    P.mono_info = m_info.data(); P.mono_huber = m_hub.data();
    P.n_stereo = (int32_t)s_pt.size(); P.stereo_point = s_pt.data(); P.stereo_pose = s_pose.data(); P.stereo_obs = s_obs.data();
    P.stereo_info = s_info.data(); P.stereo_huber = s_hub.data();
    ppo_ba_options O{};
    O.abi_version = PPO_BA_ABI_VERSION;
    O.lm_tau = po->lm_tau; O.lm_max_trials = po->lm_max_trials; O.lm_lower = po->lm_lower; O.lm_upper = po->lm_upper;
    O.numeric_delta = 1e-9;
    oracle_handle o = nullptr;
    const int rc = oracle_ba_create(&P, &O, &o);
    if (rc != PPO_BA_OK) return rc;
    o->only_pose = true;
    std::memset(&o->stats, 0, sizeof(o->stats));
    std::vector<uint8_t> out_m(P.n_mono, 0), out_s(P.n_stereo, 0);
    int nBad = 0;
    for (int it = 0; it < po->rounds; ++it) {
      o->pose = o->pose0;              // OPT:385 vSE3->setEstimate(Converter::toSE3Quat(pFrame->mTcw))
      // OPT:415, 444 `if(it==2) e->setRobustKernel(0)` at the end of round 2: the first robust_rounds (3) rounds keep the kernel
      for (int t = 0; t < 2; ++t) std::fill(o->robust[t == 0 ? E_MONO : E_STEREO].begin(), o->robust[t == 0 ? E_MONO : E_STEREO].end(), (uint8_t)(it < po->robust_rounds));
      initialize_optimization(*o, 0);  // OPT:386
      const int it0 = o->stats.n_iters;
      optimize(*o, it, po->iters_per_round, nullptr);  // OPT:387
      if (it == 0 && o->stats.n_iters > it0) st.chi2_initial = o->stats.iters[0].chi2_before;
      if (o->stats.n_iters > it0 && o->stats.n_iters <= PPO_BA_MAX_ITER_STATS) st.chi2_final = o->stats.iters[o->stats.n_iters - 1].chi2_after;
      nBad = 0;
      for (int t = 0; t < 2; ++t) {  // OPT:390-417 (mono), 419-446 (stereo)
        const int type = t == 0 ? E_MONO : E_STEREO;
        std::vector<uint8_t>& out = t == 0 ? out_m : out_s;
        const float thr = t == 0 ? po->chi2_mono : po->chi2_stereo;
        for (int i = 0; i < o->n_edges[type]; ++i) {
          if (out[i]) compute_error(*o, type, i, &o->point[o->e_v0[type][i]], &o->pose[0], &o->err[type][(size_t)kMaxDim * i]);
          const double c = edge_chi2(*o, type, i);
          const float chi2 = (float)c;
          if (chi2 > thr) { out[i] = 1; o->level[type][i] = 1; nBad++; }
          else { out[i] = 0; o->level[type][i] = 0; }
          const int e = t == 0 ? m_edge[i] : s_edge[i];
          if (r->outlier) r->outlier[e] = out[i];
          if (r->chi2) r->chi2[e] = c;
        }
      }
      st.rounds_run = it + 1;
      if (n < po->min_edges) break;  // OPT:448-449
    }
    st.lm_iterations = o->stats.n_iters;
    st.lm_trials = o->stats.n_trials;
    st.n_bad = nBad;
    st.lambda_final = o->lambda;
    const SE3& T = o->pose[0];
    double* q = &r->pose_qt[7 * f];
    q[0] = T.r.x; q[1] = T.r.y; q[2] = T.r.z; q[3] = T.r.w; q[4] = T.t[0]; q[5] = T.t[1]; q[6] = T.t[2];
    if (r->n_inliers) r->n_inliers[f] = n - nBad;  // OPT:458
    if (r->stats) r->stats[f] = st;
    if (t_pose_trace) *t_pose_trace = o->stats;
    oracle_ba_destroy(o);
  }
  return PPO_BA_OK;
}
// The same call, additionally returning the per-iteration LM statistics (pass = round, trials, lambda of the last trial, chi2
// before / after) of the last frame: the pose-only result struct has no room for them, tests/test_g2o_pose_pin.py holds them
// against the reference's own run.
int oracle_pose_optimize_trace(const ppo_pose_problem* p, const ppo_pose_options* po, ppo_pose_result* r, ppo_ba_stats* trace) {
  if (trace) std::memset(trace, 0, sizeof(*trace));
  t_pose_trace = trace;
  const int rc = oracle_pose_optimize(p, po, r);
  t_pose_trace = nullptr;
  return rc;
}

// residual and Jacobian of one unary OnlyPose edge (for the golden vectors of the prebuilt libg2o.so)
void oracle_onlypose_eval(const double qt[7], const double cam[5], const double xw[3], const double obs[3], int stereo, double* e, double* J) {
  Oracle o;
  o.only_pose = true;
  o.n_poses = 1; o.n_points = 1;
  o.pose.assign(1, se3_make(Quat{qt[0], qt[1], qt[2], qt[3]}, mk(qt[4], qt[5], qt[6])));
  o.point.assign(1, mk(xw[0], xw[1], xw[2]));
  o.pose_cam.assign(cam, cam + 5);
  o.pose_fixed.assign(1, 0); o.point_fixed.assign(1, 1);
  const int type = stereo ? E_STEREO : E_MONO;
  o.n_edges[type] = 1;
  o.e_v0[type].assign(1, 0); o.e_v1[type].assign(1, 0);
  o.e_meas[type].assign(obs, obs + (stereo ? 3 : 2));
  compute_error(o, type, 0, &o.point[0], &o.pose[0], e);
  double J0[9];
  linearize(o, type, 0, J0, J);
}

// One binary reprojection edge (point = vertex 0, pose = vertex 1) taken through computeError, the analytic linearizeOplus,
// the NUMERIC linearizeOplusXi/Xj of the base class and constructQuadraticForm, for the golden vectors that
// oracle/pin_g2o_prebuilt.py produced by calling exactly those functions in the prebuilt libg2o.so.
// Outputs (row-major): e[dim], Jpt[dim x 3], Jpose[dim x 6] analytic, Jnpt / Jnpose numeric (left untouched for a fixed
// vertex), Hll[3 x 3], Hpp[6 x 6], Hlp[3 x 6] (point rows, pose columns), bl[3], bp[6] (zero where g2o writes nothing).
int oracle_binary_edge_pin(const double qt[7], const double cam[5], const double xw[3], const double obs[3], int stereo, double info_w,
                           double huber_delta, int fix_point, int fix_pose, double* e, double* Jpt, double* Jpose, double* Jnpt,
                           double* Jnpose, double* Hll, double* Hpp, double* Hlp, double* bl, double* bp) {
  Oracle o;
  ppo_ba_options O{};
  O.numeric_delta = 1e-9;
  o.opt = O;
  o.n_poses = 1; o.n_points = 1;
  // the pinning script writes q and t straight into the vertex (no SE3Quat constructor, no normalisation)
  SE3 T;
  T.r = Quat{qt[0], qt[1], qt[2], qt[3]};
  T.t = mk(qt[4], qt[5], qt[6]);
  o.pose0.assign(1, T);
  o.point0.assign(1, mk(xw[0], xw[1], xw[2]));
  o.pose_cam.assign(cam, cam + 5);
  o.pose_fixed.assign(1, (uint8_t)(fix_pose != 0));
  o.point_fixed.assign(1, (uint8_t)(fix_point != 0));
  const int type = stereo ? E_STEREO : E_MONO, dim = stereo ? 3 : 2;
  o.n_edges[type] = 1;
  o.e_v0[type].assign(1, 0); o.e_v1[type].assign(1, 0);
  o.e_meas[type].assign(obs, obs + dim);
  o.e_info[type].assign(1, info_w);
  o.e_huber[type].assign(1, huber_delta);
  reset_state(o);
  if (fix_point && fix_pose) return 0;  // all vertices fixed: the edge is not active (sparse_optimizer.cpp:234)
  initialize_optimization(o, 0);
  compute_active_errors(o);
  std::copy(&o.err[type][0], &o.err[type][0] + dim, e);
  double J0[kMaxDim * 9] = {0}, J1[kMaxDim * 9] = {0};
  linearize(o, type, 0, J0, J1);
  std::copy(J0, J0 + dim * 3, Jpt);
  std::copy(J1, J1 + dim * 6, Jpose);
  linearize_numeric(o, type, 0, Jnpt, Jnpose);
  build_system(o);
  for (int i = 0; i < 9; ++i) Hll[i] = o.n_lm ? o.Hll[i] : 0.0;
  for (int r = 0; r < 6; ++r)
    for (int c = 0; c < 6; ++c) Hpp[r * 6 + c] = o.n_red ? o.Hpp[(size_t)c * o.n_red + r] : 0.0;
  for (int a = 0; a < 6; ++a)
    for (int c = 0; c < 3; ++c) Hlp[c * 6 + a] = (o.n_lm && o.n_red) ? o.hpl[a * 3 + c] : 0.0;
  for (int i = 0; i < 3; ++i) bl[i] = o.n_lm ? o.b[(size_t)o.n_red + i] : 0.0;
  for (int i = 0; i < 6; ++i) bp[i] = o.n_red ? o.b[i] : 0.0;
  return dim;
}

// One unary OnlyPose edge (Optimizer::PoseOptimization, OPT:247-459; the same driver EdgePointCuboidOnlyObject runs through) taken
// through computeError, BaseEdge::chi2, linearizeOplus and BaseUnaryEdge::constructQuadraticForm (base_unary_edge.hpp:43-72) with a
// RobustKernelHuber, for the golden vectors oracle/pin_g2o_prebuilt.py produced by calling exactly those functions in the prebuilt
// libg2o.so.  The oracle models the unary edge as a reprojection edge whose point vertex is fixed (`only_pose` selects the OnlyPose
// residual / Jacobian flavour): a fixed vertex contributes no block, which is what a unary edge amounts to.
// Outputs (row-major): e[dim], chi2, J[dim x 6], H[6 x 6], b[6] (H and b zero when the pose is fixed: nothing is written, :52).
int oracle_unary_edge_pin(const double qt[7], const double cam[5], const double xw[3], const double obs[3], int stereo, double info_w,
                          double huber_delta, int fix_pose, double* e, double* chi2, double* J, double* H, double* b) {
  Oracle o;
  ppo_ba_options O{};
  O.numeric_delta = 1e-9;
  o.opt = O;
  o.only_pose = true;
  o.n_poses = 1; o.n_points = 1;
  SE3 T;  // q and t are written straight into the vertex by the pinning script (no normalisation)
  T.r = Quat{qt[0], qt[1], qt[2], qt[3]};
  T.t = mk(qt[4], qt[5], qt[6]);
  o.pose0.assign(1, T);
  o.point0.assign(1, mk(xw[0], xw[1], xw[2]));
  o.pose_cam.assign(cam, cam + 5);
  o.pose_fixed.assign(1, (uint8_t)0);  // (the fixed case is evaluated on a free pose and its blocks dropped below)
  o.point_fixed.assign(1, (uint8_t)1);
  const int type = stereo ? E_STEREO : E_MONO, dim = stereo ? 3 : 2;
  o.n_edges[type] = 1;
  o.e_v0[type].assign(1, 0); o.e_v1[type].assign(1, 0);
  o.e_meas[type].assign(obs, obs + dim);
  o.e_info[type].assign(1, info_w);
  o.e_huber[type].assign(1, huber_delta);
  reset_state(o);
  initialize_optimization(o, 0);
  compute_active_errors(o);
  std::copy(&o.err[type][0], &o.err[type][0] + dim, e);
  *chi2 = edge_chi2(o, type, 0);
  double J0[kMaxDim * 9] = {0}, J1[kMaxDim * 9] = {0};
  linearize(o, type, 0, J0, J1);
  std::copy(J1, J1 + dim * 6, J);
  build_system(o);
  for (int r = 0; r < 6; ++r)
    for (int c = 0; c < 6; ++c) H[r * 6 + c] = (o.n_red && !fix_pose) ? o.Hpp[(size_t)c * o.n_red + r] : 0.0;
  for (int i = 0; i < 6; ++i) b[i] = (o.n_red && !fix_pose) ? o.b[i] : 0.0;
  return dim;
}

}  // extern "C"
