// ppo_flatten.hpp -- header-only host adapter: a local-BA window in plain structs  ->  ppo_ba_problem  ->  write-back.
//
// This is the part of Optimizer::LocalBACameraPlaneCuboids that stays on the host and sits either side of libppo_ba.so
// (SURVEY.md 8f N2).  It restates, rule by rule, what /root/reference/src/Optimizer.cc does between the window gather and the
// g2o calls, and after them:
//   flatten()     Optimizer.cc:2120-2714  vertices, edges, information matrices, Huber deltas, the reference's selection rules
//   write_back()  Optimizer.cc:2840-2966  erase lists from the per-edge chi2 / depth, float32 narrowing of the estimates
//                                         (Converter.cc:44-71, 183-190)
// It does not touch KeyFrame / MapPoint / MapPlane / MapCuboid (OpenCV / PCL types that cannot be compiled in this repository's
// image): the caller copies what Optimizer.cc:1997-2105 gathers into the plain `Window` below -- a handful of scalar fields per
// object, listed with the reference member each one mirrors -- and applies the returned write-back under Map::mMutexMapUpdate.
// INTEGRATION.md shows that glue.  No dependency beyond <vector>, <cmath> and ppo_ba.h.
//
// Reference behaviours reproduced on purpose, each with the line it follows:
//   * keyframe 0 and every "fixed camera" are fixed (OPT:2126, 2140); bad neighbours never enter (OPT:2008);
//   * a point with Observations() == 1 gets no vertex (OPT:2336) -- and, unlike the reference (OPT:2933 dereferences the missing
//     vertex), is simply left untouched by the write-back;
//   * dense per-type indices replace the colliding graph ids of OPT:2125, 2170, 2214, 2342 (a plane whose id collided with a
//     cuboid's silently lost its vertex in the reference, optimizable_graph.cpp:243-250; here every plane is kept);
//   * an observation by a bad keyframe adds no edge (OPT:2240, 2269, 2293, 2362, 2445, 2506), nor one by a keyframe that has no
//     vertex (plane observers are not collected as fixed cameras: OPT:2242-2243);
//   * bbox / corner edges only when the detection lies 5 px inside a 640 x 480 image (OPT:2456-2460, 2525-2528);
//   * the point-object edge takes the object's unique points seen with it more than twice, drops those farther than 4 m from
//     their mean, then farther than 3 m from the mean of the rest, and exists only with more than 10 survivors (OPT:2561-2642);
//   * Huber deltas are `const float` in the reference (OPT:2202-2205, 2326-2327, 2474, 2542, 2669): rounded through float here;
//   * the cuboid-plane edge is kept only when its cuboid is in the window (OPT:2684-2686).
#ifndef PPO_FLATTEN_HPP
#define PPO_FLATTEN_HPP

#include <cmath>
#include <cstdint>
#include <vector>

#include "ppo_ba.h"

namespace ppo_flatten {

// ---- the window, as plain data ---------------------------------------------------------------------------------------------
struct KeyFrameIn {
  long id = 0;          // KeyFrame::mnId
  bool bad = false;     // KeyFrame::isBad()
  bool local = true;    // member of lLocalKeyFrames (OPT:1999-2009); false = lFixedCameras (OPT:2068-2100)
  float Tcw[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};  // KeyFrame::GetPose(), 4x4 CV_32F row-major
  float fx = 0, fy = 0, cx = 0, cy = 0, bf = 0;                        // KeyFrame::fx .. mbf
  float K[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};                           // KeyFrame::mK (3x3 CV_32F)
};
struct PointObsIn {
  int kf = 0;           // index into Window::keyframes of the observing keyframe, -1 if it is neither local nor fixed
  float u = 0, v = 0;   // mvKeysUn[idx].pt
  float u_right = -1;   // mvuRight[idx]; < 0: monocular observation (OPT:2366)
  float inv_sigma2 = 1; // mvInvLevelSigma2[mvKeysUn[idx].octave] (OPT:2376, 2404)
};
struct MapPointIn {
  long id = 0;
  bool bad = false;           // MapPoint::isBad() at erase time (OPT:2853, 2868); bad points never enter lLocalMapPoints (OPT:2021)
  float xyz[3] = {0, 0, 0};   // MapPoint::GetWorldPos() (3x1 CV_32F)
  int n_observations = 0;     // MapPoint::Observations()
  std::vector<PointObsIn> obs;  // MapPoint::GetObservations(), in its std::map order
};
struct PlaneObsIn {
  int kf = 0;
  float coef[4] = {0, 0, 1, 0};  // pKFi->mvPlaneCoefficients[idx] (4x1 CV_32F)
};
struct MapPlaneIn {
  long id = 0;
  float coef[4] = {0, 0, 1, 0};  // MapPlane::GetWorldPos()
  std::vector<PlaneObsIn> obs, ver_obs, par_obs;  // GetObservations / GetVerObservations / GetParObservations
  long asso_cuboid_id = 999;                      // 999 = none (OPT:2675)
  double asso_cuboid_meas[3] = {0, 0, 0};
};
struct CuboidObsIn {
  int kf = 0;
  double bbox_vec[4] = {0, 0, 0, 0};             // local_object->bbox_vec (centre x, y, width, height)
  int bbox_x = 0, bbox_y = 0, bbox_w = 0, bbox_h = 0;  // local_object->bbox_2d (cv::Rect)
  double corners_2d[16] = {0};                   // local_object->box_corners_2d, interleaved u0 v0 .. u7 v7 (OPT:2514-2519)
  double meas_quality = 1;
  double local_meas_qt[7] = {0, 0, 0, 1, 0, 0, 0};  // local_object->cuboid_local_meas.pose (camera frame; qx qy qz qw tx ty tz) and
  double local_meas_scale[3] = {1, 1, 1};           // .scale: the 3-D measurement of Optimizer::LocalBACameraPointCuboids2D (OPT:1784)
};
struct ObjectPointIn {
  float xyz[3] = {0, 0, 0};  // UniquePoints[j]->GetWorldPos()
  bool bad = false;
  int n_object_observations = 0;  // UniquePoints[j]->MapObjObservations[pMC]
};
struct MapCuboidIn {
  long id = 0;
  double qt[7] = {0, 0, 0, 1, 0, 0, 0};  // cuboid_global_data.pose (qx qy qz qw tx ty tz)
  double scale[3] = {1, 1, 1};           // cuboid_global_data.scale (half extents)
  std::vector<CuboidObsIn> obs;          // MapCuboid::GetObservations() with pKFi->local_cuboids[idx]
  std::vector<ObjectPointIn> unique_points;  // MapCuboid::GetUniqueMapPoints()
};
struct Window {
  std::vector<KeyFrameIn> keyframes;  // local keyframes first or in any order: `local` tells them apart
  std::vector<MapPointIn> points;
  std::vector<MapPlaneIn> planes;
  std::vector<MapCuboidIn> cuboids;
};

// src/Parameters.cc:43-72 (same defaults; the flags are all false there and are switched on by the yaml file) and the arguments of
// Optimizer.h:62
struct Params {
  bool optimize_with_plane_3d = false, optimize_with_cuboid_2d = false, optimize_with_corners_2d = false, optimize_with_pt_obj_3d = false,
       optimize_with_cuboid_plane = false;
  bool optimize_with_cuboid_3d = false;  // the 9-D camera-object edge: only Optimizer::LocalBACameraPointCuboids2D reads it (OPT:1768)
  double ba_weight_SE3 = 1.0, thHuberSE3 = 900.0;   // src/Parameters.cc:60,65
  double cuboid_3d_meas_quality = 0.75;            // the literal of OPT:1787
  double plane_angle_info = 1.0, plane_dist_info = 100.0, plane_chi = 500.0;
  double cuboid_plane_angle_info = 2.0, cuboid_plane_dist_info = 100.0, cuboid_plane_chi = 500.0;
  double ba_weight_bbox = 1.0, ba_weight_corner = 1.0, thHuberBbox2d = 80.0, thHuberConer2d = 10.0;
  bool fix_camera = false, fix_point = false;  // the only call site passes false, false (LocalMapping.cc:100)
  int image_width = 640, image_height = 480, object_boundary_margin = 5;  // literals of OPT:2456-2460
  bool skip_single_observation_points = true;  // OPT:2336-2337; the points-only Optimizer::LocalBundleAdjustment (OPT:581-590) gives every local point a vertex
};

// ---- the flattened problem (owns the arrays a ppo_ba_problem points into) ----------------------------------------------------------
struct Flat {
  ppo_ba_problem problem{};
  std::vector<double> pose_qt, pose_cam, point_xyz, plane_coef, cuboid_qt, cuboid_scale;
  std::vector<uint8_t> pose_fixed, point_fixed, cuboid_flags, pl_kind;
  std::vector<int32_t> mono_point, mono_pose, stereo_point, stereo_pose, pl_plane, pl_pose, bbox_pose, bbox_cuboid, corner_pose, corner_cuboid, ptcub_cuboid,
      ptcub_offsets, cubpl_cuboid, cubpl_plane, cub3d_pose, cub3d_cuboid;
  std::vector<double> cub3d_meas, cub3d_info, cub3d_huber;
  std::vector<double> mono_obs, mono_info, mono_huber, stereo_obs, stereo_info, stereo_huber, pl_meas, pl_info, pl_huber, bbox_meas, bbox_info, bbox_huber,
      bbox_K, corner_meas, corner_info, corner_huber, corner_K, ptcub_points, ptcub_margin_ratio, ptcub_info, cubpl_meas, cubpl_info, cubpl_huber;
  // index maps back to the window (for the write-back)
  std::vector<int> pose_kf;       // problem pose -> Window::keyframes
  std::vector<int> point_src;     // problem point -> Window::points
  std::vector<int> kf_pose;       // Window::keyframes -> problem pose or -1
  std::vector<int> point_vertex;  // Window::points -> problem point or -1 (single-observation points)
  std::vector<int> mono_obs_src, stereo_obs_src;  // edge -> index into MapPointIn::obs of its point
  std::vector<int> pl_obs_src;                    // kind-0 plane edges only: edge -> index into MapPlaneIn::obs; -1 for kinds 1, 2
  std::vector<std::vector<int>> ptcub_used;       // per cuboid: indices into unique_points that survived the filter (used_points_in_BA_filtered)
};

inline double huber_float(double chi) { return (double)(float)std::sqrt(chi); }  // `const float delta = sqrt(chi)`

// Eigen::Quaterniond(Matrix3d) as SE3Quat(R, t) uses it (se3quat.h:58-60; Eigen/src/Geometry/Quaternion.h): Shoemake's branches
inline void rotation_to_quaternion(const double R[9], double q[4]) {
  const double t = R[0] + R[4] + R[8];
  if (t > 0) {
    double s = std::sqrt(t + 1.0);
    q[3] = 0.5 * s;
    s = 0.5 / s;
    q[0] = (R[7] - R[5]) * s;
    q[1] = (R[2] - R[6]) * s;
    q[2] = (R[3] - R[1]) * s;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[4 * i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    double s = std::sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0);
    q[i] = 0.5 * s;
    s = 0.5 / s;
    q[3] = (R[3 * k + j] - R[3 * j + k]) * s;
    q[j] = (R[3 * j + i] + R[3 * i + j]) * s;
    q[k] = (R[3 * k + i] + R[3 * i + k]) * s;
  }
}
// Converter::toSE3Quat (Converter.cc:37-47): float 4x4 -> (q, t); the library normalises and forces w >= 0 on ingest like the SE3Quat constructor
inline void pose_from_cv(const float T[16], double qt[7]) {
  const double R[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
  rotation_to_quaternion(R, qt);
  qt[4] = T[3]; qt[5] = T[7]; qt[6] = T[11];
}

inline void flatten(const Window& w, const Params& par, Flat& f) {
  f = Flat{};
  const int nkf = (int)w.keyframes.size();
  // ---- keyframe vertices: local ones (OPT:2120-2132), then the fixed cameras (OPT:2135-2145) --------------------------------
  f.kf_pose.assign(nkf, -1);
  for (int pass = 0; pass < 2; ++pass)
    for (int k = 0; k < nkf; ++k) {
      const KeyFrameIn& kf = w.keyframes[k];
      if (kf.local != (pass == 0) || kf.bad) continue;  // (a bad neighbour is not pushed at OPT:2008, a bad observer not at OPT:2079)
      f.kf_pose[k] = (int)f.pose_kf.size();
      f.pose_kf.push_back(k);
      double qt[7];
      pose_from_cv(kf.Tcw, qt);
      f.pose_qt.insert(f.pose_qt.end(), qt, qt + 7);
      f.pose_fixed.push_back((uint8_t)(!kf.local || kf.id == 0 || par.fix_camera));
      const double cam[5] = {kf.fx, kf.fy, kf.cx, kf.cy, kf.bf};
      f.pose_cam.insert(f.pose_cam.end(), cam, cam + 5);
    }
  // ---- cuboid vertices (OPT:2150-2179): fix roll / pitch and height --------------------------------------------------------
  for (const MapCuboidIn& c : w.cuboids) {
    f.cuboid_qt.insert(f.cuboid_qt.end(), c.qt, c.qt + 7);
    f.cuboid_scale.insert(f.cuboid_scale.end(), c.scale, c.scale + 3);
    f.cuboid_flags.push_back((uint8_t)(PPO_CUBOID_FIX_ROLLPITCH | PPO_CUBOID_FIX_HEIGHT));
  }
  // ---- plane vertices (OPT:2208-2219) and their edges (OPT:2222-2307) ------------------------------------------------------------
  const double angleInfo = 3282.8 / (par.plane_angle_info * par.plane_angle_info), disInfo = par.plane_dist_info * par.plane_dist_info;
  const double vpInfo = 3282.8 / (0.5 * 0.5);
  const double deltaPlane = huber_float(par.plane_chi), deltaVP = huber_float(200.0);
  for (const MapPlaneIn& p : w.planes)
    for (int a = 0; a < 4; ++a) f.plane_coef.push_back((double)p.coef[a]);  // (sign flip and normalisation of toPlane3D: on ingest)
  if (par.optimize_with_plane_3d)
    for (int pi = 0; pi < (int)w.planes.size(); ++pi) {
      const MapPlaneIn& p = w.planes[pi];
      auto add = [&](const std::vector<PlaneObsIn>& list, int kind) {
        for (int oi = 0; oi < (int)list.size(); ++oi) {
          const PlaneObsIn& o = list[oi];
          if (o.kf < 0 || o.kf >= nkf || w.keyframes[o.kf].bad || f.kf_pose[o.kf] < 0) continue;
          f.pl_plane.push_back(pi);
          f.pl_pose.push_back(f.kf_pose[o.kf]);
          f.pl_kind.push_back((uint8_t)kind);
          for (int a = 0; a < 4; ++a) f.pl_meas.push_back((double)o.coef[a]);
          if (kind == 0) { f.pl_info.push_back(angleInfo); f.pl_info.push_back(angleInfo); f.pl_info.push_back(disInfo); }
          else { f.pl_info.push_back(vpInfo); f.pl_info.push_back(vpInfo); f.pl_info.push_back(0.0); }
          f.pl_huber.push_back(kind == 0 ? deltaPlane : deltaVP);
          f.pl_obs_src.push_back(kind == 0 ? oi : -1);
        }
      };
      add(p.obs, 0);
      add(p.ver_obs, 1);
      add(p.par_obs, 2);
    }
  // ---- point vertices and reprojection edges (OPT:2332-2425) ------------------------------------------------------------------
  const double thMono = huber_float(5.991), thStereo = huber_float(7.815);
  f.point_vertex.assign(w.points.size(), -1);
  for (int pi = 0; pi < (int)w.points.size(); ++pi) {
    const MapPointIn& p = w.points[pi];
    if (par.skip_single_observation_points && p.n_observations == 1) continue;  // OPT:2336-2337
    const int v = (int)f.point_src.size();
    f.point_vertex[pi] = v;
    f.point_src.push_back(pi);
    for (int a = 0; a < 3; ++a) f.point_xyz.push_back((double)p.xyz[a]);
    f.point_fixed.push_back((uint8_t)par.fix_point);
    for (int oi = 0; oi < (int)p.obs.size(); ++oi) {
      const PointObsIn& o = p.obs[oi];
      if (o.kf < 0 || o.kf >= nkf || w.keyframes[o.kf].bad || f.kf_pose[o.kf] < 0) continue;
      if (o.u_right < 0) {
        f.mono_point.push_back(v); f.mono_pose.push_back(f.kf_pose[o.kf]);
        f.mono_obs.push_back((double)o.u); f.mono_obs.push_back((double)o.v);
        f.mono_info.push_back((double)o.inv_sigma2);
        f.mono_huber.push_back(thMono);
        f.mono_obs_src.push_back(oi);
      } else {
        f.stereo_point.push_back(v); f.stereo_pose.push_back(f.kf_pose[o.kf]);
        f.stereo_obs.push_back((double)o.u); f.stereo_obs.push_back((double)o.v); f.stereo_obs.push_back((double)o.u_right);
        f.stereo_info.push_back((double)o.inv_sigma2);
        f.stereo_huber.push_back(thStereo);
        f.stereo_obs_src.push_back(oi);
      }
    }
  }
  // ---- camera-cuboid edges: bounding box (OPT:2433-2487) and corners (OPT:2493-2551) --------------------------------------------
  const int mg = par.object_boundary_margin;
  for (int which = 0; which < 2; ++which) {
    if (which == 0 ? !par.optimize_with_cuboid_2d : !par.optimize_with_corners_2d) continue;
    const double weight = which == 0 ? par.ba_weight_bbox : par.ba_weight_corner;
    const double delta = huber_float(which == 0 ? par.thHuberBbox2d : par.thHuberConer2d);
    for (int ci = 0; ci < (int)w.cuboids.size(); ++ci)
      for (const CuboidObsIn& o : w.cuboids[ci].obs) {
        if (o.kf < 0 || o.kf >= nkf || w.keyframes[o.kf].bad || f.kf_pose[o.kf] < 0) continue;
        if (!((o.bbox_x > mg) && (o.bbox_y > mg) && (o.bbox_x + o.bbox_w < par.image_width - mg) && (o.bbox_y + o.bbox_h < par.image_height - mg))) continue;
        const double is = weight * o.meas_quality, info = is * is;
        std::vector<int32_t>& vp = which == 0 ? f.bbox_pose : f.corner_pose;
        std::vector<int32_t>& vc = which == 0 ? f.bbox_cuboid : f.corner_cuboid;
        std::vector<double>& vm = which == 0 ? f.bbox_meas : f.corner_meas;
        std::vector<double>& vi = which == 0 ? f.bbox_info : f.corner_info;
        std::vector<double>& vh = which == 0 ? f.bbox_huber : f.corner_huber;
        std::vector<double>& vk = which == 0 ? f.bbox_K : f.corner_K;
        vp.push_back(f.kf_pose[o.kf]);
        vc.push_back(ci);
        const int dim = which == 0 ? 4 : 16;
        for (int a = 0; a < dim; ++a) { vm.push_back(which == 0 ? o.bbox_vec[a] : o.corners_2d[a]); vi.push_back(info); }
        vh.push_back(delta);
        for (int a = 0; a < 9; ++a) vk.push_back((double)w.keyframes[o.kf].K[a]);
      }
  }
  // ---- camera-cuboid 3-D measurement edges (EdgeSE3Cuboid, OPT:1767-1800: Optimizer::LocalBACameraPointCuboids2D only) ------------
  // no image-boundary test here; information (ba_weight_SE3 * 0.75)^2 on all nine rows; the kernel delta is thHuberSE3 ITSELF (OPT:1793)
  if (par.optimize_with_cuboid_3d)
    for (int ci = 0; ci < (int)w.cuboids.size(); ++ci)
      for (const CuboidObsIn& o : w.cuboids[ci].obs) {
        if (o.kf < 0 || o.kf >= nkf || w.keyframes[o.kf].bad || f.kf_pose[o.kf] < 0) continue;
        const double is = par.ba_weight_SE3 * par.cuboid_3d_meas_quality, info = is * is;
        f.cub3d_pose.push_back(f.kf_pose[o.kf]);
        f.cub3d_cuboid.push_back(ci);
        for (int a = 0; a < 7; ++a) f.cub3d_meas.push_back(o.local_meas_qt[a]);
        for (int a = 0; a < 3; ++a) f.cub3d_meas.push_back(o.local_meas_scale[a]);
        for (int a = 0; a < 9; ++a) f.cub3d_info.push_back(info);
        f.cub3d_huber.push_back(par.thHuberSE3);
      }
  // ---- point-object edges (OPT:2556-2655) ------------------------------------------------------------------------------------
  f.ptcub_used.assign(w.cuboids.size(), std::vector<int>());
  f.ptcub_offsets.push_back(0);
  if (par.optimize_with_pt_obj_3d)
    for (int ci = 0; ci < (int)w.cuboids.size(); ++ci) {
      const MapCuboidIn& c = w.cuboids[ci];
      std::vector<int> cand;
      for (int j = 0; j < (int)c.unique_points.size(); ++j)
        if (!c.unique_points[j].bad && c.unique_points[j].n_object_observations > 2) cand.push_back(j);  // point_object_threshold = 2
      auto P = [&](int j, int a) { return (double)c.unique_points[j].xyz[a]; };
      auto dist = [&](const double m[3], int j) {
        return std::sqrt((m[0] - P(j, 0)) * (m[0] - P(j, 0)) + (m[1] - P(j, 1)) * (m[1] - P(j, 1)) + (m[2] - P(j, 2)) * (m[2] - P(j, 2)));
      };
      double mean[3] = {0, 0, 0};
      for (int j : cand) for (int a = 0; a < 3; ++a) mean[a] += P(j, a);
      for (int a = 0; a < 3; ++a) mean[a] /= (double)cand.size();  // (0 / 0 = NaN with no candidates, as in the reference: nothing passes)
      double mean2[3] = {0, 0, 0};
      int nvalid = 0;
      for (int j : cand)
        if (dist(mean, j) < 4.0) { for (int a = 0; a < 3; ++a) mean2[a] += P(j, a); ++nvalid; }  // coarse_threshold
      for (int a = 0; a < 3; ++a) mean2[a] /= (double)nvalid;
      std::vector<int> good;
      for (int j : cand)
        if (dist(mean2, j) < 3.0) good.push_back(j);  // fine_threshold
      f.ptcub_used[ci] = good;
      if (good.size() > 10) {  // OPT:2642
        f.ptcub_cuboid.push_back(ci);
        for (int j : good) for (int a = 0; a < 3; ++a) f.ptcub_points.push_back(P(j, a));
        f.ptcub_offsets.push_back((int32_t)(f.ptcub_points.size() / 3));
        f.ptcub_margin_ratio.push_back(1.0);
        for (int a = 0; a < 3; ++a) f.ptcub_info.push_back(1.0);
      }
    }
  // ---- cuboid-plane edges (OPT:2662-2714) ------------------------------------------------------------------------------------------
  if (par.optimize_with_cuboid_plane) {
    const double a_cp = 3282.8 / (par.cuboid_plane_angle_info * par.cuboid_plane_angle_info), b_cp = par.cuboid_plane_dist_info * par.cuboid_plane_dist_info;
    const double delta_cp = huber_float(par.cuboid_plane_chi);
    for (int pi = 0; pi < (int)w.planes.size(); ++pi) {
      const MapPlaneIn& p = w.planes[pi];
      if (p.asso_cuboid_id == 999) continue;
      int ci = -1;
      for (int c = 0; c < (int)w.cuboids.size(); ++c)
        if (w.cuboids[c].id == p.asso_cuboid_id) { ci = c; break; }
      if (ci < 0) continue;  // the cuboid is not in the window (OPT:2684-2686)
      f.cubpl_cuboid.push_back(ci); f.cubpl_plane.push_back(pi);
      for (int a = 0; a < 3; ++a) f.cubpl_meas.push_back(p.asso_cuboid_meas[a]);
      f.cubpl_info.push_back(a_cp); f.cubpl_info.push_back(a_cp); f.cubpl_info.push_back(b_cp);
      f.cubpl_huber.push_back(delta_cp);
    }
  }
  // ---- the C struct ------------------------------------------------------------------------------------------------------------------
  ppo_ba_problem& q = f.problem;
  auto dp = [](const std::vector<double>& v) { return v.empty() ? nullptr : v.data(); };
  auto ip = [](const std::vector<int32_t>& v) { return v.empty() ? nullptr : v.data(); };
  auto bp = [](const std::vector<uint8_t>& v) { return v.empty() ? nullptr : v.data(); };
  q.n_poses = (int32_t)f.pose_kf.size(); q.pose_qt = dp(f.pose_qt); q.pose_fixed = bp(f.pose_fixed); q.pose_cam = dp(f.pose_cam);
  q.n_points = (int32_t)f.point_src.size(); q.point_xyz = dp(f.point_xyz); q.point_fixed = bp(f.point_fixed);
  q.n_planes = (int32_t)w.planes.size(); q.plane_coef = dp(f.plane_coef);
  q.n_cuboids = (int32_t)w.cuboids.size(); q.cuboid_qt = dp(f.cuboid_qt); q.cuboid_scale = dp(f.cuboid_scale); q.cuboid_flags = bp(f.cuboid_flags);
  q.n_mono = (int32_t)f.mono_point.size(); q.mono_point = ip(f.mono_point); q.mono_pose = ip(f.mono_pose); q.mono_obs = dp(f.mono_obs);
  q.mono_info = dp(f.mono_info); q.mono_huber = dp(f.mono_huber);
  q.n_stereo = (int32_t)f.stereo_point.size(); q.stereo_point = ip(f.stereo_point); q.stereo_pose = ip(f.stereo_pose); q.stereo_obs = dp(f.stereo_obs);
  q.stereo_info = dp(f.stereo_info); q.stereo_huber = dp(f.stereo_huber);
  q.n_plane_edges = (int32_t)f.pl_plane.size(); q.pl_plane = ip(f.pl_plane); q.pl_pose = ip(f.pl_pose); q.pl_kind = bp(f.pl_kind); q.pl_meas = dp(f.pl_meas);
  q.pl_info = dp(f.pl_info); q.pl_huber = dp(f.pl_huber);
  q.n_bbox = (int32_t)f.bbox_pose.size(); q.bbox_pose = ip(f.bbox_pose); q.bbox_cuboid = ip(f.bbox_cuboid); q.bbox_meas = dp(f.bbox_meas);
  q.bbox_info = dp(f.bbox_info); q.bbox_huber = dp(f.bbox_huber); q.bbox_K = dp(f.bbox_K);
  q.n_corner = (int32_t)f.corner_pose.size(); q.corner_pose = ip(f.corner_pose); q.corner_cuboid = ip(f.corner_cuboid); q.corner_meas = dp(f.corner_meas);
  q.corner_info = dp(f.corner_info); q.corner_huber = dp(f.corner_huber); q.corner_K = dp(f.corner_K);
  q.n_ptcub = (int32_t)f.ptcub_cuboid.size(); q.ptcub_cuboid = ip(f.ptcub_cuboid); q.ptcub_offsets = f.ptcub_offsets.data(); q.ptcub_points = dp(f.ptcub_points);
  q.ptcub_margin_ratio = dp(f.ptcub_margin_ratio); q.ptcub_info = dp(f.ptcub_info);
  q.n_cubpl = (int32_t)f.cubpl_cuboid.size(); q.cubpl_cuboid = ip(f.cubpl_cuboid); q.cubpl_plane = ip(f.cubpl_plane); q.cubpl_meas = dp(f.cubpl_meas);
  q.cubpl_info = dp(f.cubpl_info); q.cubpl_huber = dp(f.cubpl_huber);
  q.n_cub3d = (int32_t)f.cub3d_pose.size(); q.cub3d_pose = ip(f.cub3d_pose); q.cub3d_cuboid = ip(f.cub3d_cuboid); q.cub3d_meas = dp(f.cub3d_meas);
  q.cub3d_info = dp(f.cub3d_info); q.cub3d_huber = dp(f.cub3d_huber);
}

// ---- write-back (OPT:2840-2966) --------------------------------------------------------------------------------------------------
struct EraseObs { int object, keyframe; };  // indices into Window::points / planes and Window::keyframes
struct WriteBack {
  std::vector<EraseObs> erase_points;  // vToErase: pKFi->EraseMapPointMatch(pMP); pMP->EraseObservation(pKFi)
  std::vector<EraseObs> erase_planes;  // vToErasePlane
  struct Pose { int keyframe; float Tcw[16]; };        // KeyFrame::SetPose(Converter::toCvMat(SE3quat)), local keyframes only (OPT:2917-2924)
  struct Point { int point; float xyz[3]; };           // MapPoint::SetWorldPos + UpdateNormalAndDepth (OPT:2927-2936)
  struct CuboidOut { int cuboid; double qt[7], scale[3]; float Twc[16]; };  // cuboid_global_opti (double) and SetWorldPos (float 4x4) (OPT:2947-2957)
  struct PlaneOut { int plane; float coef[4]; };       // MapPlane::SetWorldPos(Converter::toCvMat(Plane3D)) (OPT:2960-2966)
  std::vector<Pose> poses;
  std::vector<Point> points;
  std::vector<CuboidOut> cuboids;
  std::vector<PlaneOut> planes;
};

// SE3Quat::to_homogeneous_matrix narrowed by Converter::toCvMat (Converter.cc:49-71)
inline void pose_to_cv(const double qt[7], float T[16]) {
  const double x = qt[0], y = qt[1], z = qt[2], w = qt[3];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z, twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
  const double R[9] = {1 - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1 - (txx + tzz), tyz - twx, txz - twy, tyz + twx, 1 - (txx + tyy)};
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) T[4 * r + c] = (float)R[3 * r + c];
    T[4 * r + 3] = (float)qt[4 + r];
  }
  T[12] = 0; T[13] = 0; T[14] = 0; T[15] = 1;
}

// `r` must carry pose_qt, point_xyz, plane_coef, cuboid_qt, cuboid_scale and, for the erase lists, mono_chi2 / mono_depth_positive,
// stereo_chi2 / stereo_depth_positive and pl_chi2 (any of them may be NULL: that part is skipped).
inline void write_back(const Window& w, const Params& par, const Flat& f, const ppo_ba_result& r, WriteBack& out) {
  out = WriteBack{};
  const ppo_ba_problem& q = f.problem;
  // erase lists: the same tests as after the first pass, on the final errors (OPT:2848-2887)
  if (r.mono_chi2 && r.mono_depth_positive)
    for (int e = 0; e < q.n_mono; ++e) {
      const int pi = f.point_src[q.mono_point[e]];
      if (w.points[pi].bad) continue;
      if (r.mono_chi2[e] > 5.991 || !r.mono_depth_positive[e]) out.erase_points.push_back(EraseObs{pi, f.pose_kf[q.mono_pose[e]]});
    }
  if (r.stereo_chi2 && r.stereo_depth_positive)
    for (int e = 0; e < q.n_stereo; ++e) {
      const int pi = f.point_src[q.stereo_point[e]];
      if (w.points[pi].bad) continue;
      if (r.stereo_chi2[e] > 7.815 || !r.stereo_depth_positive[e]) out.erase_points.push_back(EraseObs{pi, f.pose_kf[q.stereo_pose[e]]});
    }
  if (r.pl_chi2)
    for (int e = 0; e < q.n_plane_edges; ++e)
      if (f.pl_kind[e] == 0 && r.pl_chi2[e] > par.plane_chi) out.erase_planes.push_back(EraseObs{q.pl_plane[e], f.pose_kf[q.pl_pose[e]]});
  // estimates, narrowed to float exactly where the reference narrows them
  if (r.pose_qt)
    for (int p = 0; p < q.n_poses; ++p) {
      const int k = f.pose_kf[p];
      if (!w.keyframes[k].local) continue;  // fixed cameras only have their markers reset (OPT:2939-2944)
      WriteBack::Pose o;
      o.keyframe = k;
      pose_to_cv(&r.pose_qt[7 * p], o.Tcw);
      out.poses.push_back(o);
    }
  if (r.point_xyz)
    for (int v = 0; v < q.n_points; ++v) {
      WriteBack::Point o;
      o.point = f.point_src[v];
      for (int a = 0; a < 3; ++a) o.xyz[a] = (float)r.point_xyz[3 * v + a];
      out.points.push_back(o);
    }
  if (r.cuboid_qt && r.cuboid_scale)
    for (int c = 0; c < q.n_cuboids; ++c) {
      WriteBack::CuboidOut o;
      o.cuboid = c;
      for (int a = 0; a < 7; ++a) o.qt[a] = r.cuboid_qt[7 * c + a];
      for (int a = 0; a < 3; ++a) o.scale[a] = r.cuboid_scale[3 * c + a];
      pose_to_cv(o.qt, o.Twc);
      out.cuboids.push_back(o);
    }
  if (r.plane_coef)
    for (int p = 0; p < q.n_planes; ++p) {
      WriteBack::PlaneOut o;
      o.plane = p;
      for (int a = 0; a < 4; ++a) o.coef[a] = (float)r.plane_coef[4 * p + a];  // (the library returns (n, d) with |n| = 1 and d >= 0, as Plane3D::toVector)
      out.planes.push_back(o);
    }
}

}  // namespace ppo_flatten

#endif
