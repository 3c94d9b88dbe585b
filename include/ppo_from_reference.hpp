// ppo_from_reference.hpp -- the reference-side half of the drop-in: Optimizer::LocalBACameraPlaneCuboids written over the
// REFERENCE'S OWN member names, as templates.
//
// include/ppo_flatten.hpp holds the construction / write-back RULES over plain structs.  What was left to the maintainer -- the
// window gather (/root/reference/src/Optimizer.cc:1997-2105), the copy loops over KeyFrame / MapPoint / MapPlane / MapCuboid and the
// application of the write-back under Map::mMutexMapUpdate (OPT:2892-2966) -- lives here, as function templates whose bodies use
// exactly the members the reference declares:
//   KeyFrame   mnId, mnBALocalForKF, mnBAFixedForKF, isBad(), GetVectorCovisibleKeyFrames(), GetMapPointMatches(), mvpMapCuboid,
//              mvpMapPlanes, GetPose(), SetPose(), fx, fy, cx, cy, mbf, mK, mvKeysUn, mvuRight, mvInvLevelSigma2, local_cuboids,
//              mvPlaneCoefficients, EraseMapPointMatch(MapPoint*), EraseMapPlaneMatch(MapPlane*)          (include/KeyFrame.h:76-242)
//   MapPoint   mnId, mnBALocalForKF, isBad(), Observations(), GetObservations(), GetWorldPos(), SetWorldPos(), UpdateNormalAndDepth(),
//              EraseObservation(KeyFrame*), MapObjObservations                                           (include/MapPoint.h)
//   MapPlane   mnId, mnBALocalForKF, isBad(), GetWorldPos(), SetWorldPos(), GetObservations(), GetVerObservations(),
//              GetParObservations(), EraseObservation(KeyFrame*), asso_cuboid_id, asso_cuboid_meas       (include/MapPlane.h:27-92)
//   MapCuboid  mnId, mnBALocalForKF, isBad(), GetObservations(), GetUniqueMapPoints(), cuboid_global_data, cuboid_global_opti,
//              obj_been_optimized, used_points_in_BA_filtered, SetWorldPos(), bbox_vec, bbox_2d, box_corners_2d, meas_quality
//                                                                                                        (include/MapCuboid.h:35-174)
//   Map        mMutexMapUpdate                                                                           (include/Map.h)
// The OpenCV / Eigen / g2o VALUE types are reached through their own member syntax too (`m.at<float>(r, c)`, `q.x()`, `v[i]`,
// `rect.width`, `corners(r, c)`); the two things a template cannot spell without the headers -- making a CV_32F matrix and a
// g2o::cuboid -- come from a small traits struct (CvTraits below is the one for the reference; it compiles only where OpenCV and
// the reference's headers are present, hence the guard).
//
// Inside the reference the whole function body becomes
//     #include "ppo_from_reference.hpp"
//     void Optimizer::LocalBACameraPlaneCuboids(KeyFrame* pKF, bool* pbStopFlag, Map* pMap, bool fixCamera, bool fixPoint) {
//       ppo_from_reference::local_ba_camera_plane_cuboids<ppo_from_reference::CvTraits>(pKF, pbStopFlag, pMap, fixCamera, fixPoint,
//                                                                                        ppo_from_reference::params_from_globals());
//     }
// (INTEGRATION.md).  This repository's image has none of OpenCV / Eigen / PCL, so the templates are instantiated in
// tests/from_reference_test.cpp on mock classes that carry exactly those member names: a typo, a wrong lock order or a forgotten
// marker reset fails a test here instead of on the maintainer's machine.
#ifndef PPO_FROM_REFERENCE_HPP
#define PPO_FROM_REFERENCE_HPP

#include <cstring>
#include <iostream>
#include <map>
#include <mutex>
#include <vector>

#include "ppo_ba.h"
#include "ppo_flatten.hpp"
#include "ppo_pose.h"

namespace ppo_from_reference {

namespace pf = ppo_flatten;

// ---- OPT:1997-2100: the window of one local BA ------------------------------------------------------------------------------
template <class KeyFrameT, class MapPointT, class MapPlaneT, class MapCuboidT>
struct Gathered {
  std::vector<KeyFrameT*> lLocalKeyFrames, lFixedCameras;
  std::vector<MapPointT*> lLocalMapPoints;
  std::vector<MapCuboidT*> lLocalMapCuboids;
  std::vector<MapPlaneT*> lLocalMapPlanes;
};

// Breadth-first gather with the reference's markers (mnBALocalForKF / mnBAFixedForKF = pKF->mnId): local keyframes = pKF + its
// covisible keyframes that are not bad (OPT:1999-2009; a bad neighbour is MARKED but not pushed); their map points / cuboids / planes
// that exist, are not bad and are not marked yet (OPT:2012-2064); fixed cameras = the other observers of the local points and of the
// local cuboids (OPT:2068-2100; plane observers are NOT collected) -- marked first, pushed only if not bad.
// with_planes = false: the gather of Optimizer::LocalBACameraPointCuboids2D (OPT:1255-1343), which never looks at the planes.
template <class MapPointT, class MapPlaneT, class MapCuboidT, class KeyFrameT>
Gathered<KeyFrameT, MapPointT, MapPlaneT, MapCuboidT> gather_window(KeyFrameT* pKF, bool with_planes = true) {
  Gathered<KeyFrameT, MapPointT, MapPlaneT, MapCuboidT> g;
  g.lLocalKeyFrames.push_back(pKF);
  pKF->mnBALocalForKF = pKF->mnId;
  const std::vector<KeyFrameT*> vNeighKFs = pKF->GetVectorCovisibleKeyFrames();
  for (KeyFrameT* pKFi : vNeighKFs) {
    pKFi->mnBALocalForKF = pKF->mnId;
    if (!pKFi->isBad()) g.lLocalKeyFrames.push_back(pKFi);
  }
  for (KeyFrameT* pKFl : g.lLocalKeyFrames) {
    const std::vector<MapPointT*> vpMPs = pKFl->GetMapPointMatches();
    for (MapPointT* pMP : vpMPs)
      if (pMP && !pMP->isBad() && pMP->mnBALocalForKF != pKF->mnId) {
        g.lLocalMapPoints.push_back(pMP);
        pMP->mnBALocalForKF = pKF->mnId;
      }
  }
  for (KeyFrameT* pKFl : g.lLocalKeyFrames) {
    const std::vector<MapCuboidT*> vpMCs = pKFl->mvpMapCuboid;
    for (MapCuboidT* pMC : vpMCs)
      if (pMC && !pMC->isBad() && pMC->mnBALocalForKF != (unsigned long)pKF->mnId) {
        g.lLocalMapCuboids.push_back(pMC);
        pMC->mnBALocalForKF = pKF->mnId;
      }
  }
  if (with_planes)
  for (KeyFrameT* pKFl : g.lLocalKeyFrames) {
    const std::vector<MapPlaneT*> vpMPs = pKFl->mvpMapPlanes;
    for (MapPlaneT* pMP : vpMPs)
      if (pMP && !pMP->isBad() && pMP->mnBALocalForKF != pKF->mnId) {
        g.lLocalMapPlanes.push_back(pMP);
        pMP->mnBALocalForKF = pKF->mnId;
      }
  }
  auto consider_fixed = [&](KeyFrameT* pKFi) {
    if (pKFi->mnBALocalForKF != pKF->mnId && pKFi->mnBAFixedForKF != pKF->mnId) {
      pKFi->mnBAFixedForKF = pKF->mnId;
      if (!pKFi->isBad()) g.lFixedCameras.push_back(pKFi);
    }
  };
  for (MapPointT* pMP : g.lLocalMapPoints) {
    const auto observations = pMP->GetObservations();  // std::map<KeyFrame*, size_t>
    for (const auto& mit : observations) consider_fixed(mit.first);
  }
  for (MapCuboidT* pMC : g.lLocalMapCuboids) {
    const auto observations = pMC->GetObservations();  // std::unordered_map<KeyFrame*, size_t>
    for (const auto& mit : observations) consider_fixed(mit.first);
  }
  return g;
}

// ---- the copy loops: reference objects -> ppo_flatten::Window ---------------------------------------------------------------------
// kfs[i] is the keyframe behind Window::keyframes[i] (local keyframes first, then the fixed cameras).
template <class KeyFrameT, class MapPointT, class MapPlaneT, class MapCuboidT>
void to_window(const Gathered<KeyFrameT, MapPointT, MapPlaneT, MapCuboidT>& g, pf::Window& w, std::vector<KeyFrameT*>& kfs) {
  w = pf::Window{};
  kfs.clear();
  std::map<const KeyFrameT*, int> kf_index;
  auto add_kf = [&](KeyFrameT* k, bool local) {
    pf::KeyFrameIn in;
    in.id = (long)k->mnId;
    in.bad = k->isBad();
    in.local = local;
    const auto T = k->GetPose();  // 4x4 CV_32F, world -> camera
    for (int r = 0; r < 4; ++r)
      for (int c = 0; c < 4; ++c) in.Tcw[4 * r + c] = T.template at<float>(r, c);
    in.fx = k->fx; in.fy = k->fy; in.cx = k->cx; in.cy = k->cy; in.bf = k->mbf;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) in.K[3 * r + c] = k->mK.template at<float>(r, c);
    kf_index[k] = (int)w.keyframes.size();
    kfs.push_back(k);
    w.keyframes.push_back(in);
  };
  for (KeyFrameT* k : g.lLocalKeyFrames) add_kf(k, true);
  for (KeyFrameT* k : g.lFixedCameras) add_kf(k, false);
  auto kf_of = [&](const KeyFrameT* k) {
    const auto it = kf_index.find(k);
    return it == kf_index.end() ? -1 : it->second;  // a bad observer: marked but never pushed, it has no vertex (OPT:2079, 2362)
  };
  for (MapPointT* pMP : g.lLocalMapPoints) {
    pf::MapPointIn p;
    p.id = (long)pMP->mnId;
    p.bad = pMP->isBad();
    p.n_observations = pMP->Observations();
    const auto X = pMP->GetWorldPos();  // 3x1 CV_32F
    for (int a = 0; a < 3; ++a) p.xyz[a] = X.template at<float>(a);
    const auto observations = pMP->GetObservations();
    for (const auto& ob : observations) {
      const KeyFrameT* pKFi = ob.first;
      pf::PointObsIn o;
      o.kf = kf_of(pKFi);
      const auto& kpUn = pKFi->mvKeysUn[ob.second];
      o.u = kpUn.pt.x;
      o.v = kpUn.pt.y;
      o.u_right = pKFi->mvuRight[ob.second];
      o.inv_sigma2 = pKFi->mvInvLevelSigma2[kpUn.octave];
      p.obs.push_back(o);
    }
    w.points.push_back(p);
  }
  for (MapPlaneT* pMP : g.lLocalMapPlanes) {
    pf::MapPlaneIn p;
    p.id = (long)pMP->mnId;
    const auto c = pMP->GetWorldPos();  // 4x1 CV_32F
    for (int a = 0; a < 4; ++a) p.coef[a] = c.template at<float>(a);
    auto copy = [&](const auto& observations, std::vector<pf::PlaneObsIn>& out) {  // std::map<KeyFrame*, int>
      for (const auto& ob : observations) {
        pf::PlaneObsIn o;
        o.kf = kf_of(ob.first);
        const auto& pc = ob.first->mvPlaneCoefficients[ob.second];
        for (int a = 0; a < 4; ++a) o.coef[a] = pc.template at<float>(a);
        out.push_back(o);
      }
    };
    copy(pMP->GetObservations(), p.obs);
    copy(pMP->GetVerObservations(), p.ver_obs);
    copy(pMP->GetParObservations(), p.par_obs);
    p.asso_cuboid_id = (long)pMP->asso_cuboid_id;
    for (int a = 0; a < 3; ++a) p.asso_cuboid_meas[a] = pMP->asso_cuboid_meas[a];
    w.planes.push_back(p);
  }
  for (MapCuboidT* pMC : g.lLocalMapCuboids) {
    pf::MapCuboidIn c;
    c.id = (long)pMC->mnId;
    const auto& gd = pMC->cuboid_global_data;  // g2o::cuboid: SE3Quat pose (object -> world) + Vector3d scale
    const auto& q = gd.pose.rotation();
    const auto& t = gd.pose.translation();
    c.qt[0] = q.x(); c.qt[1] = q.y(); c.qt[2] = q.z(); c.qt[3] = q.w();
    c.qt[4] = t[0]; c.qt[5] = t[1]; c.qt[6] = t[2];
    for (int a = 0; a < 3; ++a) c.scale[a] = gd.scale[a];
    const auto observations = pMC->GetObservations();
    for (const auto& ob : observations) {
      const auto* lo = ob.first->local_cuboids[ob.second];  // the detection in that keyframe (OPT:2449)
      pf::CuboidObsIn o;
      o.kf = kf_of(ob.first);
      for (int a = 0; a < 4; ++a) o.bbox_vec[a] = lo->bbox_vec[a];
      o.bbox_x = lo->bbox_2d.x; o.bbox_y = lo->bbox_2d.y; o.bbox_w = lo->bbox_2d.width; o.bbox_h = lo->bbox_2d.height;
      for (int k = 0; k < 8; ++k) {
        o.corners_2d[2 * k] = lo->box_corners_2d(0, k);
        o.corners_2d[2 * k + 1] = lo->box_corners_2d(1, k);
      }
      o.meas_quality = lo->meas_quality;
      {  // the 3-D measurement of the detection (camera frame), read by the 9-D edge of LocalBACameraPointCuboids2D only (OPT:1784)
        const auto& lm = lo->cuboid_local_meas;
        const auto& lq = lm.pose.rotation();
        const auto& lt = lm.pose.translation();
        o.local_meas_qt[0] = lq.x(); o.local_meas_qt[1] = lq.y(); o.local_meas_qt[2] = lq.z(); o.local_meas_qt[3] = lq.w();
        for (int a = 0; a < 3; ++a) { o.local_meas_qt[4 + a] = lt[a]; o.local_meas_scale[a] = lm.scale[a]; }
      }
      c.obs.push_back(o);
    }
    const std::vector<MapPointT*> UniquePoints = pMC->GetUniqueMapPoints();
    for (MapPointT* up : UniquePoints) {
      pf::ObjectPointIn op;
      if (!up) {  // OPT:2572 `if (UniquePoints[j])`: keeps the index aligned with GetUniqueMapPoints() for used_points_in_BA_filtered
        op.bad = true;
        c.unique_points.push_back(op);
        continue;
      }
      op.bad = up->isBad();
      // (find(), not operator[]: the reference reads MapObjObservations[pMC] only for non-bad points inside the point-object branch
      //  (OPT:2571-2574), where operator[] may insert; here nothing is ever inserted into MapPoint state other threads share)
      if (!op.bad) {
        const auto it = up->MapObjObservations.find(pMC);
        op.n_object_observations = it == up->MapObjObservations.end() ? 0 : (int)it->second;
      }
      const auto X = up->GetWorldPos();
      for (int a = 0; a < 3; ++a) op.xyz[a] = X.template at<float>(a);
      c.unique_points.push_back(op);
    }
    w.cuboids.push_back(c);
  }
}

// OPT:2587-2616 side effect of the graph construction: MapCuboid::used_points_in_BA_filtered = the object points that survived the
// two distance filters (the viewer reads it).  No lock, as in the reference.
template <class KeyFrameT, class MapPointT, class MapPlaneT, class MapCuboidT>
void store_used_points(const Gathered<KeyFrameT, MapPointT, MapPlaneT, MapCuboidT>& g, const pf::Flat& f) {
  for (size_t i = 0; i < g.lLocalMapCuboids.size() && i < f.ptcub_used.size(); ++i) {
    MapCuboidT* pMObj = g.lLocalMapCuboids[i];
    const std::vector<MapPointT*> UniquePoints = pMObj->GetUniqueMapPoints();
    pMObj->used_points_in_BA_filtered.clear();
    for (int j : f.ptcub_used[i]) pMObj->used_points_in_BA_filtered.push_back(UniquePoints[j]);
  }
}

// ---- OPT:2892-2966: everything that touches the map, under Map::mMutexMapUpdate ------------------------------------------------------
// Traits::make_mat(rows, cols) -> a CV_32F matrix; Traits::make_cuboid(qt, scale) -> a g2o::cuboid.
template <class Traits, class KeyFrameT, class MapPointT, class MapPlaneT, class MapCuboidT, class MapT>
void apply_write_back(const Gathered<KeyFrameT, MapPointT, MapPlaneT, MapCuboidT>& g, const std::vector<KeyFrameT*>& kfs, const pf::WriteBack& wb, MapT* pMap,
                      bool reset_markers = true) {
  std::unique_lock<decltype(pMap->mMutexMapUpdate)> lock(pMap->mMutexMapUpdate);  // OPT:2892 (points-only local BA: OPT:748)
  for (const pf::EraseObs& e : wb.erase_points) {  // OPT:2894-2903
    KeyFrameT* pKFi = kfs[e.keyframe];
    MapPointT* pMPi = g.lLocalMapPoints[e.object];
    pKFi->EraseMapPointMatch(pMPi);
    pMPi->EraseObservation(pKFi);
  }
  for (const pf::EraseObs& e : wb.erase_planes) {  // OPT:2905-2913
    KeyFrameT* pKFi = kfs[e.keyframe];
    MapPlaneT* pMPi = g.lLocalMapPlanes[e.object];
    pKFi->EraseMapPlaneMatch(pMPi);
    pMPi->EraseObservation(pKFi);
  }
  // keyframes (OPT:2917-2924): marker reset for EVERY local keyframe, pose for those that have a vertex
  if (reset_markers)
    for (KeyFrameT* pKFrame : g.lLocalKeyFrames) pKFrame->mnBALocalForKF = 0;
  for (const pf::WriteBack::Pose& o : wb.poses) {
    auto T = Traits::make_mat(4, 4);
    for (int r = 0; r < 4; ++r)
      for (int c = 0; c < 4; ++c) T.template at<float>(r, c) = o.Tcw[4 * r + c];
    kfs[o.keyframe]->SetPose(T);
  }
  // points (OPT:2927-2936)
  if (reset_markers)
    for (MapPointT* pMP : g.lLocalMapPoints) pMP->mnBALocalForKF = 0;
  for (const pf::WriteBack::Point& o : wb.points) {
    auto X = Traits::make_mat(3, 1);
    for (int a = 0; a < 3; ++a) X.template at<float>(a) = o.xyz[a];
    MapPointT* pMP = g.lLocalMapPoints[o.point];
    pMP->SetWorldPos(X);
    pMP->UpdateNormalAndDepth();
  }
  // fixed cameras (OPT:2939-2944): markers only
  if (reset_markers)
    for (KeyFrameT* pKFrame : g.lFixedCameras) {
      pKFrame->mnBAFixedForKF = 0;
      pKFrame->mnBALocalForKF = 0;
    }
  // cuboids (OPT:2947-2957)
  if (reset_markers)
    for (MapCuboidT* pMCuboid : g.lLocalMapCuboids) pMCuboid->mnBALocalForKF = 0;
  for (const pf::WriteBack::CuboidOut& o : wb.cuboids) {
    MapCuboidT* pMCuboid = g.lLocalMapCuboids[o.cuboid];
    pMCuboid->obj_been_optimized = true;
    auto Twc = Traits::make_mat(4, 4);
    for (int r = 0; r < 4; ++r)
      for (int c = 0; c < 4; ++c) Twc.template at<float>(r, c) = o.Twc[4 * r + c];
    pMCuboid->SetWorldPos(Twc);
    pMCuboid->cuboid_global_opti = Traits::make_cuboid(o.qt, o.scale);
  }
  // planes (OPT:2960-2966); the reference leaves MapPlane::mnBALocalForKF set
  for (const pf::WriteBack::PlaneOut& o : wb.planes) {
    auto c = Traits::make_mat(4, 1);
    for (int a = 0; a < 4; ++a) c.template at<float>(a) = o.coef[a];
    g.lLocalMapPlanes[o.plane]->SetWorldPos(c);
  }
}

// ---- the g2o section (OPT:2108-2116, 2727-2838) through the C ABI ---------------------------------------------------------------------
// Owns the result buffers a ppo_ba_result points into.
struct Solved {
  int rc = PPO_BA_OK;  // of the last library call that ran; > 0: ended early (PPO_BA_ABORTED), results are still valid
  ppo_ba_stats stats{};
  ppo_ba_result result{};
  std::vector<double> pose, pts, pln, cq, cs, mchi, schi, pchi;
  std::vector<uint8_t> mdp, sdp;
  void size_for(const ppo_ba_problem& P) {
    pose.assign(7 * (size_t)P.n_poses, 0.0); pts.assign(3 * (size_t)P.n_points, 0.0); pln.assign(4 * (size_t)P.n_planes, 0.0);
    cq.assign(7 * (size_t)P.n_cuboids, 0.0); cs.assign(3 * (size_t)P.n_cuboids, 0.0);
    mchi.assign(P.n_mono, 0.0); schi.assign(P.n_stereo, 0.0); pchi.assign(P.n_plane_edges, 0.0);
    mdp.assign(P.n_mono, 0); sdp.assign(P.n_stereo, 0);
    result = ppo_ba_result{};
    result.pose_qt = pose.data(); result.point_xyz = pts.data(); result.plane_coef = pln.data(); result.cuboid_qt = cq.data(); result.cuboid_scale = cs.data();
    result.mono_chi2 = mchi.data(); result.mono_depth_positive = mdp.data(); result.stereo_chi2 = schi.data(); result.stereo_depth_positive = sdp.data();
    result.pl_chi2 = pchi.data();
  }
};

// create / run / read / destroy on libppo_ba.so.  PPO_BA_ERR_HIP when there is no usable GPU: there is no CPU fallback.
struct LibrarySolver {
  int operator()(const ppo_ba_problem& P, const ppo_ba_options& O, const volatile uint8_t* stop, Solved& s) const {
    s.size_for(P);
    ppo_ba_handle h = nullptr;
    s.rc = ppo_ba_create(&P, &O, &h);
    if (s.rc != PPO_BA_OK) return s.rc;
    s.rc = ppo_ba_run(h, stop, &s.stats);
    if (s.rc >= 0) {
      const int rr = ppo_ba_read(h, &s.result);
      if (rr != PPO_BA_OK) s.rc = rr;
    }
    ppo_ba_destroy(h);
    return s.rc;
  }
};

inline void options_from_params(const pf::Params& par, ppo_ba_options& O) {
  ppo_ba_default_options(&O);               // 5 + 10 iterations, chi2 5.991 / 7.815 / 200, delta = 1e-9 (OPT:2728, 2837, 2326-2327)
  O.chi2_plane = par.plane_chi;             // OPT:2791
  O.th_bbox_norm = par.thHuberBbox2d;       // OPT:2774
  O.th_corner_norm = par.thHuberConer2d;    // OPT:2783
  O.th_cub3d_norm = par.thHuberSE3;         // OPT:1878
}

// ---- the whole of Optimizer::LocalBACameraPlaneCuboids (include/Optimizer.h:62) ------------------------------------------------------
// Returns the library's code (PPO_BA_OK, PPO_BA_ABORTED, or an error: the map is then left untouched except for the BA markers,
// which are reset).  `solve` is the seam the tests replace; the reference passes nothing and gets LibrarySolver.
template <class Traits, class MapPointT = typename Traits::MapPoint, class MapPlaneT = typename Traits::MapPlane, class MapCuboidT = typename Traits::MapCuboid,
          class KeyFrameT, class MapT, class SolveFn = LibrarySolver>
int local_ba_camera_plane_cuboids(KeyFrameT* pKF, bool* pbStopFlag, MapT* pMap, bool fixCamera, bool fixPoint, pf::Params par, SolveFn solve = SolveFn()) {
  static_assert(sizeof(bool) == 1, "the stop flag is handed to the library as one byte");
  auto g = gather_window<MapPointT, MapPlaneT, MapCuboidT>(pKF);  // OPT:1997-2100
  pf::Window w;
  std::vector<KeyFrameT*> kfs;
  to_window(g, w, kfs);
  par.fix_camera = fixCamera;
  par.fix_point = fixPoint;
  par.optimize_with_cuboid_3d = false;  // the 9-D camera-object edge exists only in LocalBACameraPointCuboids2D (OPT:1767-1800)
  pf::Flat f;
  pf::flatten(w, par, f);  // OPT:2120-2714
  store_used_points(g, f);
  auto reset_markers = [&]() {  // what OPT:2917-2957 does to the markers, for the paths that return before the write-back
    for (KeyFrameT* k : g.lLocalKeyFrames) k->mnBALocalForKF = 0;
    for (KeyFrameT* k : g.lFixedCameras) { k->mnBAFixedForKF = 0; k->mnBALocalForKF = 0; }
    for (MapPointT* p : g.lLocalMapPoints) p->mnBALocalForKF = 0;
    for (MapCuboidT* c : g.lLocalMapCuboids) c->mnBALocalForKF = 0;
  };
  if (pbStopFlag && *pbStopFlag) return PPO_BA_ABORTED;  // OPT:2723-2725: `return` before optimising -- no write-back, and the BA markers
                                                          // STAY set, as in the reference (the next call overwrites them with its own pKF->mnId)
  ppo_ba_options O;
  options_from_params(par, O);
  Solved s;
  const int rc = solve(f.problem, O, reinterpret_cast<const volatile uint8_t*>(pbStopFlag), s);
  if (rc < 0) {
    std::cerr << "LocalBACameraPlaneCuboids: " << ppo_ba_strerror(rc) << std::endl;
    reset_markers();
    return rc;
  }
  pf::WriteBack wb;
  pf::write_back(w, par, f, s.result, wb);         // OPT:2840-2887 erase lists; float32 narrowing of Converter.cc
  apply_write_back<Traits>(g, kfs, wb, pMap);      // OPT:2892-2966
  return rc;
}

// ---- Optimizer::LocalBACameraPointCuboids2D (include/Optimizer.h, OPT:1252-1992) ---------------------------------------
// The point + cuboid local BA (its call site in LocalMapping is commented out, src/LocalMapping.cc:97).  Differences to the function
// above, each as in the reference: planes are neither gathered nor optimised nor written back; the 9-D camera-object edge EdgeSE3Cuboid
// is added when par.optimize_with_cuboid_3d is set (OPT:1767-1800) and classified by |e| > thHuberSE3 between the passes (OPT:1875-1882).
template <class Traits, class MapPointT = typename Traits::MapPoint, class MapPlaneT = typename Traits::MapPlane, class MapCuboidT = typename Traits::MapCuboid,
          class KeyFrameT, class MapT, class SolveFn = LibrarySolver>
int local_ba_camera_point_cuboids_2d(KeyFrameT* pKF, bool* pbStopFlag, MapT* pMap, bool fixCamera, bool fixPoint, pf::Params par, SolveFn solve = SolveFn()) {
  static_assert(sizeof(bool) == 1, "the stop flag is handed to the library as one byte");
  auto g = gather_window<MapPointT, MapPlaneT, MapCuboidT>(pKF, /*with_planes=*/false);  // OPT:1255-1343
  pf::Window w;
  std::vector<KeyFrameT*> kfs;
  to_window(g, w, kfs);
  par.fix_camera = fixCamera;
  par.fix_point = fixPoint;
  par.optimize_with_plane_3d = false;      // this function has no plane vertices or edges
  par.optimize_with_cuboid_plane = false;
  pf::Flat f;
  pf::flatten(w, par, f);  // OPT:1355-1800
  store_used_points(g, f);
  auto reset_markers = [&]() {
    for (KeyFrameT* k : g.lLocalKeyFrames) k->mnBALocalForKF = 0;
    for (KeyFrameT* k : g.lFixedCameras) { k->mnBAFixedForKF = 0; k->mnBALocalForKF = 0; }
    for (MapPointT* p : g.lLocalMapPoints) p->mnBALocalForKF = 0;
    for (MapCuboidT* c : g.lLocalMapCuboids) c->mnBALocalForKF = 0;
  };
  if (pbStopFlag && *pbStopFlag) return PPO_BA_ABORTED;  // OPT:1809-1811: markers stay set, as in the reference
  ppo_ba_options O;
  options_from_params(par, O);
  Solved s;
  const int rc = solve(f.problem, O, reinterpret_cast<const volatile uint8_t*>(pbStopFlag), s);
  if (rc < 0) {
    std::cerr << "LocalBACameraPointCuboids2D: " << ppo_ba_strerror(rc) << std::endl;
    reset_markers();
    return rc;
  }
  pf::WriteBack wb;
  pf::write_back(w, par, f, s.result, wb);         // OPT:1898-1925 erase lists
  apply_write_back<Traits>(g, kfs, wb, pMap);      // OPT:1927-1991 (no planes in the window: nothing is written to one)
  return rc;
}

// ---- Optimizer::LocalBundleAdjustment (include/Optimizer.h, OPT:461-787) ---------------------------------------------
// The points-only local BA LocalMapping runs while the map has at most five keyframes (src/LocalMapping.cc:103-108): the same protocol on
// keyframes and points only.  Differences to the function above, each as in the reference: the gather does not look at cuboids or planes
// (OPT:466-511); EVERY local point gets a vertex (no Observations() == 1 rule, OPT:581-590); `if (*pbStopFlag) return;` before the first
// pass (OPT:664-665); the BA markers are NOT reset afterwards (vanilla ORB-SLAM2 compares them with the unique pKF->mnId).
template <class MapPointT, class MapPlaneT, class MapCuboidT, class KeyFrameT>
Gathered<KeyFrameT, MapPointT, MapPlaneT, MapCuboidT> gather_window_points_only(KeyFrameT* pKF) {
  Gathered<KeyFrameT, MapPointT, MapPlaneT, MapCuboidT> g;
  g.lLocalKeyFrames.push_back(pKF);
  pKF->mnBALocalForKF = pKF->mnId;
  const std::vector<KeyFrameT*> vNeighKFs = pKF->GetVectorCovisibleKeyFrames();
  for (KeyFrameT* pKFi : vNeighKFs) {
    pKFi->mnBALocalForKF = pKF->mnId;
    if (!pKFi->isBad()) g.lLocalKeyFrames.push_back(pKFi);
  }
  for (KeyFrameT* pKFl : g.lLocalKeyFrames) {
    const std::vector<MapPointT*> vpMPs = pKFl->GetMapPointMatches();
    for (MapPointT* pMP : vpMPs)
      if (pMP && !pMP->isBad() && pMP->mnBALocalForKF != pKF->mnId) {
        g.lLocalMapPoints.push_back(pMP);
        pMP->mnBALocalForKF = pKF->mnId;
      }
  }
  for (MapPointT* pMP : g.lLocalMapPoints) {
    const auto observations = pMP->GetObservations();
    for (const auto& mit : observations) {
      KeyFrameT* pKFi = mit.first;
      if (pKFi->mnBALocalForKF != pKF->mnId && pKFi->mnBAFixedForKF != pKF->mnId) {
        pKFi->mnBAFixedForKF = pKF->mnId;
        if (!pKFi->isBad()) g.lFixedCameras.push_back(pKFi);
      }
    }
  }
  return g;
}

template <class Traits, class MapPointT = typename Traits::MapPoint, class MapPlaneT = typename Traits::MapPlane, class MapCuboidT = typename Traits::MapCuboid,
          class KeyFrameT, class MapT, class SolveFn = LibrarySolver>
int local_bundle_adjustment(KeyFrameT* pKF, bool* pbStopFlag, MapT* pMap, SolveFn solve = SolveFn()) {
  auto g = gather_window_points_only<MapPointT, MapPlaneT, MapCuboidT>(pKF);  // OPT:463-511
  pf::Window w;
  std::vector<KeyFrameT*> kfs;
  to_window(g, w, kfs);
  pf::Params par;  // every plane / cuboid flag off: the window carries none
  par.skip_single_observation_points = false;
  pf::Flat f;
  pf::flatten(w, par, f);  // OPT:525-662
  if (pbStopFlag && *pbStopFlag) return PPO_BA_ABORTED;  // OPT:664-665
  ppo_ba_options O;
  ppo_ba_default_options(&O);  // 5 + 10 iterations, 5.991 / 7.815 (OPT:668, 715, 688, 703)
  Solved s;
  const int rc = solve(f.problem, O, reinterpret_cast<const volatile uint8_t*>(pbStopFlag), s);
  if (rc < 0) {
    std::cerr << "LocalBundleAdjustment: " << ppo_ba_strerror(rc) << std::endl;
    return rc;
  }
  pf::WriteBack wb;
  pf::write_back(w, par, f, s.result, wb);                                  // OPT:719-745
  apply_write_back<Traits>(g, kfs, wb, pMap, /*reset_markers=*/false);      // OPT:748-786
  return rc;
}

// ---- Optimizer::PoseOptimization (include/Optimizer.h, OPT:247-459) through include/ppo_pose.h ---------------------------------
// Frame members used: N, mvpMapPoints, mvbOutlier, mvKeysUn, mvuRight, mvInvLevelSigma2, mTcw, fx, fy, cx, cy, mbf, SetPose()
// (include/Frame.h); MapPoint::mGlobalMutex is held while the map points are read (OPT:286), nothing is locked during the solve.
struct PoseSolved {
  double qt[7] = {0, 0, 0, 1, 0, 0, 0};
  std::vector<uint8_t> outlier;
  int32_t n_inliers = 0;
};
struct LibraryPoseSolver {
  int operator()(const ppo_pose_problem& P, const ppo_pose_options& O, PoseSolved& s) const {
    ppo_pose_result R{s.qt, s.outlier.data(), &s.n_inliers, nullptr, nullptr};
    return ppo_pose_optimize(&P, &O, &R);
  }
};
template <class Traits, class MapPointT = typename Traits::MapPoint, class FrameT, class SolveFn = LibraryPoseSolver>
int pose_optimization(FrameT* pFrame, SolveFn solve = SolveFn()) {
  std::vector<double> xw, obs, inv;
  std::vector<size_t> index;
  {
    std::unique_lock<decltype(MapPointT::mGlobalMutex)> lock(MapPointT::mGlobalMutex);  // OPT:286
    const int N = pFrame->N;
    for (int i = 0; i < N; i++) {
      MapPointT* pMP = pFrame->mvpMapPoints[i];
      if (!pMP) continue;
      pFrame->mvbOutlier[i] = false;  // OPT:297, 331
      const auto& kpUn = pFrame->mvKeysUn[i];
      const auto Xw = pMP->GetWorldPos();
      for (int a = 0; a < 3; ++a) xw.push_back(Xw.template at<float>(a));
      obs.push_back(kpUn.pt.x);
      obs.push_back(kpUn.pt.y);
      obs.push_back(pFrame->mvuRight[i]);  // < 0 selects the monocular edge (OPT:294)
      inv.push_back(pFrame->mvInvLevelSigma2[kpUn.octave]);
      index.push_back((size_t)i);
    }
  }
  if (index.size() < 3) return 0;  // OPT:372-373
  float Tcw[16];
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) Tcw[4 * r + c] = pFrame->mTcw.template at<float>(r, c);
  double qt[7];
  pf::pose_from_cv(Tcw, qt);  // Converter::toSE3Quat(pFrame->mTcw), OPT:263, 385
  const double cam[5] = {pFrame->fx, pFrame->fy, pFrame->cx, pFrame->cy, pFrame->mbf};
  const int32_t off[2] = {0, (int32_t)index.size()};
  ppo_pose_problem P{1, qt, cam, off, xw.data(), obs.data(), inv.data()};
  ppo_pose_options O;
  ppo_pose_default_options(&O);  // 4 x 10 iterations, 5.991 / 7.815, the Huber kernel dropped after the third round
  PoseSolved s;
  s.outlier.assign(index.size(), 0);
  const int rc = solve(P, O, s);
  if (rc != PPO_BA_OK) {
    std::cerr << "PoseOptimization: " << ppo_ba_strerror(rc) << std::endl;
    return 0;
  }
  for (size_t k = 0; k < index.size(); k++) pFrame->mvbOutlier[index[k]] = s.outlier[k] != 0;  // OPT:403-411, 432-441
  float T[16];
  pf::pose_to_cv(s.qt, T);
  auto pose = Traits::make_mat(4, 4);
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) pose.template at<float>(r, c) = T[4 * r + c];
  pFrame->SetPose(pose);  // OPT:455-456
  return s.n_inliers;     // OPT:458
}

}  // namespace ppo_from_reference

// ---- the traits of the reference itself (compiled only inside the reference: OpenCV + its own headers) ------------------------------------
#if defined(PPO_FROM_REFERENCE_WITH_OPENCV)
#include <opencv2/core/core.hpp>
#include "MapCuboid.h"
#include "MapPlane.h"
#include "MapPoint.h"
#include "Parameters.h"
#include "g2o_cuboid.h"
namespace ppo_from_reference {
struct CvTraits {
  typedef ORB_SLAM2::MapPoint MapPoint;
  typedef ORB_SLAM2::MapPlane MapPlane;
  typedef ORB_SLAM2::MapCuboid MapCuboid;
  static cv::Mat make_mat(int rows, int cols) { return cv::Mat(rows, cols, CV_32F); }
  static g2o::cuboid make_cuboid(const double qt[7], const double scale[3]) {
    g2o::cuboid c;
    c.pose = g2o::SE3Quat(Eigen::Quaterniond(qt[3], qt[0], qt[1], qt[2]), Eigen::Vector3d(qt[4], qt[5], qt[6]));
    c.scale = Eigen::Vector3d(scale[0], scale[1], scale[2]);
    return c;
  }
};
// the globals of src/Parameters.cc:43-72
inline ppo_flatten::Params params_from_globals() {
  using namespace ORB_SLAM2;
  ppo_flatten::Params par;
  par.optimize_with_plane_3d = optimize_with_plane_3d; par.optimize_with_cuboid_2d = optimize_with_cuboid_2d;
  par.optimize_with_corners_2d = optimize_with_corners_2d; par.optimize_with_pt_obj_3d = optimize_with_pt_obj_3d;
  par.optimize_with_cuboid_plane = optimize_with_cuboid_plane;
  par.plane_angle_info = plane_angle_info; par.plane_dist_info = plane_dist_info; par.plane_chi = plane_chi;
  par.cuboid_plane_angle_info = cuboid_plane_angle_info; par.cuboid_plane_dist_info = cuboid_plane_dist_info; par.cuboid_plane_chi = cuboid_plane_chi;
  par.ba_weight_bbox = ba_weight_bbox; par.ba_weight_corner = ba_weight_corner; par.thHuberBbox2d = thHuberBbox2d; par.thHuberConer2d = thHuberConer2d;
  return par;
}
}  // namespace ppo_from_reference
#endif

#endif  // PPO_FROM_REFERENCE_HPP
