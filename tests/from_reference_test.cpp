// from_reference_test.cpp -- include/ppo_from_reference.hpp instantiated on MOCK classes that carry exactly the member names of the
// reference's KeyFrame / MapPoint / MapPlane / MapCuboid / Map (/root/reference/include/KeyFrame.h:76-242, MapPoint.h, MapPlane.h:27-92,
// MapCuboid.h:35-174, Map.h) and mock value types with the member syntax of cv::Mat, cv::KeyPoint, cv::Rect, Eigen::Vector3d /
// Vector4d / Matrix2Xd / Quaterniond, g2o::SE3Quat and g2o::cuboid.  Checks, against /root/reference/src/Optimizer.cc:
//   * gather (OPT:1997-2100): local keyframes = pKF + non-bad covisibles; local points / cuboids / planes once each, no bad ones,
//     no NULLs; fixed cameras = other non-bad observers of local points AND of local cuboids (not of planes); markers;
//   * the copy loops: every scalar lands in the right Window field (pose, intrinsics, keypoints, octave -> invSigma2, plane
//     coefficients of the OBSERVING keyframe, bbox / corners of the keyframe's LOCAL cuboid, object-point observation counts);
//   * lock order: Map::mMutexMapUpdate is taken only for the write-back (OPT:2892) -- not during gather / flatten / solve -- and
//     every mutating call on the map objects happens under it;
//   * write-back (OPT:2894-2966): erase calls in pairs, SetPose on local keyframes only, SetWorldPos + UpdateNormalAndDepth on points
//     with a vertex, cuboid_global_opti + obj_been_optimized + SetWorldPos, plane SetWorldPos, every marker reset exactly as the
//     reference resets it (planes keep theirs);
//   * the early-outs: stop flag up before the optimisation (OPT:2723-2725) and a failing library (no GPU) leave the map untouched.
#include <cassert>
#include <cmath>
#include <cstdio>
#include <map>
#include <set>
#include <string>
#include <unordered_map>
#include <vector>

#include "../include/ppo_from_reference.hpp"

namespace mock {

// ---- value types -----------------------------------------------------------------------------------------------------------
struct Mat {  // cv::Mat, CV_32F
  int rows = 0, cols = 0;
  std::vector<float> d;
  Mat() {}
  Mat(int r, int c) : rows(r), cols(c), d((size_t)r * c, 0.f) {}
  template <class T> T& at(int r, int c) { return d[(size_t)r * cols + c]; }
  template <class T> const T& at(int r, int c) const { return d[(size_t)r * cols + c]; }
  template <class T> T& at(int i) { return d[i]; }
  template <class T> const T& at(int i) const { return d[i]; }
};
struct Point2f { float x = 0, y = 0; };
struct KeyPoint { Point2f pt; int octave = 0; };
struct Rect { int x = 0, y = 0, width = 0, height = 0; };
struct Vec3 { double v[3] = {0, 0, 0}; double& operator[](int i) { return v[i]; } const double& operator[](int i) const { return v[i]; } };
struct Vec4 { double v[4] = {0, 0, 0, 0}; double& operator[](int i) { return v[i]; } const double& operator[](int i) const { return v[i]; } };
struct Mat2X { double v[2][8] = {{0}}; double& operator()(int r, int c) { return v[r][c]; } const double& operator()(int r, int c) const { return v[r][c]; } };
struct Quaterniond { double X = 0, Y = 0, Z = 0, W = 1; double x() const { return X; } double y() const { return Y; } double z() const { return Z; } double w() const { return W; } };
struct SE3Quat { Quaterniond r; Vec3 t; const Quaterniond& rotation() const { return r; } const Vec3& translation() const { return t; } };
struct cuboid { SE3Quat pose; Vec3 scale; };

// ---- the lock whose order is under test ------------------------------------------------------------------------------------------
struct Journal {
  bool map_locked = false;
  int lock_count = 0;
  std::vector<std::string> ev;
  int unlocked_mutations = 0;
  void mutate(const std::string& what) {
    if (!map_locked) ++unlocked_mutations;
    ev.push_back(what);
  }
};
static Journal* J = nullptr;
struct RecordingMutex {
  void lock() { assert(!J->map_locked); J->map_locked = true; ++J->lock_count; J->ev.push_back("LOCK"); }
  void unlock() { assert(J->map_locked); J->map_locked = false; J->ev.push_back("UNLOCK"); }
};

struct MapPoint;
struct MapPlane;
struct MapCuboid;

struct KeyFrame {
  long unsigned int mnId = 0, mnBALocalForKF = 0, mnBAFixedForKF = 0;
  bool bad = false;
  float fx = 500, fy = 510, cx = 320, cy = 240, mbf = 40;
  Mat mK{3, 3}, Tcw{4, 4};
  std::vector<KeyPoint> mvKeysUn;
  std::vector<float> mvuRight, mvInvLevelSigma2;
  std::vector<MapCuboid*> local_cuboids, mvpMapCuboid;
  std::vector<Mat> mvPlaneCoefficients;
  std::vector<MapPlane*> mvpMapPlanes;
  std::vector<MapPoint*> matches;
  std::vector<KeyFrame*> covisible;
  int n_setpose = 0;
  bool isBad() { return bad; }
  std::vector<KeyFrame*> GetVectorCovisibleKeyFrames() { assert(!J->map_locked); return covisible; }
  std::vector<MapPoint*> GetMapPointMatches() { return matches; }
  Mat GetPose() { return Tcw; }
  void SetPose(const Mat& T) { J->mutate("KF" + std::to_string(mnId) + ".SetPose"); Tcw = T; ++n_setpose; }
  void EraseMapPointMatch(MapPoint* p);
  void EraseMapPlaneMatch(MapPlane* p);
};
struct GlobalRecordingMutex {  // MapPoint::mGlobalMutex
  void lock() { assert(!held); held = true; ++count; }
  void unlock() { assert(held); held = false; }
  bool held = false;
  int count = 0;
};
struct MapPoint {
  static GlobalRecordingMutex mGlobalMutex;
  long unsigned int mnId = 0, mnBALocalForKF = 0;
  bool bad = false;
  Mat X{3, 1};
  std::map<KeyFrame*, size_t> obs;
  std::map<MapCuboid*, int> MapObjObservations;
  int n_set = 0, n_update = 0;
  bool isBad() { return bad; }
  int Observations() { return (int)obs.size(); }
  std::map<KeyFrame*, size_t> GetObservations() { return obs; }
  Mat GetWorldPos() { return X; }
  void SetWorldPos(const Mat& P) { J->mutate("MP" + std::to_string(mnId) + ".SetWorldPos"); X = P; ++n_set; }
  void UpdateNormalAndDepth() { J->mutate("MP" + std::to_string(mnId) + ".UpdateNormalAndDepth"); ++n_update; }
  void EraseObservation(KeyFrame* k) { J->mutate("MP" + std::to_string(mnId) + ".EraseObservation(KF" + std::to_string(k->mnId) + ")"); obs.erase(k); }
};
struct MapPlane {
  long unsigned int mnId = 0, mnBALocalForKF = 0, asso_cuboid_id = 999;
  Vec3 asso_cuboid_meas;
  bool bad = false;
  Mat c{4, 1};
  std::map<KeyFrame*, int> o, vo, po;
  int n_set = 0;
  bool isBad() { return bad; }
  Mat GetWorldPos() { return c; }
  void SetWorldPos(const Mat& P) { J->mutate("PL" + std::to_string(mnId) + ".SetWorldPos"); c = P; ++n_set; }
  std::map<KeyFrame*, int> GetObservations() { return o; }
  std::map<KeyFrame*, int> GetVerObservations() { return vo; }
  std::map<KeyFrame*, int> GetParObservations() { return po; }
  void EraseObservation(KeyFrame* k) { J->mutate("PL" + std::to_string(mnId) + ".EraseObservation(KF" + std::to_string(k->mnId) + ")"); o.erase(k); }
};
struct MapCuboid {
  long int mnId = 0;
  long unsigned int mnBALocalForKF = 0;
  bool bad = false, obj_been_optimized = false;
  cuboid cuboid_global_data, cuboid_global_opti;
  cuboid cuboid_local_meas;  // the detection's 3-D measurement in the camera frame (include/MapCuboid.h), read by LocalBACameraPointCuboids2D
  double meas_quality = 0.7;
  Vec4 bbox_vec;
  Rect bbox_2d;
  Mat2X box_corners_2d;
  std::unordered_map<KeyFrame*, size_t> obs;
  std::vector<MapPoint*> unique, used_points_in_BA_filtered;
  Mat Twc{4, 4};
  int n_set = 0;
  bool isBad() { return bad; }
  std::unordered_map<KeyFrame*, size_t> GetObservations() { return obs; }
  std::vector<MapPoint*> GetUniqueMapPoints() { return unique; }
  void SetWorldPos(const Mat& P) { J->mutate("CB" + std::to_string(mnId) + ".SetWorldPos"); Twc = P; ++n_set; }
};
GlobalRecordingMutex MapPoint::mGlobalMutex;
struct Frame {  // include/Frame.h
  int N = 0;
  std::vector<MapPoint*> mvpMapPoints;
  std::vector<bool> mvbOutlier;
  std::vector<KeyPoint> mvKeysUn;
  std::vector<float> mvuRight, mvInvLevelSigma2{1.f, 0.6944f, 0.4823f, 0.3349f};
  Mat mTcw{4, 4};
  float fx = 500, fy = 510, cx = 320, cy = 240, mbf = 40;
  int n_setpose = 0;
  bool points_read_under_lock = true;
  void SetPose(Mat T) { mTcw = T; ++n_setpose; }
};
void KeyFrame::EraseMapPointMatch(MapPoint* p) { J->mutate("KF" + std::to_string(mnId) + ".EraseMapPointMatch(MP" + std::to_string(p->mnId) + ")"); }
void KeyFrame::EraseMapPlaneMatch(MapPlane* p) { J->mutate("KF" + std::to_string(mnId) + ".EraseMapPlaneMatch(PL" + std::to_string(p->mnId) + ")"); }
struct Map { RecordingMutex mMutexMapUpdate; };

struct Traits {
  typedef mock::MapPoint MapPoint;
  typedef mock::MapPlane MapPlane;
  typedef mock::MapCuboid MapCuboid;
  static Mat make_mat(int r, int c) { return Mat(r, c); }
  static cuboid make_cuboid(const double qt[7], const double scale[3]) {
    cuboid c;
    c.pose.r = Quaterniond{qt[0], qt[1], qt[2], qt[3]};
    for (int a = 0; a < 3; ++a) { c.pose.t[a] = qt[4 + a]; c.scale[a] = scale[a]; }
    return c;
  }
};

}  // namespace mock

#define CHECK(c) do { if (!(c)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

namespace pfr = ppo_from_reference;
namespace pf = ppo_flatten;
using namespace mock;

// A world: keyframes 0..5 (0 = the first keyframe of the map, 4 = bad covisible, 5 = bad observer), 3 further observers (6, 7 = of points /
// of a cuboid, 8 = of a plane only), points, planes, cuboids.
struct World {
  std::vector<KeyFrame> kf{9};
  std::vector<MapPoint> mp{8};
  std::vector<MapPlane> pl{3};
  std::vector<MapCuboid> cb{3};
  std::vector<MapCuboid> local_det{4};  // detections (KeyFrame::local_cuboids)
  std::vector<MapPoint> objpts{14};
  Map map;
  Journal journal;
  World() {
    J = &journal;
    for (int i = 0; i < 9; ++i) {
      KeyFrame& k = kf[i];
      k.mnId = i;
      k.fx = 500 + i; k.fy = 510 + i; k.cx = 320 + 0.5f * i; k.cy = 240 - 0.5f * i; k.mbf = 40 + i;
      for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) k.mK.at<float>(r, c) = (r == c) ? 1.f : 0.f;
      k.mK.at<float>(0, 0) = k.fx; k.mK.at<float>(1, 1) = k.fy; k.mK.at<float>(0, 2) = k.cx; k.mK.at<float>(1, 2) = k.cy;
      for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) k.Tcw.at<float>(r, c) = (r == c) ? 1.f : 0.f;
      k.Tcw.at<float>(0, 3) = 0.1f * i; k.Tcw.at<float>(2, 3) = -0.05f * i;
      k.mvInvLevelSigma2 = {1.f, 0.6944f, 0.4823f, 0.3349f};
      for (int j = 0; j < 12; ++j) {
        KeyPoint kp; kp.pt.x = 100.f + 10 * j + i; kp.pt.y = 50.f + 7 * j - i; kp.octave = j % 4;
        k.mvKeysUn.push_back(kp);
        k.mvuRight.push_back(j % 3 == 0 ? kp.pt.x - 8.f : -1.f);
      }
      Mat pc(4, 1); pc.at<float>(0) = 0; pc.at<float>(1) = 0; pc.at<float>(2) = 1; pc.at<float>(3) = 2.f + 0.25f * i;
      k.mvPlaneCoefficients = {pc, pc};
      k.mvPlaneCoefficients[1].at<float>(3) = 3.f + 0.25f * i;
    }
    kf[4].bad = true;
    kf[5].bad = true;
    kf[1].covisible = {&kf[0], &kf[2], &kf[3], &kf[4]};  // pKF = 1
    for (int j = 0; j < 8; ++j) {
      mp[j].mnId = 100 + j;
      for (int a = 0; a < 3; ++a) mp[j].X.at<float>(a) = 0.5f * j + a;
    }
    mp[6].bad = true;
    // matches: point 0 in KF1+KF2, 1 in KF1+KF6(fixed)+KF5(bad observer), 2 in KF2+KF3 (seen twice via two keyframes), 3 only KF3 (one observation),
    // 4 in KF0+KF1, 5 in KF2+KF7... 7 seen only by fixed-side keyframes -> not local
    auto see = [&](int p, int k, size_t idx) { mp[p].obs[&kf[k]] = idx; };
    see(0, 1, 0); see(0, 2, 1); see(1, 1, 2); see(1, 6, 3); see(1, 5, 4); see(2, 2, 5); see(2, 3, 6); see(3, 3, 7); see(4, 0, 8); see(4, 1, 9);
    see(5, 2, 10); see(5, 6, 11); see(6, 1, 0); see(7, 6, 1);
    kf[1].matches = {&mp[0], nullptr, &mp[1], &mp[4], &mp[6]};
    kf[2].matches = {&mp[0], &mp[2], &mp[5]};
    kf[3].matches = {&mp[2], &mp[3]};
    kf[0].matches = {&mp[4]};
    kf[4].matches = {&mp[7]};  // bad covisible keyframe: never walked
    // planes
    for (int j = 0; j < 3; ++j) { pl[j].mnId = 10 + j; pl[j].c.at<float>(2) = 1; pl[j].c.at<float>(3) = 2.f + j; }
    pl[1].bad = true;
    pl[0].o[&kf[1]] = 0; pl[0].o[&kf[2]] = 1; pl[0].o[&kf[8]] = 0;  // KF8 observes only a plane: NOT collected as a fixed camera
    pl[0].vo[&kf[3]] = 1; pl[0].po[&kf[1]] = 1;
    pl[2].o[&kf[3]] = 0;
    pl[0].asso_cuboid_id = 7; pl[0].asso_cuboid_meas[0] = 0.01; pl[0].asso_cuboid_meas[1] = -0.02; pl[0].asso_cuboid_meas[2] = 0.03;
    kf[1].mvpMapPlanes = {&pl[0], nullptr, &pl[1]};
    kf[3].mvpMapPlanes = {&pl[0], &pl[2]};
    // cuboids: 0 local (seen by KF1, KF2 and KF7 -> KF7 becomes a fixed camera), 1 bad, 2 seen from KF3
    for (int j = 0; j < 3; ++j) {
      cb[j].mnId = 7 + j;
      cb[j].cuboid_global_data.pose.r = Quaterniond{0, 0, std::sin(0.1 * (j + 1)), std::cos(0.1 * (j + 1))};
      cb[j].cuboid_global_data.pose.t[0] = 1 + j; cb[j].cuboid_global_data.pose.t[1] = 0.2; cb[j].cuboid_global_data.pose.t[2] = 3;
      cb[j].cuboid_global_data.scale[0] = 0.3; cb[j].cuboid_global_data.scale[1] = 0.2 + 0.01 * j; cb[j].cuboid_global_data.scale[2] = 0.4;
    }
    cb[1].bad = true;
    for (int d = 0; d < 4; ++d) {
      MapCuboid& lo = local_det[d];
      lo.bbox_vec[0] = 300 + d; lo.bbox_vec[1] = 200 + d; lo.bbox_vec[2] = 80 + d; lo.bbox_vec[3] = 60 + d;
      lo.bbox_2d = Rect{260 + d, 170 + d, 80 + d, 60 + d};
      for (int k = 0; k < 8; ++k) { lo.box_corners_2d(0, k) = 260 + d + k; lo.box_corners_2d(1, k) = 170 + d + 2 * k; }
      lo.meas_quality = 0.5 + 0.1 * d;
    }
    kf[1].local_cuboids = {&local_det[0]}; kf[2].local_cuboids = {nullptr, &local_det[1]}; kf[7].local_cuboids = {&local_det[2]}; kf[3].local_cuboids = {&local_det[3]};
    cb[0].obs[&kf[1]] = 0; cb[0].obs[&kf[2]] = 1; cb[0].obs[&kf[7]] = 0;
    cb[2].obs[&kf[3]] = 0;
    kf[1].mvpMapCuboid = {&cb[0], &cb[1], nullptr};
    kf[2].mvpMapCuboid = {&cb[0]};
    kf[3].mvpMapCuboid = {&cb[2]};
    // object points of cuboid 0: 14 points, 12 of them seen with the object more than twice and close together
    for (int j = 0; j < 14; ++j) {
      objpts[j].mnId = 500 + j;
      objpts[j].X.at<float>(0) = 1.f + 0.02f * j; objpts[j].X.at<float>(1) = 0.2f; objpts[j].X.at<float>(2) = 3.f - 0.01f * j;
      objpts[j].MapObjObservations[&cb[0]] = (j < 12) ? 3 + j % 2 : 2;
      cb[0].unique.push_back(&objpts[j]);
    }
    cb[0].unique.insert(cb[0].unique.begin() + 3, nullptr);  // a hole in GetUniqueMapPoints()
  }
};

static int count(const std::vector<std::string>& ev, const std::string& prefix) {
  int n = 0;
  for (const auto& e : ev) n += e.compare(0, prefix.size(), prefix) == 0;
  return n;
}

// a stand-in for the library: returns the input estimates moved by a known amount and chosen per-edge chi2 values
struct FakeSolver {
  bool* saw_lock;
  int rc;
  int operator()(const ppo_ba_problem& P, const ppo_ba_options& O, const volatile uint8_t* stop, pfr::Solved& s) const {
    *saw_lock = J->map_locked;
    (void)stop;
    assert(O.iters_pass1 == 5 && O.iters_pass2 == 10);
    s.size_for(P);
    if (rc < 0) return rc;
    for (int i = 0; i < 7 * P.n_poses; ++i) s.pose[i] = P.pose_qt[i];
    for (int p = 0; p < P.n_poses; ++p) s.pose[7 * p + 4] += 0.5;  // tx += 0.5
    for (int i = 0; i < 3 * P.n_points; ++i) s.pts[i] = P.point_xyz[i] + 0.25;
    for (int i = 0; i < 4 * P.n_planes; ++i) s.pln[i] = P.plane_coef[i];
    for (int i = 0; i < 7 * P.n_cuboids; ++i) s.cq[i] = P.cuboid_qt[i];
    for (int i = 0; i < 3 * P.n_cuboids; ++i) s.cs[i] = P.cuboid_scale[i] + 0.125;
    for (int e = 0; e < P.n_mono; ++e) { s.mchi[e] = (e == 0) ? 9.0 : 1.0; s.mdp[e] = 1; }      // first mono edge: chi2 > 5.991 -> erased
    for (int e = 0; e < P.n_stereo; ++e) { s.schi[e] = 1.0; s.sdp[e] = (e == 0) ? 0 : 1; }       // first stereo edge: behind the camera -> erased
    for (int e = 0; e < P.n_plane_edges; ++e) s.pchi[e] = (e == 0) ? 1e6 : 1.0;                  // first plane edge: chi2 > plane_chi -> erased
    return rc;
  }
};

int main() {
  // ---- 1. the gather ---------------------------------------------------------------------------------------------------------
  {
    World W;
    auto g = pfr::gather_window<MapPoint, MapPlane, MapCuboid>(&W.kf[1]);
    CHECK(W.journal.lock_count == 0);
    std::vector<long> local, fixed;
    for (auto* k : g.lLocalKeyFrames) local.push_back((long)k->mnId);
    for (auto* k : g.lFixedCameras) fixed.push_back((long)k->mnId);
    CHECK((local == std::vector<long>{1, 0, 2, 3}));           // pKF first, covisibles in order, bad KF4 dropped (OPT:2008)
    CHECK(W.kf[4].mnBALocalForKF == 1);                        // ... but marked (OPT:2006)
    std::set<long> pts;
    for (auto* p : g.lLocalMapPoints) pts.insert((long)p->mnId);
    CHECK((pts == std::set<long>{100, 101, 102, 103, 104, 105}));  // no NULL, no bad point 106, not 107 (only seen from the bad KF4 / KF6), each once
    CHECK(g.lLocalMapPoints.size() == 6);
    CHECK((g.lLocalMapPoints[0]->mnId == 100 && g.lLocalMapPoints[1]->mnId == 101 && g.lLocalMapPoints[2]->mnId == 104));  // KF1's matches first
    CHECK(g.lLocalMapCuboids.size() == 2 && g.lLocalMapCuboids[0]->mnId == 7 && g.lLocalMapCuboids[1]->mnId == 9);
    CHECK(g.lLocalMapPlanes.size() == 2 && g.lLocalMapPlanes[0]->mnId == 10 && g.lLocalMapPlanes[1]->mnId == 12);
    CHECK((std::set<long>(fixed.begin(), fixed.end()) == std::set<long>{6, 7}));  // observers of local points (6) and of a local cuboid (7); not the bad KF5, not the plane observer KF8
    CHECK(W.kf[5].mnBAFixedForKF == 1 && W.kf[8].mnBAFixedForKF == 0);          // the bad observer is marked, not pushed (OPT:2077-2080)
    for (auto* k : g.lLocalKeyFrames) CHECK(k->mnBALocalForKF == 1);
    for (auto* p : g.lLocalMapPoints) CHECK(p->mnBALocalForKF == 1);
  }
  // ---- 2. the copy loops -------------------------------------------------------------------------------------------------------
  {
    World W;
    auto g = pfr::gather_window<MapPoint, MapPlane, MapCuboid>(&W.kf[1]);
    pf::Window w;
    std::vector<KeyFrame*> kfs;
    pfr::to_window(g, w, kfs);
    CHECK(w.keyframes.size() == 6 && kfs.size() == 6);
    for (size_t i = 0; i < kfs.size(); ++i) {
      const pf::KeyFrameIn& in = w.keyframes[i];
      CHECK(in.id == (long)kfs[i]->mnId && in.local == (i < 4) && !in.bad);
      CHECK(in.fx == kfs[i]->fx && in.fy == kfs[i]->fy && in.cx == kfs[i]->cx && in.cy == kfs[i]->cy && in.bf == kfs[i]->mbf);
      CHECK(in.Tcw[3] == 0.1f * in.id && in.Tcw[11] == -0.05f * in.id && in.Tcw[15] == 1.f && in.K[0] == in.fx && in.K[5] == in.cy);
    }
    // point 101: observed by KF1 (idx 2), KF5 (bad: no vertex -> kf = -1), KF6 (fixed)
    const pf::MapPointIn& p1 = w.points[1];
    CHECK(p1.id == 101 && p1.n_observations == 3 && p1.obs.size() == 3 && p1.xyz[0] == 0.5f && p1.xyz[2] == 2.5f);
    int n_local = 0, n_fixed = 0, n_none = 0;
    for (const auto& o : p1.obs) {
      if (o.kf < 0) { ++n_none; continue; }
      const KeyFrame* k = kfs[o.kf];
      const size_t idx = W.mp[1].obs[const_cast<KeyFrame*>(k)];
      CHECK(o.u == k->mvKeysUn[idx].pt.x && o.v == k->mvKeysUn[idx].pt.y && o.u_right == k->mvuRight[idx]);
      CHECK(o.inv_sigma2 == k->mvInvLevelSigma2[k->mvKeysUn[idx].octave]);
      (w.keyframes[o.kf].local ? n_local : n_fixed)++;
    }
    CHECK(n_local == 1 && n_fixed == 1 && n_none == 1);
    // plane 10: three observations (KF1, KF2, KF8 -> -1), one vertical (KF3), one parallel (KF1); coefficients of the OBSERVING keyframe
    const pf::MapPlaneIn& q0 = w.planes[0];
    CHECK(q0.id == 10 && q0.obs.size() == 3 && q0.ver_obs.size() == 1 && q0.par_obs.size() == 1 && q0.asso_cuboid_id == 7 && q0.asso_cuboid_meas[1] == -0.02);
    for (const auto& o : q0.obs)
      if (o.kf >= 0) CHECK(o.coef[3] == (kfs[o.kf]->mnId == 1 ? 2.f + 0.25f : 3.f + 0.5f));  // KF1 -> coefficient 0, KF2 -> coefficient 1
    CHECK(q0.ver_obs[0].coef[3] == 3.f + 0.75f && kfs[q0.ver_obs[0].kf]->mnId == 3);
    // cuboid 7: pose / scale, three observations with the detection of the observing keyframe
    const pf::MapCuboidIn& c0 = w.cuboids[0];
    CHECK(c0.id == 7 && c0.qt[2] == std::sin(0.1) && c0.qt[3] == std::cos(0.1) && c0.qt[4] == 1 && c0.scale[1] == 0.2 && c0.obs.size() == 3);
    for (const auto& o : c0.obs) {
      CHECK(o.kf >= 0);
      const long id = (long)kfs[o.kf]->mnId;
      const int d = id == 1 ? 0 : (id == 2 ? 1 : 2);
      CHECK(o.bbox_vec[0] == 300 + d && o.bbox_x == 260 + d && o.bbox_w == 80 + d && o.corners_2d[2 * 3] == 260 + d + 3 && o.corners_2d[2 * 3 + 1] == 170 + d + 6);
      CHECK(o.meas_quality == 0.5 + 0.1 * d);
    }
    CHECK(c0.unique_points.size() == 15 && c0.unique_points[3].bad && c0.unique_points[0].n_object_observations == 3 && c0.unique_points[14].n_object_observations == 2);
    CHECK(c0.unique_points[1].xyz[0] == 1.f + 0.02f);
    CHECK(W.journal.lock_count == 0 && W.journal.ev.empty());
  }
  // ---- 3. the whole function with a stand-in solver: lock order, erase calls, write-back, markers -----------------------------------------
  {
    World W;
    pf::Params par;
    par.optimize_with_plane_3d = par.optimize_with_cuboid_2d = par.optimize_with_pt_obj_3d = par.optimize_with_cuboid_plane = true;
    bool solver_saw_lock = true, stop = false;
    const int rc = pfr::local_ba_camera_plane_cuboids<Traits>(&W.kf[1], &stop, &W.map, false, false, par, FakeSolver{&solver_saw_lock, PPO_BA_OK});
    CHECK(rc == PPO_BA_OK);
    CHECK(!solver_saw_lock);                                       // the library runs WITHOUT the map lock (all GPU work before OPT:2892)
    CHECK(W.journal.lock_count == 1 && W.journal.unlocked_mutations == 0 && !W.journal.map_locked);
    CHECK(W.journal.ev.front() == "LOCK" && W.journal.ev.back() == "UNLOCK");  // every mutation between the two
    // erase lists: one mono edge, one stereo edge, one plane edge -> calls in pairs
    CHECK(count(W.journal.ev, "KF") >= 3);
    int n_erase_mp = 0, n_erase_obs = 0, n_erase_pl = 0;
    for (size_t i = 0; i < W.journal.ev.size(); ++i) {
      const std::string& e = W.journal.ev[i];
      if (e.find(".EraseMapPointMatch(") != std::string::npos) { ++n_erase_mp; CHECK(W.journal.ev[i + 1].find(".EraseObservation(") != std::string::npos); }
      if (e.find(".EraseMapPlaneMatch(") != std::string::npos) { ++n_erase_pl; CHECK(W.journal.ev[i + 1].compare(0, 2, "PL") == 0); }
      if (e.compare(0, 2, "MP") == 0 && e.find(".EraseObservation(") != std::string::npos) ++n_erase_obs;
    }
    CHECK(n_erase_mp == 2 && n_erase_obs == 2 && n_erase_pl == 1);
    // poses: local keyframes only (KF0 is fixed but local: the reference writes it back too), tx moved by 0.5 and narrowed to float
    for (int i : {1, 0, 2, 3}) CHECK(W.kf[i].n_setpose == 1 && W.kf[i].Tcw.at<float>(0, 3) == (float)(0.1f * i + 0.5));
    CHECK(W.kf[6].n_setpose == 0 && W.kf[7].n_setpose == 0 && W.kf[4].n_setpose == 0);
    // points: the single-observation point 103 has no vertex (OPT:2336) and is left alone; the others move by 0.25
    for (int j : {0, 1, 2, 4, 5}) CHECK(W.mp[j].n_set == 1 && W.mp[j].n_update == 1 && W.mp[j].X.at<float>(1) == (float)(0.5f * j + 1 + 0.25));
    CHECK(W.mp[3].n_set == 0 && W.mp[6].n_set == 0 && W.mp[7].n_set == 0);
    // cuboids and planes
    CHECK(W.cb[0].obj_been_optimized && W.cb[0].n_set == 1 && W.cb[0].cuboid_global_opti.scale[1] == 0.2 + 0.125 && W.cb[0].cuboid_global_opti.pose.r.W == std::cos(0.1));
    CHECK(W.cb[0].Twc.at<float>(0, 3) == 1.f && W.cb[0].Twc.at<float>(3, 3) == 1.f);
    CHECK(W.cb[2].obj_been_optimized && !W.cb[1].obj_been_optimized);
    CHECK(W.pl[0].n_set == 1 && W.pl[2].n_set == 1 && W.pl[1].n_set == 0);
    // the point-object filter's side effect (OPT:2587-2616): 12 object points with more than two observations, all within the radii
    CHECK(W.cb[0].used_points_in_BA_filtered.size() == 12 && W.cb[0].used_points_in_BA_filtered[3] == &W.objpts[3]);
    // markers: keyframes (local, fixed), points, cuboids reset; bad / unpushed ones keep theirs, and so do planes (OPT:2960-2966 has no reset)
    for (int i : {0, 1, 2, 3, 6, 7}) CHECK(W.kf[i].mnBALocalForKF == 0 && W.kf[i].mnBAFixedForKF == 0);
    CHECK(W.kf[4].mnBALocalForKF == 1 && W.kf[5].mnBAFixedForKF == 1);
    for (int j : {0, 1, 2, 3, 4, 5}) CHECK(W.mp[j].mnBALocalForKF == 0);
    CHECK(W.cb[0].mnBALocalForKF == 0 && W.cb[2].mnBALocalForKF == 0 && W.pl[0].mnBALocalForKF == 1);
  }
  // ---- 4. stop flag up before the optimisation (OPT:2723-2725): no solve, no lock, no write-back ---------------------------------------
  {
    World W;
    pf::Params par;
    bool solver_saw_lock = false, called = false, stop = true;
    struct Never { bool* called; int operator()(const ppo_ba_problem&, const ppo_ba_options&, const volatile uint8_t*, pfr::Solved&) const { *called = true; return 0; } };
    (void)solver_saw_lock;
    const int rc = pfr::local_ba_camera_plane_cuboids<Traits>(&W.kf[1], &stop, &W.map, false, false, par, Never{&called});
    CHECK(rc == PPO_BA_ABORTED && !called && W.journal.lock_count == 0 && W.journal.ev.empty());
    for (int i = 0; i < 9; ++i) CHECK(W.kf[i].n_setpose == 0);
    for (int i : {0, 1, 2, 3}) CHECK(W.kf[i].mnBALocalForKF == 1);  // the reference `return`s with the markers set (OPT:2723-2725) ...
    CHECK(W.kf[6].mnBAFixedForKF == 1 && W.mp[0].mnBALocalForKF == 1 && W.cb[0].mnBALocalForKF == 1);  // ... every one of them
  }
  // ---- 5. the real library without a usable GPU fails loudly and leaves the map alone (with a GPU it succeeds) -------------------------
  {
    World W;
    pf::Params par;
    bool stop = false;
    const int rc = pfr::local_ba_camera_plane_cuboids<Traits>(&W.kf[1], &stop, &W.map, false, false, par);  // LibrarySolver: ppo_ba_create / run / read / destroy
    if (rc == PPO_BA_ERR_HIP) {
      CHECK(W.journal.lock_count == 0 && W.journal.ev.empty());
      for (int i : {0, 1, 2, 3, 6, 7}) CHECK(W.kf[i].mnBALocalForKF == 0 && W.kf[i].mnBAFixedForKF == 0);
      std::printf("(no GPU: ppo_ba_create -> %s, map untouched)\n", ppo_ba_strerror(rc));
    } else {
      CHECK(rc == PPO_BA_OK);
      CHECK(W.journal.lock_count == 1 && W.journal.unlocked_mutations == 0);
      for (int i : {1, 2, 3}) CHECK(W.kf[i].n_setpose == 1);
      std::printf("(GPU: the mock window went through libppo_ba.so)\n");
    }
  }
  // ---- 5b. Optimizer::LocalBACameraPointCuboids2D (OPT:1252-1992): no planes anywhere, the 9-D camera-object edges when the flag is set -----
  {
    World W;
    for (size_t d = 0; d < W.local_det.size(); ++d) {  // give every detection a 3-D measurement
      W.local_det[d].cuboid_local_meas = W.cb[0].cuboid_global_data;
      W.local_det[d].cuboid_local_meas.scale[0] = 0.3 + 0.01 * (double)d;
    }
    pf::Params par;
    par.optimize_with_plane_3d = par.optimize_with_cuboid_plane = true;  // must be ignored by this entry point
    par.optimize_with_cuboid_2d = par.optimize_with_cuboid_3d = par.optimize_with_pt_obj_3d = true;
    bool solver_saw_lock = true, stop = false;
    int n_cub3d = -1, n_bbox = -1;
    double huber0 = 0, info0 = 0, th = 0, meas7 = 0;
    struct Capture {
      bool* saw_lock; int* n_cub3d; int* n_bbox; double* huber0; double* info0; double* th; double* meas7;
      int operator()(const ppo_ba_problem& P, const ppo_ba_options& O, const volatile uint8_t* st, pfr::Solved& s) const {
        assert(P.n_planes == 0 && P.n_plane_edges == 0 && P.n_cubpl == 0);
        *n_cub3d = P.n_cub3d; *n_bbox = P.n_bbox; *th = O.th_cub3d_norm;
        if (P.n_cub3d > 0) { *huber0 = P.cub3d_huber[0]; *info0 = P.cub3d_info[4]; *meas7 = P.cub3d_meas[7]; }
        return FakeSolver{saw_lock, PPO_BA_OK}(P, O, st, s);
      }
    };
    const int rc = pfr::local_ba_camera_point_cuboids_2d<Traits>(&W.kf[1], &stop, &W.map, false, false, par,
                                                                 Capture{&solver_saw_lock, &n_cub3d, &n_bbox, &huber0, &info0, &th, &meas7});
    CHECK(rc == PPO_BA_OK && !solver_saw_lock);
    CHECK(n_cub3d >= 3 && n_cub3d >= n_bbox);                   // every non-bad observer of a local cuboid, no image-boundary test (OPT:1767-1800)
    CHECK(huber0 == 900.0 && th == 900.0);                      // thHuberSE3 itself, as delta and as the classification threshold (OPT:1793, 1878)
    CHECK(info0 == 0.75 * 0.75);                                // (ba_weight_SE3 * 0.75)^2 (OPT:1785-1789)
    CHECK(meas7 >= 0.3 && meas7 < 0.35);                        // the detection's cuboid_local_meas.scale
    CHECK(W.journal.lock_count == 1 && W.journal.unlocked_mutations == 0);
    for (int i : {1, 0, 2, 3}) CHECK(W.kf[i].n_setpose == 1);
    CHECK(W.cb[0].obj_been_optimized && W.cb[0].n_set == 1);
    for (int k = 0; k < 3; ++k) CHECK(W.pl[k].n_set == 0 && W.pl[k].mnBALocalForKF == 0);  // planes are never touched, not even marked
    for (int i : {0, 1, 2, 3, 6, 7}) CHECK(W.kf[i].mnBALocalForKF == 0 && W.kf[i].mnBAFixedForKF == 0);
    // the main entry point never builds the 9-D edge, whatever the flag says
    World W2;
    int n2 = -1;
    struct Count3d { int* n; int operator()(const ppo_ba_problem& P, const ppo_ba_options& O, const volatile uint8_t* st, pfr::Solved& s) const {
      *n = P.n_cub3d; bool dummy = false; return FakeSolver{&dummy, PPO_BA_OK}(P, O, st, s); } };
    CHECK(pfr::local_ba_camera_plane_cuboids<Traits>(&W2.kf[1], &stop, &W2.map, false, false, par, Count3d{&n2}) == PPO_BA_OK && n2 == 0);
  }
  // ---- 6. Optimizer::LocalBundleAdjustment (points only, OPT:461-787): every point gets a vertex, markers stay, same lock order ---------
  {
    World W;
    bool solver_saw_lock = true, stop = false;
    int n_points_seen = -1;
    struct PointsOnly {
      bool* saw_lock; int* n_points;
      int operator()(const ppo_ba_problem& P, const ppo_ba_options& O, const volatile uint8_t* st, pfr::Solved& s) const {
        *n_points = P.n_points;
        assert(P.n_planes == 0 && P.n_cuboids == 0 && P.n_plane_edges == 0 && P.n_bbox == 0 && P.n_ptcub == 0);
        return FakeSolver{saw_lock, PPO_BA_OK}(P, O, st, s);
      }
    };
    const int rc = pfr::local_bundle_adjustment<Traits>(&W.kf[1], &stop, &W.map, PointsOnly{&solver_saw_lock, &n_points_seen});
    CHECK(rc == PPO_BA_OK && !solver_saw_lock);
    CHECK(n_points_seen == 6);                                   // the single-observation point 103 HAS a vertex here (OPT:581-590)
    CHECK(W.journal.lock_count == 1 && W.journal.unlocked_mutations == 0);
    CHECK(W.mp[3].n_set == 1 && W.mp[3].n_update == 1);
    for (int i : {1, 0, 2, 3}) CHECK(W.kf[i].n_setpose == 1);
    CHECK(W.kf[6].n_setpose == 0);
    CHECK(W.kf[7].mnBAFixedForKF == 0);                          // cuboid observers are not looked at
    for (int i : {0, 1, 2, 3}) CHECK(W.kf[i].mnBALocalForKF == 1);  // vanilla ORB-SLAM2 leaves the markers
    CHECK(W.kf[6].mnBAFixedForKF == 1 && W.mp[0].mnBALocalForKF == 1);
    CHECK(!W.cb[0].obj_been_optimized && W.pl[0].n_set == 0);
    bool up = true;
    World W2;
    bool called = false;
    struct Never2 { bool* called; int operator()(const ppo_ba_problem&, const ppo_ba_options&, const volatile uint8_t*, pfr::Solved&) const { *called = true; return 0; } };
    CHECK(pfr::local_bundle_adjustment<Traits>(&W2.kf[1], &up, &W2.map, Never2{&called}) == PPO_BA_ABORTED && !called && W2.journal.lock_count == 0);
  }
  // ---- 7. Optimizer::PoseOptimization (OPT:247-459) over a mock Frame --------------------------------------------------------------------
  {
    World W;  // (for its map points)
    Frame F;
    F.N = 10;
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) F.mTcw.at<float>(r, c) = (r == c) ? 1.f : 0.f;
    F.mTcw.at<float>(0, 3) = 0.25f;
    for (int i = 0; i < 10; ++i) {
      F.mvpMapPoints.push_back(i % 3 == 2 ? nullptr : &W.mp[i % 8]);
      F.mvbOutlier.push_back(true);
      KeyPoint kp; kp.pt.x = 100.f + 7 * i; kp.pt.y = 90.f + 3 * i; kp.octave = i % 4;
      F.mvKeysUn.push_back(kp);
      F.mvuRight.push_back(i % 2 ? kp.pt.x - 5.f : -1.f);
    }
    struct FakePose {
      int* n_obs; bool* under_lock;
      int operator()(const ppo_pose_problem& P, const ppo_pose_options& O, pfr::PoseSolved& s) const {
        *under_lock = MapPoint::mGlobalMutex.held;
        *n_obs = P.edge_offsets[1];
        assert(P.n_frames == 1 && O.rounds == 4 && O.iters_per_round == 10);
        assert(P.cam[0] == 500 && P.cam[4] == 40 && P.pose_qt[3] == 1.0 && P.pose_qt[4] == 0.25);
        // observation 0 = feature 0: point 100, monocular, octave 0; observation 1 = feature 1: stereo
        assert(P.obs[0] == 100.f && P.obs[1] == 90.f && P.obs[2] < 0 && P.obs[5] == 107.f - 5.f && P.inv_sigma2[1] == 0.6944f);
        assert(P.xw[3] == 0.5f && P.xw[4] == 1.5f);
        for (int a = 0; a < 7; ++a) s.qt[a] = P.pose_qt[a];
        s.qt[5] = 0.75;  // ty
        s.outlier[2] = 1;
        s.n_inliers = *n_obs - 1;
        return PPO_BA_OK;
      }
    };
    int n_obs = 0;
    bool under_lock = true;
    const int inl = pfr::pose_optimization<Traits>(&F, FakePose{&n_obs, &under_lock});
    CHECK(n_obs == 7 && inl == 6 && !under_lock && MapPoint::mGlobalMutex.count == 1 && !MapPoint::mGlobalMutex.held);
    CHECK(F.n_setpose == 1 && F.mTcw.at<float>(1, 3) == 0.75f && F.mTcw.at<float>(0, 3) == 0.25f && F.mTcw.at<float>(3, 3) == 1.f);
    // features 0 1 3 4 6 7 9 carry points: the third of them (feature 3) was marked an outlier, every other one is an inlier; features
    // without a map point keep their flag
    CHECK(F.mvbOutlier[3] && !F.mvbOutlier[0] && !F.mvbOutlier[1] && !F.mvbOutlier[4] && !F.mvbOutlier[9] && F.mvbOutlier[2] && F.mvbOutlier[5]);
    // fewer than three correspondences: 0, nothing touched (OPT:372-373)
    Frame G = F;
    G.n_setpose = 0;
    for (int i = 2; i < 10; ++i) G.mvpMapPoints[i] = nullptr;
    bool called = false;
    struct NeverPose { bool* c; int operator()(const ppo_pose_problem&, const ppo_pose_options&, pfr::PoseSolved&) const { *c = true; return 0; } };
    CHECK(pfr::pose_optimization<Traits>(&G, NeverPose{&called}) == 0 && !called && G.n_setpose == 0);
    // the real library: fails loudly without a GPU (returns 0 inliers, pose untouched), runs with one
    Frame H = F;
    H.n_setpose = 0;
    const int r = pfr::pose_optimization<Traits>(&H);
    CHECK((r == 0 && H.n_setpose == 0) || (r >= 0 && H.n_setpose == 1));
  }
  std::printf("from_reference_test ok\n");
  return 0;
}
