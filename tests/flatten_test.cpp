// flatten_test.cpp -- unit test of include/ppo_flatten.hpp on mock containers (plain g++, no GPU, no OpenCV): every construction
// rule of Optimizer.cc:2120-2714 and every write-back rule of :2840-2966 the header claims, one assertion group each.
// Built and run by tests/test_flatten.py.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../include/ppo_flatten.hpp"

using namespace ppo_flatten;

static int g_fail = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #c); ++g_fail; } } while (0)

static KeyFrameIn kf(long id, bool local, bool bad = false) {
  KeyFrameIn k;
  k.id = id; k.local = local; k.bad = bad;
  // rotation of 90 degrees about z (trace = 1 > 0) for id 1, 180 degrees about x (trace = -1: Shoemake's second branch) for id 2
  if (id == 1) { const float T[16] = {0, -1, 0, 0.5f, 1, 0, 0, -0.25f, 0, 0, 1, 2.0f, 0, 0, 0, 1}; std::memcpy(k.Tcw, T, sizeof(T)); }
  if (id == 2) { const float T[16] = {1, 0, 0, 1.0f, 0, -1, 0, 2.0f, 0, 0, -1, 3.0f, 0, 0, 0, 1}; std::memcpy(k.Tcw, T, sizeof(T)); }
  k.fx = 480; k.fy = 481; k.cx = 320; k.cy = 240; k.bf = 40;
  const float K[9] = {480, 0, 320, 0, 481, 240, 0, 0, 1};
  std::memcpy(k.K, K, sizeof(K));
  return k;
}
static PointObsIn pobs(int kfi, float u, float v, float ur = -1, float w = 1) { PointObsIn o; o.kf = kfi; o.u = u; o.v = v; o.u_right = ur; o.inv_sigma2 = w; return o; }

int main() {
  Window w;
  // keyframes: 0 local id 0 (fixed by id), 1 local id 1, 2 local bad neighbour (never enters), 3 fixed camera id 2, 4 fixed camera but bad
  w.keyframes = {kf(0, true), kf(1, true), kf(7, true, true), kf(2, false), kf(9, false, true)};
  // points: 0 two observations (mono + stereo); 1 a single observation (no vertex); 2 observed by the bad keyframe and by a keyframe outside the graph
  MapPointIn p0; p0.id = 10; p0.xyz[0] = 1.5f; p0.xyz[1] = -0.25f; p0.xyz[2] = 4.0f; p0.n_observations = 2;
  p0.obs = {pobs(0, 100.5f, 200.25f, -1, 0.694444f), pobs(1, 110, 210, 95.5f, 0.5f)};
  MapPointIn p1; p1.id = 11; p1.n_observations = 1; p1.obs = {pobs(0, 1, 2)};
  MapPointIn p2; p2.id = 12; p2.n_observations = 3; p2.obs = {pobs(2, 5, 6), pobs(-1, 7, 8), pobs(3, 9, 10)};
  w.points = {p0, p1, p2};
  // planes: 0 with an ordinary, a vertical and a parallel observation and a cuboid association; 1 with negative d and an association to an absent cuboid
  MapPlaneIn pl0; pl0.id = 3; pl0.coef[0] = 0; pl0.coef[1] = 0; pl0.coef[2] = 1; pl0.coef[3] = 2;
  PlaneObsIn po; po.kf = 1; po.coef[0] = 0; po.coef[1] = 0.6f; po.coef[2] = 0.8f; po.coef[3] = 1.5f;
  PlaneObsIn po_bad = po; po_bad.kf = 2;
  pl0.obs = {po, po_bad}; pl0.ver_obs = {po}; pl0.par_obs = {po};
  pl0.asso_cuboid_id = 40; pl0.asso_cuboid_meas[0] = 0.01; pl0.asso_cuboid_meas[1] = -0.02; pl0.asso_cuboid_meas[2] = 0.03;
  MapPlaneIn pl1; pl1.id = 4; pl1.coef[0] = 1; pl1.coef[3] = -3; pl1.asso_cuboid_id = 77;
  w.planes = {pl0, pl1};
  // cuboids: 0 (id 40) with an in-FOV and an out-of-FOV detection and 14 unique points (12 near, 1 far at 5 m, 1 seen twice only)
  MapCuboidIn c0; c0.id = 40; c0.qt[2] = std::sin(0.2); c0.qt[3] = std::cos(0.2); c0.qt[4] = 1; c0.qt[5] = 0.5; c0.qt[6] = 5; c0.scale[0] = 0.3; c0.scale[1] = 0.2; c0.scale[2] = 0.4;
  CuboidObsIn in; in.kf = 1; in.bbox_x = 100; in.bbox_y = 120; in.bbox_w = 80; in.bbox_h = 60; in.bbox_vec[0] = 140; in.bbox_vec[1] = 150; in.bbox_vec[2] = 80; in.bbox_vec[3] = 60; in.meas_quality = 0.7;
  for (int i = 0; i < 16; ++i) in.corners_2d[i] = 100 + i;
  CuboidObsIn edge = in; edge.bbox_x = 5;                       // x > 5 fails
  CuboidObsIn right = in; right.bbox_x = 556; right.bbox_w = 79;  // 556 + 79 = 635 < 635 fails
  CuboidObsIn ok2 = in; ok2.kf = 3; ok2.bbox_x = 555; ok2.bbox_w = 79;  // 634 < 635 passes, seen by the fixed camera
  c0.obs = {in, edge, right, ok2};
  for (int i = 0; i < 12; ++i) { ObjectPointIn q; q.xyz[0] = 1.0f + 0.01f * i; q.xyz[1] = 0.5f; q.xyz[2] = 5.0f; q.n_object_observations = 3; c0.unique_points.push_back(q); }
  { ObjectPointIn q; q.xyz[0] = 1.0f; q.xyz[1] = 0.5f; q.xyz[2] = 10.0f; q.n_object_observations = 5; c0.unique_points.push_back(q); }   // 5 m from the rest: out at the 4 m stage
  { ObjectPointIn q; q.xyz[0] = 1.0f; q.xyz[1] = 0.5f; q.xyz[2] = 5.0f; q.n_object_observations = 2; c0.unique_points.push_back(q); }    // not "> 2"
  { ObjectPointIn q; q.xyz[0] = 1.0f; q.xyz[1] = 0.5f; q.xyz[2] = 5.0f; q.n_object_observations = 9; q.bad = true; c0.unique_points.push_back(q); }
  MapCuboidIn c1; c1.id = 41;  // ten unique points only: no point-object edge ("> 10")
  for (int i = 0; i < 10; ++i) { ObjectPointIn q; q.xyz[2] = 3.0f; q.n_object_observations = 3; c1.unique_points.push_back(q); }
  w.cuboids = {c0, c1};

  Params par;
  par.optimize_with_plane_3d = par.optimize_with_cuboid_2d = par.optimize_with_corners_2d = par.optimize_with_pt_obj_3d = par.optimize_with_cuboid_plane = true;
  Flat f;
  flatten(w, par, f);
  const ppo_ba_problem& q = f.problem;

  // ---- keyframes (OPT:2120-2145) ----
  CHECK(q.n_poses == 3);
  CHECK(f.pose_kf[0] == 0 && f.pose_kf[1] == 1 && f.pose_kf[2] == 3);          // locals first, then the fixed cameras; bad ones nowhere
  CHECK(f.kf_pose[2] == -1 && f.kf_pose[4] == -1);
  CHECK(f.pose_fixed[0] == 1 && f.pose_fixed[1] == 0 && f.pose_fixed[2] == 1);  // id 0 fixed, fixed camera fixed
  CHECK(std::fabs(f.pose_qt[7 + 2] - std::sqrt(0.5)) < 1e-15 && std::fabs(f.pose_qt[7 + 3] - std::sqrt(0.5)) < 1e-15);  // Rz(90): q = (0,0,s,s)
  CHECK(f.pose_qt[7 + 4] == 0.5 && f.pose_qt[7 + 5] == -0.25 && f.pose_qt[7 + 6] == 2.0);
  CHECK(f.pose_qt[14 + 0] == 1.0 && f.pose_qt[14 + 3] == 0.0);                   // Rx(180): q = (1,0,0,0) through the trace <= 0 branch
  CHECK(f.pose_cam[5] == 480 && f.pose_cam[6] == 481 && f.pose_cam[9] == 40);
  // ---- points (OPT:2332-2425) ----
  CHECK(q.n_points == 2 && f.point_vertex[0] == 0 && f.point_vertex[1] == -1 && f.point_vertex[2] == 1);
  CHECK(q.n_mono == 2 && q.n_stereo == 1);  // p0: mono + stereo; p2: only the observation by the fixed camera survives
  CHECK(f.mono_point[0] == 0 && f.mono_pose[0] == 0 && f.mono_obs[0] == 100.5 && f.mono_obs[1] == 200.25 && f.mono_info[0] == (double)0.694444f);
  CHECK(f.mono_huber[0] == (double)(float)std::sqrt(5.991));
  CHECK(f.stereo_pose[0] == 1 && f.stereo_obs[2] == 95.5 && f.stereo_huber[0] == (double)(float)std::sqrt(7.815));
  CHECK(f.mono_point[1] == 1 && f.mono_pose[1] == 2 && f.mono_obs_src[1] == 2);
  // ---- planes (OPT:2208-2307) ----
  CHECK(q.n_planes == 2 && q.n_plane_edges == 3);  // ordinary + vertical + parallel by keyframe 1; the bad keyframe's observation dropped
  CHECK(f.pl_kind[0] == 0 && f.pl_kind[1] == 1 && f.pl_kind[2] == 2);
  CHECK(f.pl_info[0] == 3282.8 && f.pl_info[2] == 10000.0 && f.pl_info[3] == 3282.8 / 0.25);
  CHECK(f.pl_huber[0] == (double)(float)std::sqrt(500.0) && f.pl_huber[1] == (double)(float)std::sqrt(200.0));
  CHECK(f.pl_meas[1] == (double)0.6f && f.pl_obs_src[0] == 0 && f.pl_obs_src[1] == -1);
  CHECK(f.plane_coef[4 + 3] == -3.0);  // raw: the library applies toPlane3D's sign flip and the normalisation
  // ---- cuboid edges (OPT:2433-2551) ----
  CHECK(q.n_cuboids == 2 && f.cuboid_flags[0] == 3);
  CHECK(q.n_bbox == 2 && q.n_corner == 2);  // the two detections 5 px inside 640 x 480
  CHECK(f.bbox_pose[0] == 1 && f.bbox_pose[1] == 2 && f.bbox_cuboid[0] == 0);
  CHECK(std::fabs(f.bbox_info[0] - 0.49) < 1e-15 && f.bbox_huber[0] == (double)(float)std::sqrt(80.0) && f.corner_huber[0] == (double)(float)std::sqrt(10.0));
  CHECK(f.bbox_K[0] == 480 && f.bbox_K[4] == 481 && f.corner_meas[15] == 115);
  // ---- point-object edge (OPT:2556-2655) ----
  CHECK(q.n_ptcub == 1 && f.ptcub_cuboid[0] == 0 && f.ptcub_offsets[1] == 12);
  CHECK(f.ptcub_used[0].size() == 12 && f.ptcub_used[1].size() == 10);  // used_points_in_BA_filtered is filled even when no edge is added
  CHECK(f.ptcub_margin_ratio[0] == 1.0 && q.ptcub_huber == nullptr);
  // ---- cuboid-plane edge (OPT:2662-2714) ----
  CHECK(q.n_cubpl == 1 && f.cubpl_cuboid[0] == 0 && f.cubpl_plane[0] == 0);
  CHECK(f.cubpl_info[0] == 3282.8 / 4.0 && f.cubpl_info[2] == 10000.0 && f.cubpl_meas[1] == -0.02);
  // ---- flags off: nothing but points ----
  Flat g;
  flatten(w, Params(), g);
  CHECK(g.problem.n_plane_edges == 0 && g.problem.n_bbox == 0 && g.problem.n_corner == 0 && g.problem.n_ptcub == 0 && g.problem.n_cubpl == 0);
  CHECK(g.problem.n_planes == 2 && g.problem.n_cuboids == 2);  // the vertices are still created (OPT:2158-2179, 2208-2219) and written back

  // ---- write-back (OPT:2840-2966) ----
  std::vector<double> pose_qt = f.pose_qt, point_xyz = f.point_xyz, plane_coef = {0, 0, 1, 2, 1, 0, 0, 3}, cuboid_qt = f.cuboid_qt, cuboid_scale = f.cuboid_scale;
  point_xyz[0] = 1.23456789012;  // narrowed to float on the way out
  std::vector<double> mono_chi2 = {6.0, 1.0}, stereo_chi2 = {1.0}, pl_chi2 = {600.0, 900.0, 900.0};
  std::vector<uint8_t> mono_dp = {1, 0}, stereo_dp = {1};
  ppo_ba_result r{};
  r.pose_qt = pose_qt.data(); r.point_xyz = point_xyz.data(); r.plane_coef = plane_coef.data(); r.cuboid_qt = cuboid_qt.data(); r.cuboid_scale = cuboid_scale.data();
  r.mono_chi2 = mono_chi2.data(); r.mono_depth_positive = mono_dp.data(); r.stereo_chi2 = stereo_chi2.data(); r.stereo_depth_positive = stereo_dp.data(); r.pl_chi2 = pl_chi2.data();
  WriteBack wb;
  write_back(w, par, f, r, wb);
  CHECK(wb.erase_points.size() == 2);  // chi2 > 5.991 on the first mono edge, depth <= 0 on the second
  CHECK(wb.erase_points[0].object == 0 && wb.erase_points[0].keyframe == 0 && wb.erase_points[1].object == 2 && wb.erase_points[1].keyframe == 3);
  CHECK(wb.erase_planes.size() == 1 && wb.erase_planes[0].object == 0 && wb.erase_planes[0].keyframe == 1);  // only EdgePlane edges are tested (OPT:2878-2887)
  CHECK(wb.poses.size() == 2 && wb.poses[1].keyframe == 1);  // local keyframes only
  CHECK(wb.poses[1].Tcw[1] == -1.0f && wb.poses[1].Tcw[4] == 1.0f && wb.poses[1].Tcw[3] == 0.5f && wb.poses[1].Tcw[15] == 1.0f);
  CHECK(wb.points.size() == 2 && wb.points[0].point == 0 && wb.points[0].xyz[0] == (float)1.23456789012 && wb.points[1].point == 2);
  CHECK(wb.cuboids.size() == 2 && wb.cuboids[0].scale[2] == 0.4 && std::fabs(wb.cuboids[0].Twc[3] - 1.0f) < 1e-7);
  CHECK(wb.planes.size() == 2 && wb.planes[1].coef[3] == 3.0f);
  // a point that has gone bad since the graph was built is not touched by the erase list (OPT:2853)
  w.points[0].bad = true;
  write_back(w, par, f, r, wb);
  CHECK(wb.erase_points.size() == 1 && wb.erase_points[0].object == 2);

  if (g_fail) { std::printf("%d check(s) failed\n", g_fail); return 1; }
  std::printf("flatten_test ok\n");
  return 0;
}
