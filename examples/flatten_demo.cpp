// flatten_demo.cpp -- the seam of INTEGRATION.md from a plain C++ caller: a window held in the adapter's structs
// (include/ppo_flatten.hpp, what Optimizer.cc:1997-2105 gathers) -> flatten() -> libppo_ba.so -> write_back().
// Synthetic scene: keyframes on an arc looking at a wall of points, one plane, one cuboid; truth perturbed, then recovered.
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

#include "ppo_ba.h"
#include "ppo_flatten.hpp"

using namespace ppo_flatten;

int main() {
  std::mt19937_64 rng(7);
  std::normal_distribution<double> N(0.0, 1.0);
  std::uniform_real_distribution<double> U(0.0, 1.0);
  Window w;
  const int n_kf = 8, n_pt = 400;
  std::vector<std::vector<double>> Rt(n_kf, std::vector<double>(12));
  for (int k = 0; k < n_kf; ++k) {
    KeyFrameIn kf;
    kf.id = k; kf.local = k < 6;  // the last two only observe: fixed cameras
    const double a = 0.08 * (k - 3.5), c = std::cos(a), s = std::sin(a);
    const double R[9] = {c, 0, -s, 0, 1, 0, s, 0, c};  // rotation about y
    const double C[3] = {0.4 * (k - 3.5), 0.02 * k, 0.0};
    double t[3];
    for (int r = 0; r < 3; ++r) t[r] = -(R[3 * r] * C[0] + R[3 * r + 1] * C[1] + R[3 * r + 2] * C[2]);
    for (int r = 0; r < 3; ++r) { for (int cc = 0; cc < 3; ++cc) Rt[k][3 * r + cc] = R[3 * r + cc]; Rt[k][9 + r] = t[r]; }
    for (int r = 0; r < 3; ++r) {
      for (int cc = 0; cc < 3; ++cc) kf.Tcw[4 * r + cc] = (float)R[3 * r + cc];
      kf.Tcw[4 * r + 3] = (float)(t[r] + (k > 0 && kf.local ? 0.02 * N(rng) : 0.0));  // perturbed start for the free keyframes
    }
    kf.fx = kf.fy = 480; kf.cx = 320; kf.cy = 240; kf.bf = 40;
    const float K[9] = {480, 0, 320, 0, 480, 240, 0, 0, 1};
    for (int i = 0; i < 9; ++i) kf.K[i] = K[i];
    w.keyframes.push_back(kf);
  }
  auto project = [&](int k, const double X[3], double& u, double& v, double& z) {
    double pc[3];
    for (int r = 0; r < 3; ++r) pc[r] = Rt[k][3 * r] * X[0] + Rt[k][3 * r + 1] * X[1] + Rt[k][3 * r + 2] * X[2] + Rt[k][9 + r];
    z = pc[2]; u = 480 * pc[0] / pc[2] + 320; v = 480 * pc[1] / pc[2] + 240;
  };
  std::vector<std::vector<double>> Xtrue;
  for (int i = 0; i < n_pt; ++i) {
    const double X[3] = {-2.5 + 5.0 * U(rng), -1.2 + 2.4 * U(rng), 4.0 + 0.8 * U(rng)};
    MapPointIn p;
    p.id = 100 + i;
    for (int k = 0; k < n_kf; ++k) {
      if (U(rng) > 0.7) continue;
      double u, v, z;
      project(k, X, u, v, z);
      if (z < 0.3 || u < 1 || u > 639 || v < 1 || v > 479) continue;
      PointObsIn o;
      o.kf = k; o.u = (float)(u + 0.5 * N(rng)); o.v = (float)(v + 0.5 * N(rng));
      if (i % 4 == 0) o.u_right = (float)(o.u - 40.0 / z);
      o.inv_sigma2 = 1.0f;
      if (U(rng) < 0.03) { o.u = (float)(640 * U(rng)); o.v = (float)(480 * U(rng)); }  // gross outlier
      p.obs.push_back(o);
    }
    p.n_observations = (int)p.obs.size();
    if (p.obs.empty()) continue;
    for (int a = 0; a < 3; ++a) p.xyz[a] = (float)(X[a] + 0.03 * N(rng));
    w.points.push_back(p);
    Xtrue.push_back({X[0], X[1], X[2]});
  }
  // the back wall z = 5 as a plane (n = (0,0,-1), d = 5 in the world), observed by the first four keyframes
  MapPlaneIn pl;
  pl.id = 1; pl.coef[0] = 0.01f; pl.coef[1] = 0; pl.coef[2] = -1; pl.coef[3] = 5.02f;
  for (int k = 0; k < 4; ++k) {
    PlaneObsIn o;
    o.kf = k;
    // camera-frame plane: n_c = R n, d_c = d - t . n_c
    const double n[3] = {0, 0, -1};
    double nc[3];
    for (int r = 0; r < 3; ++r) nc[r] = Rt[k][3 * r] * n[0] + Rt[k][3 * r + 1] * n[1] + Rt[k][3 * r + 2] * n[2];
    const double dc = 5.0 - (Rt[k][9] * nc[0] + Rt[k][10] * nc[1] + Rt[k][11] * nc[2]);
    for (int a = 0; a < 3; ++a) o.coef[a] = (float)nc[a];
    o.coef[3] = (float)dc;
    pl.obs.push_back(o);
  }
  pl.asso_cuboid_id = 40; pl.asso_cuboid_meas[0] = 0.01; pl.asso_cuboid_meas[2] = -0.01;
  w.planes.push_back(pl);
  // an upright box in front of the wall, its bounding box measured in every keyframe
  MapCuboidIn cb;
  cb.id = 40; cb.qt[2] = std::sin(0.15); cb.qt[3] = std::cos(0.15); cb.qt[4] = 0.3; cb.qt[5] = 0.4; cb.qt[6] = 3.6; cb.scale[0] = 0.35; cb.scale[1] = 0.25; cb.scale[2] = 0.3;
  for (int k = 0; k < n_kf; ++k) {
    double ulo = 1e9, uhi = -1e9, vlo = 1e9, vhi = -1e9;
    const double cy = std::cos(0.3), sy = std::sin(0.3);
    CuboidObsIn o;
    o.kf = k;
    for (int c = 0; c < 8; ++c) {
      const double sx = ((c & 3) == 0 || (c & 3) == 1) ? 1.0 : -1.0, syy = ((c & 3) == 0 || (c & 3) == 3) ? 1.0 : -1.0, sz = c < 4 ? -1.0 : 1.0;
      const double b[3] = {sx * 0.35, syy * 0.25, sz * 0.3};
      const double X[3] = {cy * b[0] - sy * b[1] + 0.3, sy * b[0] + cy * b[1] + 0.4, b[2] + 3.6};
      double u, v, z;
      project(k, X, u, v, z);
      o.corners_2d[2 * c] = u; o.corners_2d[2 * c + 1] = v;
      ulo = std::fmin(ulo, u); uhi = std::fmax(uhi, u); vlo = std::fmin(vlo, v); vhi = std::fmax(vhi, v);
    }
    o.bbox_vec[0] = 0.5 * (ulo + uhi) + N(rng); o.bbox_vec[1] = 0.5 * (vlo + vhi) + N(rng); o.bbox_vec[2] = uhi - ulo; o.bbox_vec[3] = vhi - vlo;
    o.bbox_x = (int)ulo; o.bbox_y = (int)vlo; o.bbox_w = (int)(uhi - ulo); o.bbox_h = (int)(vhi - vlo);
    o.meas_quality = 0.7;
    cb.obs.push_back(o);
  }
  for (int i = 0; i < 30; ++i) {
    ObjectPointIn q;
    q.xyz[0] = (float)(0.3 + 0.3 * (U(rng) - 0.5)); q.xyz[1] = (float)(0.4 + 0.2 * (U(rng) - 0.5)); q.xyz[2] = (float)(3.6 + 0.3 * (U(rng) - 0.5));
    q.n_object_observations = 3 + i % 3;
    cb.unique_points.push_back(q);
  }
  w.cuboids.push_back(cb);

  Params par;
  par.optimize_with_plane_3d = par.optimize_with_cuboid_2d = par.optimize_with_pt_obj_3d = par.optimize_with_cuboid_plane = true;
  Flat f;
  flatten(w, par, f);
  const ppo_ba_problem& q = f.problem;
  std::printf("window: %d poses (%d fixed cameras), %d points (of %zu), %d planes, %d cuboids; edges: %d mono, %d stereo, %d plane, %d bbox, %d point-object, %d cuboid-plane\n",
              q.n_poses, 2, q.n_points, w.points.size(), q.n_planes, q.n_cuboids, q.n_mono, q.n_stereo, q.n_plane_edges, q.n_bbox, q.n_ptcub, q.n_cubpl);
  ppo_ba_options opt;
  ppo_ba_default_options(&opt);
  ppo_ba_handle h = nullptr;
  int rc = ppo_ba_create(&q, &opt, &h);
  if (rc != PPO_BA_OK) { std::printf("create failed: %s\n", ppo_ba_strerror(rc)); return 1; }
  ppo_ba_stats st;
  rc = ppo_ba_run(h, nullptr, &st);
  if (rc < 0) { std::printf("run failed: %s\n", ppo_ba_strerror(rc)); return 1; }
  std::vector<double> pose_qt(7 * q.n_poses), point_xyz(3 * q.n_points), plane_coef(4 * q.n_planes), cuboid_qt(7 * q.n_cuboids), cuboid_scale(3 * q.n_cuboids);
  std::vector<double> mono_chi2(q.n_mono), stereo_chi2(q.n_stereo), pl_chi2(q.n_plane_edges);
  std::vector<uint8_t> mono_dp(q.n_mono), stereo_dp(q.n_stereo);
  ppo_ba_result r{};
  r.pose_qt = pose_qt.data(); r.point_xyz = point_xyz.data(); r.plane_coef = plane_coef.data(); r.cuboid_qt = cuboid_qt.data(); r.cuboid_scale = cuboid_scale.data();
  r.mono_chi2 = mono_chi2.data(); r.mono_depth_positive = mono_dp.data(); r.stereo_chi2 = stereo_chi2.data(); r.stereo_depth_positive = stereo_dp.data(); r.pl_chi2 = pl_chi2.data();
  rc = ppo_ba_read(h, &r);
  if (rc != PPO_BA_OK) { std::printf("read failed: %s\n", ppo_ba_strerror(rc)); return 1; }
  ppo_ba_destroy(h);
  WriteBack wb;
  write_back(w, par, f, r, wb);
  // the optimisation must have pulled the perturbed keyframes back towards the truth
  double err0 = 0, err1 = 0;
  for (const WriteBack::Pose& p : wb.poses)
    for (int a = 0; a < 3; ++a) {
      err0 = std::fmax(err0, std::fabs((double)w.keyframes[p.keyframe].Tcw[4 * a + 3] - Rt[p.keyframe][9 + a]));
      err1 = std::fmax(err1, std::fabs((double)p.Tcw[4 * a + 3] - Rt[p.keyframe][9 + a]));
    }
  std::printf("%d LM iterations, chi2 %.1f -> %.1f; keyframe translation error %.4f -> %.4f m; %zu point and %zu plane observations to erase; %zu poses, %zu points, %zu cuboids, %zu planes written back\n",
              st.n_iters, st.iters[0].chi2_before, st.iters[st.n_iters - 1].chi2_after, err0, err1, wb.erase_points.size(), wb.erase_planes.size(), wb.poses.size(),
              wb.points.size(), wb.cuboids.size(), wb.planes.size());
  const bool ok = st.n_iters > 0 && st.iters[st.n_iters - 1].chi2_after < 0.5 * st.iters[0].chi2_before && err1 < 0.5 * err0 && wb.poses.size() == 6 &&
                  wb.points.size() == (size_t)q.n_points && !wb.erase_points.empty();
  std::printf(ok ? "flatten_demo ok\n" : "flatten_demo FAILED\n");
  return ok ? 0 : 1;
}
