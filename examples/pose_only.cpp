// examples/pose_only.cpp -- the pose-only optimisation (include/ppo_pose.h) from a plain C++ caller: synthetic frames the way the
// Tracking thread hands them over, one call per frame like Optimizer::PoseOptimization (src/Optimizer.cc:247-459), then one
// call for a whole batch.  Built by csrc/Makefile with g++ against libppo_ba.so; no HIP headers, no torch.
//   examples/pose_only [observations per frame = 400] [frames = 64]
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "ppo_pose.h"

int main(int argc, char** argv) {
  const int n_obs = argc > 1 ? std::atoi(argv[1]) : 400, n_frames = argc > 2 ? std::atoi(argv[2]) : 64;
  std::mt19937_64 rng(7);
  std::normal_distribution<double> gauss(0.0, 1.0);
  std::uniform_real_distribution<double> uni(0.0, 1.0);
  const double fx = 480, fy = 480, cx = 320, cy = 240, bf = 40;
  std::vector<double> pose, truth, cam, xw, obs, inv;
  std::vector<int32_t> off(1, 0);
  for (int f = 0; f < n_frames; ++f) {
    // true pose: a small rotation about a random axis (quaternion x y z w) and a translation; the estimate is off by ~2 cm
    double a[3] = {0.2 * gauss(rng), 0.2 * gauss(rng), 0.2 * gauss(rng)};
    const double th = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]) + 1e-12, s = std::sin(th / 2) / th;
    const double q[4] = {s * a[0], s * a[1], s * a[2], std::cos(th / 2)}, t[3] = {gauss(rng), gauss(rng), gauss(rng)};
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 2 * (x * y + z * w), 1 - 2 * (x * x + z * z),
                         2 * (y * z - x * w), 2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)};
    for (int k = 0; k < 4; ++k) { truth.push_back(q[k]); pose.push_back(q[k]); }
    for (int k = 0; k < 3; ++k) { truth.push_back(t[k]); pose.push_back(t[k] + 0.02 * gauss(rng)); }
    const double c[5] = {fx, fy, cx, cy, bf};
    cam.insert(cam.end(), c, c + 5);
    for (int i = 0; i < n_obs; ++i) {
      const double zc = 0.5 + 7.5 * uni(rng), pc[3] = {(2 * uni(rng) - 1) * zc, (1.6 * uni(rng) - 0.8) * zc, zc};
      double X[3];  // Xw = R^T (pc - t), kept in float like the reference's map points
      for (int k = 0; k < 3; ++k) X[k] = (double)(float)(R[k] * (pc[0] - t[0]) + R[3 + k] * (pc[1] - t[1]) + R[6 + k] * (pc[2] - t[2]));
      const bool gross = uni(rng) < 0.1, stereo = uni(rng) < 0.5;
      double u = fx * pc[0] / pc[2] + cx + gauss(rng), v = fy * pc[1] / pc[2] + cy + gauss(rng);
      if (gross) { u = 640 * uni(rng); v = 480 * uni(rng); }
      const double ur = stereo ? u - bf / pc[2] + gauss(rng) : -1.0;
      xw.insert(xw.end(), X, X + 3);
      obs.push_back((double)(float)u); obs.push_back((double)(float)v); obs.push_back((double)(float)ur);
      inv.push_back(1.0);
    }
    off.push_back(off.back() + n_obs);
  }
  ppo_pose_options O;
  ppo_pose_default_options(&O);  // 4 rounds x 10 iterations, chi2 5.991 / 7.815, Huber off after round 2
  std::vector<double> out(7 * (size_t)n_frames);
  std::vector<uint8_t> outlier(xw.size() / 3);
  std::vector<int32_t> inliers(n_frames);
  auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  // one frame per call, like the reference
  double worst = 0, total = 0;
  for (int f = 0; f < n_frames; ++f) {
    const int32_t o2[2] = {0, n_obs};
    ppo_pose_problem P{1, &pose[7 * (size_t)f], &cam[5 * (size_t)f], o2, &xw[3 * (size_t)off[f]], &obs[3 * (size_t)off[f]], &inv[off[f]]};
    ppo_pose_result R{&out[7 * (size_t)f], &outlier[off[f]], &inliers[f], nullptr, nullptr};
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = ppo_pose_optimize(&P, &O, &R);
    const auto t1 = std::chrono::steady_clock::now();
    if (rc != PPO_BA_OK) { std::fprintf(stderr, "ppo_pose_optimize: %s\n", ppo_ba_strerror(rc)); return 1; }  // PPO_BA_ERR_HIP without a GPU
    if (f > 0) { total += ms(t0, t1); worst = std::fmax(worst, ms(t0, t1)); }
  }
  double err = 0;
  for (int f = 0; f < n_frames; ++f)
    for (int k = 4; k < 7; ++k) err = std::fmax(err, std::fabs(out[7 * (size_t)f + k] - truth[7 * (size_t)f + k]));
  std::printf("one frame per call: %.3f ms per call (worst %.3f), %d observations each, inliers of frame 0: %d, max |t - truth| %.4f m\n",
              total / std::max(1, n_frames - 1), worst, n_obs, inliers[0], err);
  // all frames in one call
  ppo_pose_problem P{n_frames, pose.data(), cam.data(), off.data(), xw.data(), obs.data(), inv.data()};
  std::vector<double> out2(out.size());
  ppo_pose_result R{out2.data(), nullptr, nullptr, nullptr, nullptr};
  const auto t0 = std::chrono::steady_clock::now();
  const int rc = ppo_pose_optimize(&P, &O, &R);
  const auto t1 = std::chrono::steady_clock::now();
  if (rc != PPO_BA_OK) return 1;
  bool same = true;
  for (size_t i = 0; i < out.size(); ++i) same = same && out[i] == out2[i];
  std::printf("%d frames in one call: %.3f ms (%.0f frames/s); poses %s the per-frame calls\n", n_frames, ms(t0, t1), n_frames / ms(t0, t1) * 1e3,
              same ? "bit-identical to" : "DIFFERENT from");
  ppo_ba_release_cached();
  return same ? 0 : 2;
}
