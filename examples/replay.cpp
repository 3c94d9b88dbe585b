// replay.cpp -- the C ABI of libppo_ba.so used from C++ the way the adapter inside the reference's Optimizer.cc would
// (INTEGRATION.md): load a window captured with ppo_ba_problem_save(), create / run / read / destroy, print the LM trace.
//
// Built by point-plane-object-slam_amd/csrc/Makefile with plain g++ (no HIP headers needed on the caller's side).
//   examples/replay window.ppoba [repeats]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ppo_ba.h"

int main(int argc, char** argv) {
  if (argc < 2) {
    std::fprintf(stderr, "usage: %s window.ppoba [repeats]\n", argv[0]);
    return 2;
  }
  const int repeats = argc > 2 ? std::atoi(argv[2]) : 1;
  ppo_ba_problem* P = nullptr;
  int rc = ppo_ba_problem_load(argv[1], &P);
  if (rc != PPO_BA_OK) {
    std::fprintf(stderr, "cannot load %s: %s\n", argv[1], ppo_ba_strerror(rc));
    return 1;
  }
  std::printf("%d keyframes, %d points, %d planes, %d cuboids; %d mono + %d stereo + %d plane + %d bbox edges\n", P->n_poses, P->n_points, P->n_planes,
              P->n_cuboids, P->n_mono, P->n_stereo, P->n_plane_edges, P->n_bbox);
  ppo_ba_options O;
  ppo_ba_default_options(&O);  // the reference's constants: 5 + 10 iterations, chi2 thresholds, delta = 1e-9
  if (std::getenv("PPO_BA_VERBOSE")) O.verbose = 1;  // phase times of ppo_ba_create on stderr
  for (int rep = 0; rep < repeats; ++rep) {
    const auto t0 = std::chrono::steady_clock::now();
    ppo_ba_handle h = nullptr;
    rc = ppo_ba_create(P, &O, &h);
    if (rc != PPO_BA_OK) {
      std::fprintf(stderr, "ppo_ba_create: %s\n", ppo_ba_strerror(rc));  // PPO_BA_ERR_HIP without a GPU: there is no CPU path
      ppo_ba_problem_free(P);
      return 1;
    }
    const auto t1 = std::chrono::steady_clock::now();
    volatile uint8_t stop = 0;  // the reference's pbStopFlag
    ppo_ba_stats S;
    rc = ppo_ba_run(h, &stop, &S);
    const auto t2 = std::chrono::steady_clock::now();
    std::vector<double> pose(7 * (size_t)P->n_poses), pts(3 * (size_t)P->n_points), chi2(P->n_mono);
    std::vector<uint8_t> level(P->n_mono), depth_ok(P->n_mono);
    ppo_ba_result R{};
    R.pose_qt = pose.data();
    R.point_xyz = pts.data();
    R.mono_chi2 = chi2.data();
    R.mono_level = level.data();
    R.mono_depth_positive = depth_ok.data();
    const int rr = ppo_ba_read(h, &R);
    const auto t3 = std::chrono::steady_clock::now();
    ppo_ba_destroy(h);
    if (rc < 0 || rr != PPO_BA_OK) {
      std::fprintf(stderr, "run/read failed: %s / %s\n", ppo_ba_strerror(rc), ppo_ba_strerror(rr));
      ppo_ba_problem_free(P);
      return 1;
    }
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    if (rep == 0)
      for (int i = 0; i < S.n_iters && i < PPO_BA_MAX_ITER_STATS; ++i)
        std::printf("  pass %d iter %2d  trials %d  chi2 %.6e -> %.6e  lambda %.3e\n", S.iters[i].pass, S.iters[i].iteration, S.iters[i].trials,
                    S.iters[i].chi2_before, S.iters[i].chi2_after, S.iters[i].lambda);
    int n_out = 0;
    for (int e = 0; e < P->n_mono; ++e) n_out += level[e] || chi2[e] > O.chi2_mono || !depth_ok[e];  // Optimizer.cc:2840-2905 erase list
    std::printf("call %d: create %.2f ms, run %.2f ms (%d LM iterations), read %.2f ms; %d of %d mono observations to erase; pose[1].t = %.6f %.6f %.6f\n", rep,
                ms(t0, t1), ms(t1, t2), S.n_iters, ms(t2, t3), n_out, P->n_mono, P->n_poses > 1 ? pose[11] : 0.0, P->n_poses > 1 ? pose[12] : 0.0,
                P->n_poses > 1 ? pose[13] : 0.0);
  }
  ppo_ba_problem_free(P);
  ppo_ba_release_cached();
  return 0;
}
