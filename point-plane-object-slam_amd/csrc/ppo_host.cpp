// ppo_host.cpp -- host-only entry points of libppo_ba.so: no device work, usable without a GPU.
#include <algorithm>
#include <cstring>
#include <vector>

#include "../../include/ppo_ba.h"

extern "C" {

const char* ppo_ba_strerror(int code) {
  switch (code) {
    case PPO_BA_OK: return "ok";
    case PPO_BA_ABORTED: return "aborted by stop flag";
    case PPO_BA_ERR_ARG: return "bad argument";
    case PPO_BA_ERR_NONFINITE: return "non-finite input";
    case PPO_BA_ERR_NOMEM: return "out of memory";
    case PPO_BA_ERR_HIP: return "HIP runtime / device failure";
    case PPO_BA_ERR_SOLVER: return "dense solver backend failure";
    case PPO_BA_ERR_COMM: return "RCCL failure";
    case PPO_BA_ERR_STATE: return "call order";
  }
  return "unknown";
}

// constants of the reference: Optimizer.cc:2728,2837; optimization_algorithm_levenberg.cpp:47-51;
// base_binary_edge.hpp:232; Optimizer.cc:2747,2763; src/Parameters.cc:58-70; Optimizer.cc:2204
void ppo_ba_default_options(ppo_ba_options* o) {
  if (!o) return;
  std::memset(o, 0, sizeof(*o));
  o->abi_version = PPO_BA_ABI_VERSION;
  o->iters_pass1 = 5;
  o->iters_pass2 = 10;
  o->lm_tau = 1e-5;
  o->lm_max_trials = 10;
  o->lm_lower = 1. / 3.;
  o->lm_upper = 2. / 3.;
  o->numeric_delta = 1e-9;
  o->jacobian_mode = PPO_JAC_NUMERIC;
  o->chi2_mono = 5.991;
  o->chi2_stereo = 7.815;
  o->chi2_plane = 500.0;
  o->chi2_plane_vp = 200.0;
  o->th_bbox_norm = 80.0;
  o->th_corner_norm = 10.0;
  o->th_cub3d_norm = 900.0;  /* thHuberSE3, src/Parameters.cc:65 */
  o->solve_on_owner = 0;
  o->lm_control = PPO_LM_DEVICE;
  o->deterministic = 1;  /* round 6: the bit-reproducible mode is the default (within 1 % of the floating-point atomics) */
  o->solver = PPO_SOLVER_AUTO;
  o->eliminate_cuboids = 1;
  o->world_size = 1;
}

// Contiguous landmark ranges with balanced edge counts: points are split so that every rank gets about the same
// number of reprojection edges, planes so that every rank gets about the same number of plane edges.
int ppo_ba_partition(const ppo_ba_problem* p, int32_t world, int32_t* first_point, int32_t* first_plane) {
  if (!p || world <= 0 || !first_point || !first_plane) return PPO_BA_ERR_ARG;
  auto split = [&](int n_items, const std::vector<long>& weight, int32_t* first) {
    std::vector<long> prefix(n_items + 1, 0);
    for (int i = 0; i < n_items; ++i) prefix[i + 1] = prefix[i] + weight[i] + 1;  // +1 keeps edge-less items spread
    const long total = prefix[n_items];
    first[0] = 0;
    for (int r = 1; r < world; ++r) {
      const long target = (total * r) / world;
      first[r] = (int32_t)(std::lower_bound(prefix.begin(), prefix.end(), target) - prefix.begin());
      if (first[r] > n_items) first[r] = n_items;
      if (first[r] < first[r - 1]) first[r] = first[r - 1];
    }
    first[world] = n_items;
  };
  std::vector<long> wp(std::max(p->n_points, 0), 0), wl(std::max(p->n_planes, 0), 0);
  for (int e = 0; e < p->n_mono; ++e) {
    if (p->mono_point[e] < 0 || p->mono_point[e] >= p->n_points) return PPO_BA_ERR_ARG;
    wp[p->mono_point[e]]++;
  }
  for (int e = 0; e < p->n_stereo; ++e) {
    if (p->stereo_point[e] < 0 || p->stereo_point[e] >= p->n_points) return PPO_BA_ERR_ARG;
    wp[p->stereo_point[e]]++;
  }
  for (int e = 0; e < p->n_plane_edges; ++e) {
    if (p->pl_plane[e] < 0 || p->pl_plane[e] >= p->n_planes) return PPO_BA_ERR_ARG;
    wl[p->pl_plane[e]]++;
  }
  split(p->n_points, wp, first_point);
  split(p->n_planes, wl, first_plane);
  return PPO_BA_OK;
}

int64_t ppo_ba_sizeof(int32_t which) {
  switch (which) {
    case 0: return (int64_t)sizeof(ppo_ba_problem);
    case 1: return (int64_t)sizeof(ppo_ba_options);
    case 2: return (int64_t)sizeof(ppo_ba_stats);
    case 3: return (int64_t)sizeof(ppo_ba_result);
    case 4: return (int64_t)sizeof(ppo_ba_iter_stat);
  }
  return -1;
}

}  // extern "C"
