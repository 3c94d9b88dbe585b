// ppo_chol.hip -- native dense Cholesky for the reduced system (placeholder until the blocked kernel lands).
#include "ppo_chol.hpp"

#include "../../include/ppo_ba.h"

namespace ppo {
int chol_factor_solve(double*, int, double*, int*, double*, hipStream_t) { return PPO_BA_ERR_SOLVER; }
size_t chol_workspace_doubles(int) { return 0; }
int chol_default_backend() { return PPO_SOLVER_ROCSOLVER; }
}  // namespace ppo
