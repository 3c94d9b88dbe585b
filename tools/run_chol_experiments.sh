#!/bin/bash
# development aid: what bounds a panel step of the native Cholesky?  Timing-only builds of tools/chol_check.hip (results are WRONG in
# the experiment builds): the register factorisation without its extra rows (half of its multiply-adds), without the MFMA jobs
# that run in its shadow on the side waves, without both.
cd "$(dirname "$0")"
for flags in ${CHOL_FLAGS:-"" "-DPPO_CHOL_EXPERIMENT_NO_LOOKAHEAD" "-DPPO_CHOL_EXPERIMENT_NO_E" "-DPPO_CHOL_EXPERIMENT_NO_SIDE" "-DPPO_CHOL_TRAIL_DELAY_TICKS=300" "-DPPO_CHOL_TRAIL_DELAY_TICKS=500" "-DPPO_CHOL_TRAIL_DELAY_TICKS=700"}; do
  /opt/rocm/bin/hipcc -w -O3 -std=c++17 --offload-arch=gfx950 -DPPO_BA_TEST_HOOKS $flags -munsafe-fp-atomics -ffp-contract=off -mllvm -amdgpu-kernarg-preload-count=16 -o /tmp/chol_check chol_check.hip
  echo "== flags: $flags"
  timeout 120 /tmp/chol_check panels 2>&1 | grep -E "^n (1194|1644| 600)"
done
# round 5: the pair bound -- CHOL_FLAGS='"" "-DPPO_CHOL_EXPERIMENT_PAIR_BOUND=1" "-DPPO_CHOL_EXPERIMENT_PAIR_BOUND=2"' tools/run_chol_experiments.sh
# (=1: two panel chains per launch, nothing more loaded or stored: the upper bound of what two panels per launch can save;
#  =2: the same with the redundant block-row work of a real pair design in the shadow of the first chain)
