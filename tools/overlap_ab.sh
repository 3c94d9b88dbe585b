#!/bin/bash
# A/B of the overlapped reduction (test-hooks build): PPO_BA_OVERLAP=0 switches it off, "first_blocks,share0,share[,min_chunks]" sets its schedule
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PPO_BA_LIB=$PWD/point-plane-object-slam_amd/csrc/libppo_ba_test.so
CFG=${CFG:-cfg4}
for f in ${VARIANTS:-0 default}; do
  if [ "$f" = default ]; then unset PPO_BA_OVERLAP; else export PPO_BA_OVERLAP=$f; fi
  for rep in 1 2; do
    python bench.py --config $CFG --steps 30 --warmup 3 --no-cpu-baseline --no-extras 2>>gpurun_out/overlap_ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$CFG overlap=%s' % '$f', 'value', round(d['value'],1), 'ms/iter', round(d['config']['ms_per_lm_iter'],4), 'factor_ms', round(d['config']['factor_solve_ms'],4))"
  done
done | tee -a gpurun_out/overlap_ab.txt
