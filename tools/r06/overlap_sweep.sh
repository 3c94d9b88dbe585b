#!/bin/bash
# round 6: the schedule of the overlapped reduction in deterministic mode (no atomic trailing tiles any more: do larger shares per launch pay now?) -- test-hooks build,
# PPO_BA_OVERLAP="first_blocks,share0,share[,min_chunks[,min_panels]]"; two rounds, alternating
export PPO_BA_LIB=$PWD/point-plane-object-slam_amd/csrc/libppo_ba_test.so
for round in 1 2; do
  for ov in "1,0.14,0.10" "1,0.20,0.14" "1,0.10,0.07" "2,0.14,0.10" "1,0.14,0.20" "1,0.30,0.10" "1,0.05,0.10"; do
    echo "round $round PPO_BA_OVERLAP=$ov: $(PPO_BA_OVERLAP=$ov python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3 2>/dev/null | grep -o '"value": [0-9.]*')"
  done
done
