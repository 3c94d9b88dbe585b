#!/bin/bash
# round 6: A/B of library builds in tools/r06/ab/*.so (PPO_BA_LIB selects the library the harness loads): headline rate (3 alternating rounds) and the average
# duration of the kernels matching $1 from one kernel trace per build.  Every rocprofv3 call under `timeout` (a counter pass that aborted once hung for 20 min).
RE="${1:-k_update_all|k_linearize_all}"; ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
for round in 1 2 3; do
  for v in $ROOT/tools/r06/ab/*.so; do
    echo "round $round $(basename $v): $(PPO_BA_LIB=$v python $ROOT/bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3 2>/dev/null | grep -o '"value": [0-9.]*')"
  done
done
for v in $ROOT/tools/r06/ab/*.so; do
  rm -rf /tmp/abk; PPO_BA_LIB=$v timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/abk -o p -- python $ROOT/bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 > /tmp/abk.log 2>&1
  echo "== $(basename $v)"
  python $ROOT/tools/rocpd_summary.py $(find /tmp/abk -name "*.db" | head -1) | grep -E "$RE" | awk '{printf "   %-60s calls %s avg %s us\n", substr($1,1,60), $2, $4}'
done
