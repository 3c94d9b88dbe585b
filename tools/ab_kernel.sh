#!/bin/bash
# development aid: average duration of kernels matching $1 (regex) for every library build in build/*.so, from rocprofv3 kernel traces
RE="$1"; ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
for v in $ROOT/build/*.so; do
  rm -rf /tmp/abk; PPO_BA_LIB=$v rocprofv3 --kernel-trace --stats -d /tmp/abk -o p -- python $ROOT/bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 > /tmp/abk.log 2>&1
  echo "== $(basename $v): $(grep -o '"value": [0-9.]*' /tmp/abk.log | head -1)"
  python $ROOT/tools/rocpd_summary.py $(find /tmp/abk -name "*.db" | head -1) | grep -E "$RE" | awk '{printf "   %-60s calls %s avg %s us\n", substr($1,1,60), $2, $4}'
done
