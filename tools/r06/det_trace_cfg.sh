#!/bin/bash
# per-kernel statistics of one configuration, atomics mode (det 0) against deterministic mode (det 1).  usage: det_trace_cfg.sh cfg3
CFG=${1:-cfg3}; ROOT=$PWD; cd /tmp && export TMPDIR=/tmp
for det in 0 1; do
  rm -rf /tmp/dtc$det
  timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/dtc$det -o p -- python $ROOT/bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 --config $CFG --deterministic $det > /tmp/dtc.log 2>&1
  echo "== $CFG deterministic=$det: $(grep -o '"value": [0-9.]*' /tmp/dtc.log | head -1)"
  python $ROOT/tools/rocpd_summary.py $(find /tmp/dtc$det -name "*.db" | head -1) | head -12 | awk '{printf "   %-64s calls %6s avg %7s us\n", substr($1,1,64), $2, $4}'
  python $ROOT/tools/gap_report.py $(find /tmp/dtc$det -name "*.db" | head -1) | head -1
done
