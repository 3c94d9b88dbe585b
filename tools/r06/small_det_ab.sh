#!/bin/bash
# round 6: ICL-sized windows, deterministic (default) against floating-point-atomics mode: run time on one handle, per-kernel averages from a kernel trace
ROOT=$PWD; cd /tmp && export TMPDIR=/tmp
for kf in "15 800" "30 2000"; do
  for det in 1 0; do
    PPO_DET=$det python $ROOT/tools/small_window_trace.py $kf 60
    rm -rf /tmp/sw$det; PPO_DET=$det timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/sw$det -o p -- python $ROOT/tools/small_window_trace.py $kf 40 > /tmp/sw.log 2>&1
    python $ROOT/tools/rocpd_summary.py $(find /tmp/sw$det -name "*.db" | head -1) | head -12 | awk '{printf "   %-58s calls %6s avg %7s us\n", substr($1,1,58), $2, $4}'
    python $ROOT/tools/gap_report.py $(find /tmp/sw$det -name "*.db" | head -1) | head -1
  done
done
