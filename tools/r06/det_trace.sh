#!/bin/bash
# round 6: per-kernel statistics and one iteration's timeline of the BASELINE window, default mode and deterministic mode side by side
ROOT=$PWD; OUT=$ROOT/gpurun_out/r06_det_trace; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
for det in 0 1; do
  rm -rf /tmp/dt$det
  rocprofv3 --kernel-trace --stats -d /tmp/dt$det -o p -- python $ROOT/bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 --deterministic $det > $OUT/bench_det$det.log 2>&1
  DB=$(find /tmp/dt$det -name "*.db" | head -1)
  python $ROOT/tools/rocpd_summary.py $DB > $OUT/kernel_stats_det$det.txt 2>&1
  python $ROOT/tools/gap_report.py $DB > $OUT/gap_det$det.txt 2>&1
done
head -30 $OUT/kernel_stats_det0.txt; head -30 $OUT/kernel_stats_det1.txt; head -3 $OUT/gap_det0.txt; head -3 $OUT/gap_det1.txt
