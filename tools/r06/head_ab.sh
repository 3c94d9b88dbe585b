#!/bin/bash
# round 6: the head of a pipelined pass on the host (a: k_ctl_set after a round trip) against on the device (b: HeadInit in the hand-over launch), libraries in
# tools/r06/ab/: headline rate, three alternating rounds, and the run time of the ICL-sized windows on a resident handle (tools/small_window_trace.py)
ROOT=$PWD; cd /tmp && export TMPDIR=/tmp
for round in 1 2 3; do
  for v in $ROOT/tools/r06/ab/*.so; do
    echo "round $round $(basename $v): $(PPO_BA_LIB=$v python $ROOT/bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3 2>/dev/null | grep -o '"value": [0-9.]*')"
  done
done
for round in 1 2; do
  for v in $ROOT/tools/r06/ab/*.so; do
    for kf in "15 800" "30 2000"; do
      echo "round $round $(basename $v): $(PPO_BA_LIB=$v python $ROOT/tools/small_window_trace.py $kf 60 | head -1)"
    done
  done
done
