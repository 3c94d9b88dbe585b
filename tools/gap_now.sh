#!/bin/bash
# development aid: kernel trace of the headline workload -> idle gaps per iteration and one iteration's timeline
ROOT=$PWD; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/gapnow
rocprofv3 --kernel-trace --stats -d /tmp/gapnow -o p -- python $ROOT/bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 > /tmp/gapnow.log 2>&1
python $ROOT/tools/gap_report.py $(find /tmp/gapnow -name "*.db" | head -1) | head -14
