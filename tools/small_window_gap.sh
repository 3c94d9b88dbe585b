#!/bin/bash
# development aid: kernel trace of one ICL-sized window -> per-iteration busy fraction, idle gaps, one iteration's timeline
ROOT=$PWD; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/swgap
python $ROOT/tools/small_window_trace.py ${1:-30} ${2:-2000} 40
rocprofv3 --kernel-trace --stats -d /tmp/swgap -o p -- python $ROOT/tools/small_window_trace.py ${1:-30} ${2:-2000} 40 > /tmp/swgap.log 2>&1
python $ROOT/tools/gap_report.py $(find /tmp/swgap -name "*.db" | head -1) | head -${3:-40}
