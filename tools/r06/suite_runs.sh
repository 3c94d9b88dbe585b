#!/bin/bash
# N consecutive runs of the GPU suite on one box; every run's summary line AND the names of any failures (round 5 lost the name of its one red run)
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-3}; TAG=${2:-a}
for i in $(seq 1 $N); do
  python -m pytest tests -m gpu -q --timeout 900 2>&1 | grep -E "^FAILED|^ERROR|passed|failed" | tail -12 | sed "s/^/run $TAG$i: /"
done | tee gpurun_out/r06_suite_runs_$TAG.txt
