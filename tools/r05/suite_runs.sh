# N consecutive runs of the GPU suite on one box: summary line of each into gpurun_out/r05_suite_runs_<tag>.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
N=${1:-5}; TAG=${2:-a}
for i in $(seq 1 $N); do
  python -m pytest tests -m gpu -q --timeout 600 2>&1 | grep -E "^FAILED|passed|failed|error" | tail -4 | sed "s/^/run $TAG$i: /"
done | tee gpurun_out/r05_suite_runs_$TAG.txt
