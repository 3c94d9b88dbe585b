cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
t0=$(date +%s)
for i in $(seq 1 ${1:-9}); do
  python -m pytest tests -m gpu -q --timeout 600 -k "full_size or baseline_configurations or config4 or beyond_the_resident or deterministic_mode_is_bit" 2>&1 | grep -E "^FAILED|^E  |passed|failed" | head -24 | sed "s/^/rep $i ($(( $(date +%s) - t0 )) s): /"
  [ $(( $(date +%s) - t0 )) -gt ${2:-2900} ] && break
done | tee gpurun_out/r05_flaky_hunt3.txt
