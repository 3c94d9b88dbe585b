cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
t0=$(date +%s)
for i in $(seq 1 ${1:-20}); do
  python -m pytest tests -m gpu -q --timeout 300 -k "not full_size and not baseline_configurations and not config4 and not beyond_the_resident and not deterministic_mode_is_bit" 2>&1 | grep -E "^FAILED|^E  |passed|failed" | head -14 | sed "s/^/rep $i ($(( $(date +%s) - t0 )) s): /"
  [ $(( $(date +%s) - t0 )) -gt ${2:-1500} ] && break
done | tee gpurun_out/r05_flaky_hunt2.txt | grep -v " passed" | head -60
tail -3 gpurun_out/r05_flaky_hunt2.txt
