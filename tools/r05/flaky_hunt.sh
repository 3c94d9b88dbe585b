cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for i in $(seq 1 ${1:-40}); do
  python -m pytest tests/test_lm_control.py tests/test_parity_gpu.py tests/test_orbslam_63_pin.py -m gpu -q --timeout 300 -k "decision or default_mode_with_the_lm or warmup or one_launch or device_and_host_control" 2>&1 | grep -E "^FAILED|^E  |passed|failed" | head -12 | sed "s/^/rep $i: /"
done | grep -v " passed" | head -80
echo hunt done
