cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_lm_control.py tests/test_orbslam_lm_pin.py -m gpu -x -q --timeout 200 -k "full_protocol or rejected or failed_factor or nan_residual or randomly or stop_flag or lm or warmup or one_launch" 2>&1 | tail -4
for i in 1 2; do
for cfg in "15 800" "30 2000"; do
  PPO_HOOKS=1 timeout 120 python tools/small_window_trace.py $cfg 200
  PPO_HOOKS=1 PPO_BA_DECIDE_TAIL=0 timeout 120 python tools/small_window_trace.py $cfg 200
done; done 2>&1 | tee gpurun_out/r05_decide_tail.txt
python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('tail: %.0f it/s, %.4f ms/iter, lin %.1f us' % (j['value'], j['config']['ms_per_lm_iter'], j['roofline']['launch_ms']*1e3))" | tee -a gpurun_out/r05_decide_tail.txt
python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('tail: %.0f it/s, %.4f ms/iter, lin %.1f us' % (j['value'], j['config']['ms_per_lm_iter'], j['roofline']['launch_ms']*1e3))" | tee -a gpurun_out/r05_decide_tail.txt
