cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 tools/run_chol_small_check.sh > gpurun_out/r05_chol_small_check.txt 2>&1; echo "check rc $?"
tail -40 gpurun_out/r05_chol_small_check.txt
for cfg in "15 800" "30 2000"; do
  timeout 120 python tools/small_window_trace.py $cfg 200
  PPO_HOOKS=1 PPO_BA_CHOL_SMALL=0 timeout 120 python tools/small_window_trace.py $cfg 200
  PPO_HOOKS=1 timeout 120 python tools/small_window_trace.py $cfg 200
done 2>&1 | tee gpurun_out/r05_small_first.txt
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "native_cholesky or full_protocol or rejected or failed_factor or nan_residual or randomly or stop_flag" 2>&1 | tail -15
