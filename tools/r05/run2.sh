cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for cfg in "15 800" "30 2000" "8 400"; do
  timeout 120 python tools/small_window_trace.py $cfg 200
  PPO_HOOKS=1 PPO_BA_CHOL_SMALL=0 timeout 120 python tools/small_window_trace.py $cfg 200
  PPO_HOOKS=1 PPO_BA_CHOL_SMALL=192 timeout 120 python tools/small_window_trace.py $cfg 200
done 2>&1 | tee gpurun_out/r05_small_second.txt
timeout 1200 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "one_launch or native_cholesky or full_protocol or rejected or failed_factor or nan_residual or randomly or stop_flag" 2>&1 | tail -15
