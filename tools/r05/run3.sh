cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 700 python -m pytest tests/test_parity_gpu.py -m gpu -x -v --timeout 100 -k "one_launch or native_cholesky or full_protocol or rejected or failed_factor or nan_residual or randomly or stop_flag" --durations=15 > gpurun_out/r05_pytest3.txt 2>&1
tail -40 gpurun_out/r05_pytest3.txt
