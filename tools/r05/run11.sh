cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python tools/create_breakdown.py 15 800 2>&1 | head -1
python tools/create_breakdown.py 30 2000 2>&1 | head -1
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_dump_replay.py tests/test_orbslam_63_pin.py -m gpu -x -q --timeout 300 -k "full_protocol or device_built or block_cache or order_of or randomly or landmark_sharding or cpp_caller or builder or replay or zero_iter or empty or 63 or fixed_points" 2>&1 | tail -3
python tools/small_window_times.py 2>&1 | tail -5 | tee gpurun_out/r05_small_window_times.txt
