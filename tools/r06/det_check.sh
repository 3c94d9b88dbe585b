#!/bin/bash
# round 6 development aid: the deterministic mode on the pipelined / overlapped path -- the tests that hold it, then the headline bench line condensed
# (value = default mode, det = options.deterministic = 1 on the same window).  Usage: gpurun -- 'bash tools/r06/det_check.sh [pytest -k expression]'
K=${1:-"deterministic or overlapped or device_built or builders or rejected_trials or failed_factorisation or nan_residual or stop_flag or full_protocol or linearisation_matches or native_cholesky or one_launch"}
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -k "$K" 2>&1 | tail -15 | tee gpurun_out/r06_det_check_tests.txt
python bench.py --no-cpu-baseline --steps 10 2>gpurun_out/r06_det_check_bench.err | tee gpurun_out/r06_det_check_bench.json | python -c "
import sys, json
j = json.loads(sys.stdin.read())
d = j['config'].get('deterministic_mode') or {}
o = d.get('floating_point_atomics_mode') or {}
print('value (deterministic, default) %.0f it/s (%.4f ms/iter) | atomics mode %.0f it/s | det/atomics %.3f | lin %.1f us | factor %.3f ms' % (
    j['value'], j['config']['ms_per_lm_iter'], o.get('lm_iters_per_s', 0), j['value'] / max(o.get('lm_iters_per_s', 1), 1),
    j['roofline']['launch_ms'] * 1e3, j['config']['factor_solve_ms']))
print('atomics-vs-det deviation', o.get('its_run_deviation_from_the_timed_run'))
print('icl', j['config'].get('icl_sized_windows'))
"
tail -5 gpurun_out/r06_det_check_bench.err
