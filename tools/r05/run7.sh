cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_orbslam_63_pin.py tests/test_lm_control.py tests/test_parity_gpu.py -m gpu -x -q --timeout 300 -k "63 or reference or decision or randomly or default_mode or warmup or lm_ or stop" 2>&1 | tail -4
timeout 1200 python tools/strict_sweep.py 300 > gpurun_out/r05_strict_sweep.txt 2>&1; tail -3 gpurun_out/r05_strict_sweep.txt
timeout 600 python bench.py > gpurun_out/r05_bench7.json 2> gpurun_out/r05_bench7.err
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r05_bench7.json').read().strip().splitlines()[-1])
c=j['config']
print(j['value'], c['ms_per_lm_iter'], c['factor_solve_ms'], c.get('warmup_ms_incl_module_load'), c.get('first_create_ms_after_warmup'))
print(json.dumps(c.get('deterministic_mode')))
print(json.dumps(c.get('icl_sized_windows'))[:900])
PY
timeout 600 tools/scale_run.sh r05_smoke 5 2>&1 | tail -12; mv profiles/r05_smoke_scale.txt gpurun_out/ 2>/dev/null
