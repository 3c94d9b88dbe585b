cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 300 > gpurun_out/r05_pytest4.txt 2>&1; tail -5 gpurun_out/r05_pytest4.txt
timeout 600 python bench.py > gpurun_out/r05_bench4.json 2> gpurun_out/r05_bench4.err; tail -c 600 gpurun_out/r05_bench4.err
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r05_bench4.json').read().strip().splitlines()[-1])
c=j['config']
print(j['value'], c['ms_per_lm_iter'], c['factor_solve_ms'], c.get('warmup_ms_incl_module_load'), c.get('first_create_ms_after_warmup'))
print(json.dumps(c.get('icl_sized_windows'))[:1500])
PY
