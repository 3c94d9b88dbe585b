cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python __graft_entry__.py smoke 2>&1 | tail -3
timeout 300 python -m pytest tests/test_parity_gpu.py tests/test_abi.py -m gpu -x -q --timeout 200 -k "warmup or block_cache or full_protocol or one_launch" 2>&1 | tail -2
timeout 300 python bench.py 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=j['config']
print('value %.0f, ms/iter %.4f, warmup %.1f ms, first create after warmup %.2f ms, frac %.3f' % (j['value'], c['ms_per_lm_iter'], c['warmup_ms_incl_module_load'], c['first_create_ms_after_warmup'], j['roofline']['frac']))
print('traffic source', j['roofline']['traffic_source'], j['roofline']['traffic'])
"
