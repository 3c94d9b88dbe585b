#!/bin/bash
# GPU parity suite (summary line AND the names of any failures) + the headline bench line condensed (development aid)
python -m pytest tests -m gpu -q 2>&1 | grep -E "^FAILED|^ERROR|passed|failed" | tail -12
python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('%.0f it/s, %.4f ms/iter, lin %.1f us (frac %.3f), factor %.3f ms' % (j['value'], j['config']['ms_per_lm_iter'], j['roofline']['launch_ms']*1e3, j['roofline']['frac'], j['config']['factor_solve_ms']), {k:round(v['us'],1) for k,v in j['config']['kernels'].items()})"
