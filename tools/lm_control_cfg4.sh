cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for lm in 0 1 0 1; do python $R/bench.py --steps 40 --no-cpu-baseline --no-extras --lm-control $lm 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lm_control', '$lm', 'value', round(d['value'],1), 'ms_per_step', round(d['ms_per_step'],3), 'factor_solve_ms', round(d['config']['factor_solve_ms'],4))"; done
for lm in 0 1; do rm -rf /tmp/prof_lm$lm; rocprofv3 --kernel-trace --stats -d /tmp/prof_lm$lm -o p -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --lm-control $lm > /dev/null 2>&1; f=$(find /tmp/prof_lm$lm -name "*kernel_stats.csv" | head -1); echo "== lm_control=$lm kernel stats"; python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print("%-70s calls %6s avg %8.2f us total %9.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e3))
PY
done
