#!/bin/bash
# round 6: what produced profiles/r06_final_* -- the default bench line, kernel statistics + gap report of the same command under rocprofv3, the PMC traffic passes
# (FETCH_SIZE / WRITE_SIZE each on their own), seeds 1-5 and the other BASELINE configurations.  Every rocprofv3 call under `timeout`.
cd /tmp && export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT; [ -z "$ROOT" ] && ROOT=/root/repo
OUT=$ROOT/gpurun_out/r06_final; mkdir -p $OUT; cd $ROOT
python bench.py > $OUT/bench.log 2> $OUT/bench.err
grep "^{" $OUT/bench.log > $OUT/bench.json
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o p -- python $ROOT/bench.py --no-cpu-baseline --no-extras > $OUT/prof.log 2>&1
DB=$(find $OUT/prof -name "*.db" | head -1)
python $ROOT/tools/rocpd_summary.py $DB > $OUT/kernel_stats.txt 2>&1
python $ROOT/tools/gap_report.py $DB > $OUT/gap.txt 2>&1
rm -rf $OUT/prof
CMD="python $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras --lin-reps 20"
RE="k_linearize_all|k_lin_poses|k_lin_points|k_schur_pairs|k_reduce_stage|k_zero_ranges|k_backsub|k_update_all|k_chi2_all|k_chi2_points|chol_step|chol_back_all"
mkdir -p $OUT/pmc
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex "$RE" --output-format csv -d $OUT/pmc/fetch -o p -- $CMD > $OUT/pmc/fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --kernel-include-regex "$RE" --output-format csv -d $OUT/pmc/write -o p -- $CMD > $OUT/pmc/write.log 2>&1
python $ROOT/tools/pmc_traffic_summary.py $OUT/pmc $OUT/pmc/traffic.json $OUT/pmc/traffic.txt > $OUT/pmc/summary.log 2>&1
cd $ROOT
bash tools/seeds_and_configs.sh > $OUT/seeds_and_configs.txt 2>&1
python -c "
import json
j = json.load(open('$OUT/bench.json')); c = j['config']
print('value %.0f LM it/s, %.4f ms/iter, roofline frac %.3f (launch %.2f us), dense frac %.4f (%.3f ms), cpu baseline %.2f it/s' % (j['value'], c['ms_per_lm_iter'], j['roofline']['frac'], j['roofline']['launch_ms'] * 1e3, j['roofline_dense']['frac'], j['roofline_dense']['launch_ms'], j['cpu_baseline']['value']))
print('deterministic_mode', {k: v for k, v in c['deterministic_mode'].items() if k != 'floating_point_atomics_mode'})
print('atomics', c['deterministic_mode'].get('floating_point_atomics_mode'))
print('icl', c['icl_sized_windows'])
print('warmup %.1f ms, first create %.2f ms, steady call %s' % (c['warmup_ms_incl_module_load'], c['first_create_ms_after_warmup'], c['steady_state_call_ms']))
" 2>&1 | tee $OUT/summary.txt
head -24 $OUT/kernel_stats.txt | cut -c1-200; head -12 $OUT/gap.txt; cat $OUT/pmc/traffic.txt | head -20; cat $OUT/seeds_and_configs.txt
