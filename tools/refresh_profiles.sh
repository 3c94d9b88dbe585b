set -x
cd /tmp && export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/final; mkdir -p $OUT
python $ROOT/bench.py > $OUT/bench.log 2>&1
grep "^{" $OUT/bench.log > $OUT/bench.json
rocprofv3 --kernel-trace --stats -d $OUT/prof -o p -- python $ROOT/bench.py --no-cpu-baseline --no-extras > $OUT/prof.log 2>&1
DB=$(find $OUT/prof -name "*.db" | head -1)
python $ROOT/tools/rocpd_summary.py $DB > $OUT/kernel_stats.txt 2>&1
python $ROOT/tools/gap_report.py $DB > $OUT/gap.txt 2>&1
bash $ROOT/tools/pmc_traffic.sh $OUT/pmc > $OUT/pmc.log 2>&1
ls -la $OUT $OUT/pmc | head -40
