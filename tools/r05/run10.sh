cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
bash tools/refresh_profiles.sh > gpurun_out/r05_refresh.log 2>&1
for cfg in "15 800" "30 2000" "8 400"; do
  timeout 120 python tools/small_window_trace.py $cfg 200
done > gpurun_out/final/small_windows.txt 2>&1
bash tools/small_window_gap.sh 15 800 >> gpurun_out/final/small_windows.txt 2>&1
bash tools/small_window_gap.sh 30 2000 >> gpurun_out/final/small_windows.txt 2>&1
python tools/small_window_times.py >> gpurun_out/final/small_windows.txt 2>&1
tail -30 gpurun_out/final/small_windows.txt; head -40 gpurun_out/final/kernel_stats.txt; cat gpurun_out/final/pmc/traffic.txt | head -20
