#!/bin/bash
# workgroups per launch of a solve attempt on the ICL-sized windows (what a grid barrier of a persistent kernel would have to synchronise): from a kernel trace
ROOT=$PWD; cd /tmp && export TMPDIR=/tmp
for kf in "15 800" "30 2000"; do
  rm -rf /tmp/swg; timeout 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/swg -o p -- python $ROOT/tools/small_window_trace.py $kf 10 > /tmp/swg.log 2>&1
  f=$(find /tmp/swg -name "*kernel_trace.csv" | head -1)
  echo "== $kf (keyframes points): kernel, workgroup size, workgroups per launch (launch count)"
  python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
c = collections.Counter()
for r in rows:
    wg = int(r["Workgroup_Size_X"]) if "Workgroup_Size_X" in r else int(r["Workgroup_Size"])
    gs = int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r["Grid_Size"])
    c[(r["Kernel_Name"][:60], wg, gs // wg)] += 1
for (k, wg, n), cnt in sorted(c.items(), key=lambda kv: -kv[1])[:14]:
    print("   %-60s %4d x %4d   (%d)" % (k, wg, n, cnt))
PY
done 2>&1 | tee $ROOT/gpurun_out/r06_small_grids.txt
