#!/bin/bash
# development aid: instruction-cache counters of the fused linearisation (whole launch and the numeric part alone)
# (the PPO_BA_* switches exist only in the test-hooks build)
export PPO_BA_LIB="${PPO_BA_LIB:-$(cd "$(dirname "$0")/.." && pwd)/point-plane-object-slam_amd/csrc/libppo_ba_test.so}"
OUT="${1:-gpurun_out/ic}"; mkdir -p "$OUT"; OUT="$(cd "$OUT" && pwd)"
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp
for parts in 31 7 24; do
PPO_BA_LIN_PARTS=$parts rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU --kernel-include-regex "k_linearize_all" --output-format csv -d "$OUT/p$parts" -o p -- python $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras --lin-reps 5 > "$OUT/p$parts.log" 2>&1
echo "parts=$parts"
python - "$OUT/p$parts" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in sorted(acc.items()):
    v = v[len(v)//2:]  # the timed launches (the PARTS switch applies to all)
    print("   %-28s %14.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
done
