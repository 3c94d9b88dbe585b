#!/bin/bash
# development aid: SQ occupancy / issue / wait counters of the linearisation kernels (fused launch and its parts alone)
# usage (on the GPU box, from the repo root):  tools/pmc_sq.sh gpurun_out/sq
OUT="${1:-gpurun_out/sq}"; mkdir -p "$OUT"; OUT="$(cd "$OUT" && pwd)"
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --lin-reps 5"
RE="${PMC_RE:-k_linearize_all|k_lin_poses|k_lin_points|k_schur_pairs|k_reduce_stage3|k_update_all|k_chi2_all|k_backsub}"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-include-regex "$RE" --output-format csv -d "$OUT/a" -o p -- $CMD > "$OUT/a.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR --kernel-include-regex "$RE" --output-format csv -d "$OUT/b" -o p -- $CMD > "$OUT/b.log" 2>&1
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    print(k)
    for c, v in sorted(d.items()):
        print("   %-26s %14.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
