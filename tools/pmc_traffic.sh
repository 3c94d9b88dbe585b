#!/bin/bash
# HBM traffic of the streaming kernels from the PMC counters, collected the way MI355X_MICROARCH.md prescribes:
# FETCH_SIZE and WRITE_SIZE in SEPARATE passes (3 + 2 TCC slots), --kernel-trace only, no other trace domain.
# usage (on the GPU box, from the repo root):  tools/pmc_traffic.sh gpurun_out/pmc
OUT="${1:-gpurun_out/pmc}"; mkdir -p "$OUT"; OUT="$(cd "$OUT" && pwd)"
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras --lin-reps 20"
RE="k_linearize_all|k_lin_poses|k_lin_points|k_schur_z|k_schur_pairs|k_reduce_stage|k_zero_ranges|k_backsub|k_chi2_all|k_chi2_points|chol_step|chol_back_all"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex "$RE" --output-format csv -d "$OUT/fetch" -o p -- $CMD > "$OUT/fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --kernel-include-regex "$RE" --output-format csv -d "$OUT/write" -o p -- $CMD > "$OUT/write.log" 2>&1
python "$ROOT/tools/pmc_traffic_summary.py" "$OUT" "$OUT/traffic.json" "$OUT/traffic.txt"
