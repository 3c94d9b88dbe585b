#!/bin/bash
# development aid: PMC passes for selected kernels of one bench step.  usage: tools/pmc_probe.sh "<kernel regex>" <outdir>
RE="$1"; OUT="$2"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python /root/repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline --lin-reps 2"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-include-regex "$RE" --output-format csv -d "$OUT/p1" -o p -- $CMD > "$OUT/p1.log" 2>&1
rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE --kernel-include-regex "$RE" --output-format csv -d "$OUT/p2" -o p -- $CMD > "$OUT/p2.log" 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum TCC_EA0_RDREQ_sum --kernel-include-regex "$RE" --output-format csv -d "$OUT/p3" -o p -- $CMD > "$OUT/p3.log" 2>&1
find "$OUT" -name "*counter_collection.csv" | head
