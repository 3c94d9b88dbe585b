#!/bin/bash
# round 6 (review item 5): (a) what the waves of k_linearize_all and k_update_all wait for -- SQ / TCP counter passes on the round-6 build (each --pmc
# pass on its own, kernel trace only); (b) the timing-only bound of fusing the landmark-major half of the linearisation into the update launch
# (PPO_BA_FUSE_BOUND=1 of the test-hooks build: the landmark workgroups of k_update_all also run lin_points_body on the trial state) next to the
# fused linearisation without its landmark part (PPO_BA_LIN_PARTS=15) -> gpurun_out/r06_lin_stalls.txt
ROOT=$PWD; OUT=$ROOT/gpurun_out/r06_lin_stalls; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --lin-reps 5"
RE="k_linearize_all|k_update_all"
pass() { timeout 150 rocprofv3 --kernel-trace --pmc $2 --kernel-include-regex "$RE" --output-format csv -d $OUT/$1 -o p -- $CMD > $OUT/$1.log 2>&1 || echo "pass $1 failed (see $1.log)"; }  # (a pass the hardware cannot schedule aborts and then hangs: bounded)
pass a "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
pass b "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM"
pass c "TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"
pass d "TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum"
pass e "TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum TCC_EA0_RDREQ_sum"
RE="k_|chol_"  # (the LDS conflict pass over every kernel of the library: where else do lanes gather rows?)
pass f "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS"
{
echo "# round 6, tools/r06/lin_stalls.sh on MI355X: counters per launch (average over the launches of one bench run), BASELINE window, default (deterministic) mode"
python - $OUT <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    print(k)
    for c, v in sorted(d.items()):
        print("   %-40s %16.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
    g = lambda n: (sum(d[n]) / len(d[n])) if n in d else float("nan")
    wc = g("SQ_WAVE_CYCLES")
    if wc == wc and wc > 0:
        print("   -> of the wave cycles: waiting for any instruction %.0f %%, for LDS %.0f %%, VALU issuing %.0f %%, LDS issuing %.0f %%, VMEM issuing %.0f %%" % (
            100 * g("SQ_WAIT_INST_ANY") / wc, 100 * g("SQ_WAIT_INST_LDS") / wc, 100 * g("SQ_ACTIVE_INST_VALU") / wc, 100 * g("SQ_ACTIVE_INST_LDS") / wc, 100 * g("SQ_ACTIVE_INST_VMEM") / wc))
        print("   -> waves %.0f, cycles per wave %.0f, VALU instructions per wave %.0f" % (g("SQ_WAVES"), wc / max(g("SQ_WAVES"), 1), g("SQ_INSTS_VALU") / max(g("SQ_WAVES"), 1)))
PY
echo
echo "# fusion bound (timing only; the results of the run stay right: the extra work lands in the linearisation buffer nobody reads)"
export PPO_BA_LIB=$ROOT/point-plane-object-slam_amd/csrc/libppo_ba_test.so
for fb in 0 1; do
  rm -rf /tmp/fb$fb
  if [ $fb = 1 ]; then export PPO_BA_FUSE_BOUND=1; else unset PPO_BA_FUSE_BOUND; fi
  rocprofv3 --kernel-trace --stats -d /tmp/fb$fb -o p -- python $ROOT/bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 > $OUT/fb$fb.log 2>&1
  echo "PPO_BA_FUSE_BOUND=$fb: $(grep -o '"value": [0-9.]*' $OUT/fb$fb.log | head -1) LM iterations/s"
  python $ROOT/tools/rocpd_summary.py $(find /tmp/fb$fb -name "*.db" | head -1) | grep -E "k_update_all|k_linearize_all" | awk '{printf "   %-50s calls %s avg %s us\n", substr($1,1,50), $2, $4}'
done
unset PPO_BA_FUSE_BOUND
for m in 31 15 16; do echo -n "fused linearisation alone, parts mask $m (16 = landmark side, 15 = everything else): "; PPO_BA_LIN_PARTS=$m python $ROOT/bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('%.1f us' % (j['roofline']['launch_ms']*1e3))"; done
} 2>&1 | tee $ROOT/gpurun_out/r06_lin_stalls.txt
