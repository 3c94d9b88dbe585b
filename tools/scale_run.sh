#!/bin/bash
# For the day an 8-GPU node is available (no run with more than one real RCCL rank has happened yet: SCALE_r01..r04 are `skipped` records).
# One command, everything the multi-GPU row of SURVEY section 8(e) needs, written to profiles/rNN_scale.txt:
#   1. tests/test_multigpu.py  -- two real RCCL ranks against the single-rank answer (skipped on a one-GPU box)
#   2. bench.py --gpus 1/2/4/8 on the BASELINE graph (cfg4 = cfg5's graph, landmark-range sharded: strong scaling), all-reduce variant
#   3. the same with options.solve_on_owner = 1 (reduce to rank 0, solve there, broadcast x)
#   4. the weak-scaling configuration cfg5w (50 000 points per rank)
# Every bench line carries config.multi_gpu (ms of the exchange steps of one solve attempt, timed alone) next to the headline rate.
#   tools/scale_run.sh [round tag, default r06] [steps, default 20]
set -u
cd "$(dirname "$0")/.."
TAG=${1:-r06}
STEPS=${2:-20}
OUT=profiles/${TAG}_scale.txt
NGPU=$(python -c "import torch; print(torch.cuda.device_count())")
export HSA_ENABLE_IPC_MODE_LEGACY=0
{
  echo "# tools/scale_run.sh $TAG on $(hostname), $NGPU visible GPUs, $(date -u +%Y-%m-%dT%H:%MZ)"
  echo "## tests/test_multigpu.py"
  python -m pytest tests/test_multigpu.py -m gpu -q 2>&1 | tail -3
  port=29610
  for variant in "cfg4 0" "cfg4 1" "cfg5w 0"; do
    set -- $variant
    cfg=$1; owner=$2
    echo "## bench.py --config $cfg --solve-on-owner $owner"
    for n in 1 2 4 8; do
      [ "$n" -le "$NGPU" ] || { echo "n_gpus $n: skipped ($NGPU visible)"; continue; }
      port=$((port + 1))
      if [ "$n" -eq 1 ]; then
        line=$(python bench.py --gpus 1 --steps "$STEPS" --warmup 3 --config "$cfg" --no-cpu-baseline --no-extras 2>/dev/null | tail -1)
      else
        line=$(python -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" --master-addr 127.0.0.1 --master-port "$port" bench.py --gpus "$n" --steps "$STEPS" --warmup 3 \
               --config "$cfg" --solve-on-owner "$owner" --no-cpu-baseline --no-extras 2>/dev/null | tail -1)
      fi
      echo "$line" | python -c "
import sys, json
try:
    j = json.loads(sys.stdin.read())
    print('n_gpus %d: %.0f LM it/s, %.4f ms per LM iteration, scaling %s, exchange %s' % (j['n_gpus'], j['value'], j['config']['ms_per_lm_iter'], j['scaling'], json.dumps(j['config'].get('multi_gpu'))))
except Exception as e:
    print('n_gpus $n: no bench line (%s)' % e)
"
    done
  done
} 2>&1 | tee "$OUT"
echo "wrote $OUT"
