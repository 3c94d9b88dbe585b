#!/bin/bash
# A/B of library builds placed in build/*.so (development aid): fused linearisation and its parts; extra environment in $1
for v in ${VARIANTS:-build/*.so}; do
  for m in ${PARTS:-8 16 24 31}; do
    echo -n "$v $1 parts=$m: "
    env $1 PPO_BA_LIB=$PWD/$v PPO_BA_LIN_PARTS=$m python bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('%.1f us' % (j['roofline']['launch_ms']*1e3))"
  done
done
