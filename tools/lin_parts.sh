#!/bin/bash
# development aid: time the parts of k_linearize_all alone (1 plane, 2 bbox, 4 point-cuboid, 8 pose side, 16 landmark side)
P='import json,sys; j=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(sys.argv[1], round(j["config"]["kernels"]["k_linearize_all"]["us"],1))'
for m in 1 2 4 7 8 16 24 31; do PPO_BA_LIN_PARTS=$m python bench.py --no-cpu-baseline --steps 2 --warmup 1 2>&1 | python -c "$P" "parts=$m"; done
