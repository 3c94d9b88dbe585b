#!/bin/bash
# development aid: duration of the fused linearisation for several grid orders of its parts (digits: 0 plane 1 bbox 2 corner 3 ptcub 4 pose 5 points)
export PPO_BA_LIB="${PPO_BA_LIB:-$(cd "$(dirname "$0")/.." && pwd)/point-plane-object-slam_amd/csrc/libppo_ba_test.so}"
for o in ${ORDERS:-504213 312045 054213 540213 502413 450213 504321}; do echo -n "order=$o: "; PPO_BA_LIN_ORDER=$o python bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('%.1f us' % (j['roofline']['launch_ms']*1e3))"; done
