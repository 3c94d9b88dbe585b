#!/bin/bash
# development aid: duration of the fused linearisation with only some of its parts in the grid
# (1 plane, 2 bbox, 4 point-cuboid, 8 pose side, 16 point side; 7 = numeric classes, 24 = reprojection, 31 = all)
# (the PPO_BA_* switches exist only in the test-hooks build)
export PPO_BA_LIB="${PPO_BA_LIB:-$(cd "$(dirname "$0")/.." && pwd)/point-plane-object-slam_amd/csrc/libppo_ba_test.so}"
for m in ${PARTS:-1 2 4 7 8 16 24 31}; do echo -n "parts=$m: "; PPO_BA_LIN_PARTS=$m python bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('%.1f us' % (j['roofline']['launch_ms']*1e3))"; done
