#!/bin/bash
# round 6: create / run / read of ICL-sized windows measured from C++ (examples/replay: plain g++ caller of the C ABI, a fresh handle per call -- no Python
# marshalling in the numbers), with the library's own create phases (PPO_BA_VERBOSE) for the last call.
mkdir -p gpurun_out
python - <<'PY'
import sys, os
sys.path.insert(0, "point-plane-object-slam_amd")
from ppo_ba import synth
for n_kf, n_pt, n_pl, n_cb in ((15, 800, 4, 2), (30, 2000, 10, 3), (50, 5000, 16, 3)):
    synth.make_problem(dict(n_kf=n_kf, n_points=n_pt, n_planes=n_pl, n_cuboids=n_cb), seed=1).save("/tmp/w%d.ppoba" % n_kf)
PY
for k in 15 30 50; do
  echo "== $k keyframes"
  ./examples/replay /tmp/w$k.ppoba 30 | grep "^call" | tail -12 | awk '{c+=$4; r+=$7; d+=$13; n++} END {printf "   mean of the last %d calls: create %.3f ms, run %.3f ms, read %.3f ms\n", n, c/n, r/n, d/n}'
  ./examples/replay /tmp/w$k.ppoba 30 | grep "^call" | tail -3
done
PPO_BA_VERBOSE=1 ./examples/replay /tmp/w15.ppoba 3 2>&1 | grep "create:" | tail -12
