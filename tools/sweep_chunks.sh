#!/bin/bash
# development aid: sweep the chunk sizes of the pose-major linearisation and of the Schur pair kernel
# (the PPO_BA_* switches exist only in the test-hooks build)
export PPO_BA_LIB="${PPO_BA_LIB:-$(cd "$(dirname "$0")/.." && pwd)/point-plane-object-slam_amd/csrc/libppo_ba_test.so}"
P='import json,sys; j=json.loads(sys.stdin.read()); k=j["config"]["kernels"]; print(sys.argv[1], round(j["value"],1), {n: round(v["us"],1) for n,v in k.items()})'
for c in ${POSE_CHUNKS:-}; do PPO_BA_POSE_CHUNK=$c python bench.py --no-cpu-baseline --steps 10 2>&1 | tail -1 | python -c "$P" "pose_chunk=$c"; done
for c in ${SCHUR_CHUNKS:-256 128 64}; do PPO_BA_SCHUR_CHUNK=$c python bench.py --no-cpu-baseline --steps 10 2>&1 | tail -1 | python -c "$P" "schur_chunk=$c"; done
