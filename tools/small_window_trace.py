#!/usr/bin/env python3
"""development aid: repeated 5+10 runs of one ICL-sized window (for a rocprofv3 --kernel-trace): python tools/small_window_trace.py [n_kf n_points reps]
PPO_LM=1: options.lm_control = host (the LM decisions after a round trip per trial) instead of the default (a kernel behind every trial);
PPO_DET=0: the opt-in floating-point-atomics mode instead of the default deterministic one"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "point-plane-object-slam_amd"))
from ppo_ba import binding as B, synth
n_kf = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n_pt = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
P = synth.make_problem(dict(n_kf=n_kf, n_points=n_pt, n_planes=max(2, n_kf // 3), n_cuboids=3 if n_kf >= 30 else 2), seed=1)
s = B.hip_backend(test_hooks=bool(os.environ.get("PPO_HOOKS"))).create(P, B.default_options(lm_control=int(os.environ.get("PPO_LM", "0")), deterministic=int(os.environ.get("PPO_DET", "1"))))
for _ in range(3):
    s.reset(); st = s.run()
t0 = time.perf_counter()
for _ in range(reps):
    s.reset(); st = s.run()
dt = time.perf_counter() - t0
print("%d KF / %d points (lm_control=%s): run %.3f ms per 5+10 call, %d LM iterations, %.1f us per iteration" % (n_kf, n_pt, os.environ.get("PPO_LM", "0"), 1e3 * dt / reps, st["n_iters"], 1e6 * dt / reps / st["n_iters"]))
s.close()
