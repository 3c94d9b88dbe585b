#!/usr/bin/env python3
"""development aid: device-side plan builder (csrc/ppo_plan.hpp) against the host builder on randomly structured windows, deterministic
mode, bit for bit (test-hooks library: PPO_BA_DEV_PLAN_MIN=0 lets the device builder take small windows, PPO_BA_HOST_PLAN=1 forces the host's).
  python tools/builder_sweep.py [n_cases] [first_seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "point-plane-object-slam_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from ppo_ba import binding as B, synth  # noqa: E402
import random_sweep as rs  # noqa: E402

n, first = (int(sys.argv[1]) if len(sys.argv) > 1 else 200), (int(sys.argv[2]) if len(sys.argv) > 2 else 9000)
hip = B.hip_backend(test_hooks=True)
os.environ["PPO_BA_DEV_PLAN_MIN"] = "0"
bad = 0
for seed in range(first, first + n):
    cfg, kw, elim = rs.random_case(seed)
    P = synth.make_problem(cfg, **kw)
    out = []
    for host in (False, True):
        if host:
            os.environ["PPO_BA_HOST_PLAN"] = "1"
        else:
            os.environ.pop("PPO_BA_HOST_PLAN", None)
        s = hip.create(P, B.default_options(deterministic=1, eliminate_cuboids=elim))
        st = s.run()
        out.append((st, s.read()))
        s.close()
    (sa, ra), (sb, rb) = out
    same = [(i["trials"], i["accepted"], i["chi2_after"]) for i in sa["iters"]] == [(i["trials"], i["accepted"], i["chi2_after"]) for i in sb["iters"]]
    for name in ("pose_qt", "point_xyz", "plane_coef", "cuboid_qt", "cuboid_scale", "mono_chi2", "stereo_chi2", "pl_chi2", "mono_level", "stereo_level", "pl_level"):
        same = same and np.array_equal(getattr(ra, name), getattr(rb, name))
    if not same:
        bad += 1
        print("seed", seed, cfg, kw, "elim", elim, "DIFFERS")
print("%d cases, %d differ" % (n, bad))
