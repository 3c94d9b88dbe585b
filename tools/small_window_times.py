#!/usr/bin/env python3
"""development aid: what a per-call user pays on ICL-sized windows (the reference's real workload: O(10^3) points)
-- create / run / read / destroy medians with the host-built plans (what the product library uses below 40 000 reprojection edges) and
with the device-built plans forced on (test-hooks library, PPO_BA_DEV_PLAN_MIN=0)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "point-plane-object-slam_amd"))
from ppo_ba import binding as B  # noqa: E402
from ppo_ba import synth  # noqa: E402


def med(xs):
    return sorted(xs)[len(xs) // 2]


def measure(backend, P, reps=30):
    o = B.default_options()
    c, r, d, x = [], [], [], []
    for i in range(reps + 3):
        t0 = time.perf_counter(); s = backend.create(P, o)
        t1 = time.perf_counter(); st = s.run()
        t2 = time.perf_counter(); s.read()
        t3 = time.perf_counter(); s.close()
        t4 = time.perf_counter()
        if i >= 3:
            c.append(t1 - t0); r.append(t2 - t1); d.append(t3 - t2); x.append(t4 - t3)
    return 1e3 * med(c), 1e3 * med(r), 1e3 * med(d), 1e3 * med(x), st["n_iters"]


def main():
    prod, hooks = B.hip_backend(), B.hip_backend(test_hooks=True)
    for cfg in (dict(n_kf=15, n_points=800, n_planes=4, n_cuboids=2), dict(n_kf=30, n_points=2000, n_planes=10, n_cuboids=3),
                dict(n_kf=50, n_points=5000, n_planes=20, n_cuboids=5), "cfg2", "cfg4"):
        P = synth.make_problem(cfg, seed=1)
        sz = P.sizes()
        a = measure(prod, P)
        os.environ["PPO_BA_DEV_PLAN_MIN"] = "0"
        b = measure(hooks, P)
        os.environ["PPO_BA_DEV_PLAN_MIN"] = "1000000000"
        h = measure(hooks, P)
        del os.environ["PPO_BA_DEV_PLAN_MIN"]
        print("%s: %d KF %d points %d mono edges | product  create %.2f run %.2f read %.2f destroy %.2f = %.2f ms (%d LM iterations) | device plans create %.2f run %.2f | host plans create %.2f run %.2f" % (
            cfg if isinstance(cfg, str) else "%dkf/%dpt" % (cfg["n_kf"], cfg["n_points"]), sz["n_poses"], sz["n_points"], sz["n_mono"], a[0], a[1], a[2], a[3], sum(a[:4]), a[4], b[0], b[1], h[0], h[1]), flush=True)


if __name__ == "__main__":
    main()
