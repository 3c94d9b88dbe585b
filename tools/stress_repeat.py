#!/usr/bin/env python3
"""Development aid: run the same graph many times (fresh solver each time, so recycled blocks are exercised too) and print the
distribution of run-to-run differences.  Atomic summation order makes runs differ at the 1e-12 level; a data race shows
up as rare, much larger outliers."""
import sys
import numpy as np
sys.path.insert(0, "/root/repo/point-plane-object-slam_amd")
from ppo_ba import binding as B, synth

cfgs = [("small", dict(seed=1)), ("small", dict(seed=3, with_corners=True)), (dict(n_kf=12, n_points=400, n_planes=6, n_cuboids=0), dict(seed=7)),
        (dict(n_kf=40, n_points=2000, n_planes=10, n_cuboids=5), dict(seed=2))]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for cfg, kw in cfgs:
    P = synth.make_problem(cfg, **kw)
    ref = None
    worst = []
    for it in range(n):
        s = B.hip_backend().create(P, B.default_options())
        st = s.run()
        r = s.read()
        s.close()
        key = (st["iters"][-1]["chi2_after"], r.pose_qt.copy(), tuple(i["trials"] for i in st["iters"]))
        if ref is None:
            ref = key
            continue
        dchi = abs(key[0] - ref[0]) / abs(ref[0])
        dpose = np.abs(key[1] - ref[1]).max()
        worst.append((dpose, dchi, key[2] == ref[2], it))
    worst.sort(reverse=True)
    print(cfg, kw, "worst pose diffs:", ["%.1e/%.1e/%s" % (a, b, c) for a, b, c, _ in worst[:5]], "median %.1e" % np.median([w[0] for w in worst]))
