#!/usr/bin/env python3
"""Development aid: the overlapped path repeated on fresh handles of configs 3 and 4 -- recycled device blocks, the hand-off
granules of the backward substitution, the atomic trailing tiles -- looking for the rare large outlier a race would leave.
  python tools/stress_overlap.py [repeats]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "point-plane-object-slam_amd"))
from ppo_ba import binding as B  # noqa: E402
from ppo_ba import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
for cfg, kw in (("cfg4", dict(seed=1)), ("cfg3", dict(seed=2)), ("cfg4", dict(seed=3, stereo_frac=0.2, with_corners=True))):
    P = synth.make_problem(cfg, **kw)
    ref, worst, hist = None, [], set()
    for it in range(n):
        s = B.hip_backend().create(P, B.default_options())
        st = s.run()
        r = s.read()
        s.close()
        assert np.all(np.isfinite(r.pose_qt)) and np.all(np.isfinite(r.point_xyz))
        key = (st["iters"][-1]["chi2_after"], r.pose_qt.copy())
        hist.add(tuple((i["trials"], i["accepted"]) for i in st["iters"]))
        if ref is None:
            ref = key
            continue
        worst.append((float(np.abs(key[1] - ref[1]).max()), abs(key[0] - ref[0]) / abs(ref[0])))
    worst.sort(reverse=True)
    print(cfg, kw, "repeats %d, distinct LM histories %d, worst pose / chi2 differences:" % (n, len(hist)), ["%.1e/%.1e" % w for w in worst[:4]],
          "median %.1e" % np.median([w[0] for w in worst]))
