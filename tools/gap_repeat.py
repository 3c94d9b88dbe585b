#!/usr/bin/env python3
"""development aid: ONE random window (tools/random_sweep.py seed) solved many times on the device; the optimality gap of every run against the
oracle's minimum (tests/sensitivity.py).  A rare large value is a race, a steady spread is summation order.
  python tools/gap_repeat.py seed [repeats]      (PPO_TREE=/path/to/another/checkout runs that tree's library, oracle and helpers)"""
import os
import sys

import numpy as np

ROOT = os.environ.get("PPO_TREE") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ("oracle", "point-plane-object-slam_amd", "tests", "tools"):
    sys.path.insert(0, os.path.join(ROOT, d))
import oracle_py  # noqa: E402
import random_sweep as rs  # noqa: E402
import sensitivity as S  # noqa: E402
from ppo_ba import binding as B  # noqa: E402
from ppo_ba import synth  # noqa: E402

seed = int(sys.argv[1])
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
hip, orc = B.hip_backend(), oracle_py.backend()
cfg, kw, elim = rs.random_case(seed)
P = synth.make_problem(cfg, **kw)
opt = B.default_options(eliminate_cuboids=elim)
so = orc.create(P, opt)
to = so.run()
ro = so.read()
gaps, scheds = [], set()
for _ in range(reps):
    sh = hip.create(P, opt)
    th = sh.run()
    rh = sh.read()
    sh.close()
    gaps.append(S.optimality_gap(orc, so, ro, rh)[0])
    scheds.add(tuple(S.schedule(th)))
gaps = np.array(gaps)
print("tree %s seed %d %s elim %d: %d runs, gap median %.1e, 90%% %.1e, max %.1e, runs above 1e-6: %d, distinct LM histories %d (oracle's among them: %s)" % (
    ROOT, seed, cfg, elim, reps, np.median(gaps), np.quantile(gaps, 0.9), gaps.max(), int((gaps > 1e-6).sum()), len(scheds), tuple(S.schedule(to)) in scheds))
