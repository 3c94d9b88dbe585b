#!/usr/bin/env python3
"""development aid (test-hooks library): does the backward substitution on explicit inverses cost accuracy on ill-conditioned
windows?  Random windows of tests/test_parity_gpu.py::test_randomly_structured_windows_against_the_oracle, device against oracle, several
runs each with the inverses and with the triangular sweeps (PPO_BA_CHOL_NO_INV=1 is read once per process: the two arms are two processes).
  python tools/inv_accuracy.py [first_seed] [n] [repeats]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ("oracle", "point-plane-object-slam_amd", "tests", "tools"):
    sys.path.insert(0, os.path.join(ROOT, d))
import oracle_py  # noqa: E402
import random_sweep as rs  # noqa: E402
import sensitivity as S  # noqa: E402
from ppo_ba import binding as B  # noqa: E402
from ppo_ba import synth  # noqa: E402

s0 = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
hip, orc = B.hip_backend(test_hooks=True), oracle_py.backend()
n_out = 0
worst = {k: 0.0 for k in S.BARS}
for seed in range(s0, s0 + n):
    cfg, kw, elim = rs.random_case(seed)
    P = synth.make_problem(cfg, **kw)
    opt = B.default_options(eliminate_cuboids=elim)
    so = orc.create(P, opt)
    so.run()
    ro = so.read()
    outs = []
    for _ in range(reps):
        sh = hip.create(P, opt)
        sh.run()
        m = S.metrics(ro, sh.read())
        sh.close()
        outs.append([k for k, bar in S.BARS.items() if m[k] >= bar])
        for k in worst:
            worst[k] = max(worst[k], m[k] / S.BARS[k])
    so.close()
    if any(outs):
        n_out += 1
        print("seed %d outside the bar in %d of %d runs: %s" % (seed, sum(bool(o) for o in outs), reps, sorted(set(k for o in outs for k in o))))
print("NO_INV=%s: %d of %d windows outside the bar in at least one of %d runs; worst deviation / bar: %s" % (
    os.environ.get("PPO_BA_CHOL_NO_INV", "0"), n_out, n, reps, " ".join("%s %.2f" % kv for kv in worst.items())))
