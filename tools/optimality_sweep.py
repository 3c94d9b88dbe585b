#!/usr/bin/env python3
"""development aid: for randomly structured windows and the full-size configurations, how good a minimiser of the ORACLE's cost is the
HIP final state (tests/sensitivity.py: optimality_gap), next to the deviations of the parity bar, the oracle's own spread under
rounding changes, and the HIP library's own spread (default against deterministic = 1).
  python tools/optimality_sweep.py random [n] [first_seed]      python tools/optimality_sweep.py full"""
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


def one(hip, orc, P, opt_kw, tag):
    o = B.default_options(**opt_kw)
    jac = int(os.environ.get("JAC", "0"))  # JAC=1: the device runs with analytic Jacobians (the oracle is always the numeric reference)
    so, sh = orc.create(P, o), hip.create(P, B.default_options(jacobian_mode=jac, **opt_kw))
    to, th = so.run(), sh.run()
    ro, rh = so.read(), sh.read()
    sd = hip.create(P, B.default_options(deterministic=1, jacobian_mode=jac, **opt_kw))
    td = sd.run()
    rd = sd.read()
    sd.close()
    gap, c0, c1 = S.optimality_gap(orc, so, ro, rh)
    gapd, _, _ = S.optimality_gap(orc, so, ro, rd)
    m = S.metrics(ro, rh)
    mh = S.metrics(rh, rd)
    sched = S.schedule(to) == S.schedule(th)
    sched_d = S.schedule(td) == S.schedule(th)
    out_of_bar = [k for k, bar in S.BARS.items() if m[k] >= bar]
    runs = S.oracle_runs(orc, P, o, variants=("reversed", "fma", "fma+reversed", "compensated"))
    ogap = max(S.optimality_gap(orc, so, ro, r)[0] for _, r in runs[1:])
    line = "%s sched %s (hip det=default %s) gap %.1e (det %.1e; oracle variants %.1e) chi2 %.6g | " % (tag, sched, sched_d, gap, gapd, ogap, c0) + " ".join("%s %.1e" % kv for kv in m.items())
    if out_of_bar or not sched:
        oo, s_cb, s_pt = S.spread(runs)
        flips = any(S.schedule(st) != S.schedule(runs[0][0]) for st, _ in runs[1:])
        line += " | OUT %s oracle-spread %s flips %s | hip-spread %s" % (out_of_bar, " ".join("%s %.1e" % (k, oo[k]) for k in out_of_bar), flips,
                                                                  " ".join("%s %.1e" % (k, mh[k]) for k in out_of_bar))
        d_cb = S.per_cuboid(ro, rh)
        if d_cb.size and (d_cb >= 1e-5).any():
            idx = np.nonzero(d_cb >= 1e-5)[0]
            line += " | cuboids " + " ".join("#%d d %.1e s %.1e h %.1e" % (i, d_cb[i], s_cb[i], S.per_cuboid(rh, rd)[i]) for i in idx)
    print(line, flush=True)
    so.close(); sh.close()


def main():
    hip, orc = B.hip_backend(), oracle_py.backend()
    mode = sys.argv[1] if len(sys.argv) > 1 else "random"
    if mode == "random":
        n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
        s0 = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
        for seed in range(s0, s0 + n):
            cfg, kw, elim = rs.random_case(seed)
            one(hip, orc, synth.make_problem(cfg, **kw), dict(eliminate_cuboids=elim), "seed %d" % seed)
    else:
        for cfg, kw in [("cfg2", {}), ("cfg3", {}), ("cfg4", {}), ("cfg4", dict(stereo_frac=0.2)), ("cfg4", dict(with_corners=True)), ("cfg4", dict(with_vp_planes=1000)),
                        ("cfg4", dict(fixed_observer_frac=0.1)), ("cfg4", dict(stereo_frac=0.2, with_corners=True, with_vp_planes=1000, fixed_observer_frac=0.1)),
                        ("cfg4", dict(seed=2)), ("cfg4", dict(seed=3))]:
            kw = dict(kw)
            seed = kw.pop("seed", 1)
            one(hip, orc, synth.make_problem(cfg, seed=seed, **kw), {}, "%s seed %d %s" % (cfg, seed, kw))


if __name__ == "__main__":
    main()
