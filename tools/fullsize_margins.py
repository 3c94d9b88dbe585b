#!/usr/bin/env python3
"""development aid: HIP-vs-oracle deviations and the oracle's own spread (tests/sensitivity.py) at full config-4 size for the
edge-class options of SURVEY 8(d).  python tools/fullsize_margins.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("oracle", "point-plane-object-slam_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import oracle_py  # noqa: E402
import sensitivity as S  # noqa: E402
from ppo_ba import binding as B  # noqa: E402
from ppo_ba import synth  # noqa: E402

hip, orc = B.hip_backend(), oracle_py.backend()
for name, kw in (("base", {}), ("stereo20", dict(stereo_frac=0.2)), ("corners", dict(with_corners=True)), ("vp1000", dict(with_vp_planes=1000)),
                 ("fixedobs10", dict(fixed_observer_frac=0.1)), ("all", dict(stereo_frac=0.2, with_corners=True, with_vp_planes=1000, fixed_observer_frac=0.1))):
    P = synth.make_problem("cfg4", seed=1, **kw)
    o = B.default_options()
    sh = hip.create(P, o)
    th, rh = sh.run(), sh.read()
    sh.close()
    runs = S.oracle_runs(orc, P, o, variants=("reversed", "fma"))
    to, ro = runs[0]
    m = S.metrics(ro, rh)
    oo, s_cb, s_pt = S.spread(runs)
    d_pt, d_cb = S.per_point(ro, rh), S.per_cuboid(ro, rh)
    well = s_pt < 1e-6
    print(name, "sched eq", S.schedule(to) == S.schedule(th), "| HIP-oracle", " ".join("%s %.1e" % kv for kv in m.items()), "| oracle-oracle", " ".join("%s %.1e" % kv for kv in oo.items()))
    print("   points: well %.4f, rmse(well) %.1e, max d %.1e, max d/(10 s) over ill %.2f; cuboids: d>1e-5 %d, s>1e-6 %d, max d/max(1e-5,10s) %.2f; dchi2 %.1e" % (
        well.mean(), np.sqrt(np.mean(d_pt[well] ** 2)), d_pt.max(), (d_pt[~well] / np.maximum(1e-5, 10 * s_pt[~well])).max() if (~well).any() else 0,
        int((d_cb > 1e-5).sum()), int((s_cb > 1e-6).sum()), (d_cb / np.maximum(1e-5, 10 * s_cb)).max(),
        abs(to["iters"][-1]["chi2_after"] - th["iters"][-1]["chi2_after"]) / to["iters"][-1]["chi2_after"]))
    sys.stdout.flush()
