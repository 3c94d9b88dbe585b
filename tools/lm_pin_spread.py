#!/usr/bin/env python3
"""development aid: the reference-run fixture (tests/golden/orbslam_prebuilt_lm.json) per case -- deviation of the HIP result from the reference's
final state in deterministic mode, the spread of the default mode over N runs around it, and the oracle's own rounding spread (tests/sensitivity.py).
python tools/lm_pin_spread.py [runs]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "point-plane-object-slam_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from ppo_ba import binding as B
import oracle_py, sensitivity as S
import test_orbslam_lm_pin as T

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 20
hip, oracle = B.hip_backend(), oracle_py.backend()


class Ref:  # the reference's final state with the attribute names of a Result
    def __init__(self, fs, like):
        self.pose_qt = np.array(fs["pose_qt"]); self.point_xyz = np.array(fs["point_xyz"]).reshape(like.point_xyz.shape)
        self.plane_coef = np.array(fs["plane_coef"]).reshape(like.plane_coef.shape)
        self.cuboid_qt = np.array(fs["cuboid_qt"]).reshape(like.cuboid_qt.shape); self.cuboid_scale = np.array(fs["cuboid_scale"]).reshape(like.cuboid_scale.shape)


def fmt(m):
    return " ".join("%s %.1e" % (k, m[k]) for k in ("t", "r", "p", "pl", "cb"))


for k, case in enumerate(T.CASES):
    P = T.problem_of(case)
    sd = hip.create(P, B.default_options(deterministic=1)); sd.run(); rd = sd.read(); sd.close()
    ref = Ref(case["result"]["final_state"], rd)
    pts = S.determined_points(P, rd)
    oo, _, _ = S.spread(S.oracle_runs(oracle, P, B.default_options(), variants=S.VARIANTS), pts)
    worst = {q: 0.0 for q in S.BARS}
    worst_ref = {q: 0.0 for q in S.BARS}
    for _ in range(runs):
        sh = hip.create(P, B.default_options()); sh.run(); rh = sh.read(); sh.close()
        m, mr = S.metrics(rd, rh, pts), S.metrics(ref, rh, pts)
        for q in worst:
            worst[q] = max(worst[q], m[q]); worst_ref[q] = max(worst_ref[q], mr[q])
    print("case %2d  det vs reference: %s" % (k, fmt(S.metrics(ref, rd, pts))))
    print("         oracle spread   : %s" % fmt(oo))
    print("         default vs det  : %s   (worst of %d runs)" % (fmt(worst), runs))
    print("         default vs ref  : %s" % fmt(worst_ref))
