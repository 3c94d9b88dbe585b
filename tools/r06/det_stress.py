#!/usr/bin/env python3
"""round 6: the deterministic (default) mode on fresh handles over RECYCLED device blocks -- the factor's array S is never cleared (first touch does not read it,
csrc/ppo_chol.hip: StepFix), the fixed-point array is cleared by the update launch, the linearisation ranges are converted by the decide / hand-over launches:
every repeat of a window must be BIT-identical whatever lay in the recycled blocks before (other windows of other sizes are solved in between), with the LM decisions
on the device and on the host, through the overlapped reduction, the two-panel path and the one-launch solve.   python tools/r06/det_stress.py [repeats]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "point-plane-object-slam_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from ppo_ba import binding as B, synth
import random_sweep as rs
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
hip = B.hip_backend()
windows = [("cfg4", dict(seed=1), 1), ("cfg3", dict(seed=2), 1), ("cfg2", dict(seed=1), 1), ("cfg4", dict(seed=3, stereo_frac=0.2, with_corners=True), 0),
           (dict(n_kf=15, n_points=800, n_planes=4, n_cuboids=2), dict(seed=1), 1), (dict(n_kf=30, n_points=2000, n_planes=10, n_cuboids=3), dict(seed=1), 1),
           ("small", dict(seed=2, noise_scale=20.0), 1)]
for seed in range(7000, 7012):
    cfg, kw, elim = rs.random_case(seed)
    cfg["n_kf"] += 10 * (seed % 4)
    windows.append((cfg, kw, elim))
probs = [(synth.make_problem(c, **kw), elim) for c, kw, elim in windows]
ref = [None] * len(probs)
bad = 0
rng = np.random.default_rng(1)
for rep in range(n):
    for k in rng.permutation(len(probs)):
        P, elim = probs[k]
        lm = int(rng.integers(0, 2)) if rep > 0 else 0
        s = hip.create(P, B.default_options(eliminate_cuboids=elim, lm_control=lm))
        st = s.run(); r = s.read(); s.close()
        key = ([(i["trials"], i["accepted"], i["chi2_after"], i["result"]) for i in st["iters"]], r.pose_qt.copy(), r.point_xyz.copy(), r.plane_coef.copy(), r.cuboid_qt.copy(), r.mono_chi2.copy())
        if ref[k] is None:
            ref[k] = key
            continue
        same = key[0] == ref[k][0] and all(np.array_equal(a, b) for a, b in zip(key[1:], ref[k][1:]))
        # (lm_control = host differs from the device control in pow(x, 3): one ulp of lambda in about one accepted trial in a thousand; chi2 to 1e-12 then)
        if not same and lm == 1:
            same = [t[:2] for t in key[0]] == [t[:2] for t in ref[k][0]] and np.abs(key[1] - ref[k][1]).max() < 1e-9
        if not same:
            bad += 1
            print("window %d rep %d lm %d DIFFERS: max |pose diff| %.3e, chi2 %r vs %r" % (k, rep, lm, np.abs(key[1] - ref[k][1]).max(), key[0][-1], ref[k][0][-1]))
print("%d windows x %d repeats on recycled blocks: %d differ" % (len(probs), n, bad))
sys.exit(1 if bad else 0)
