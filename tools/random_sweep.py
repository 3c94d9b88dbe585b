#!/usr/bin/env python3
"""development aid: randomly structured small windows (sizes, edge classes, fixed observers, stereo share, elimination mode),
HIP library against the oracle through the full 5+10 protocol.  Prints every disagreement beyond the parity bar.
  python tools/random_sweep.py [n_cases] [first_seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "point-plane-object-slam_amd"))
import oracle_py  # noqa: E402
from ppo_ba import binding as B  # noqa: E402
from ppo_ba import synth  # noqa: E402


def random_case(seed):
    r = np.random.default_rng(seed)
    cfg = dict(n_kf=int(r.integers(3, 24)), n_points=int(r.integers(30, 700)), n_planes=int(r.integers(0, 9)), n_cuboids=int(r.integers(0, 5)))
    kw = dict(seed=seed, outlier_frac=float(r.choice([0.0, 0.05, 0.2])), stereo_frac=float(r.choice([0.0, 0.0, 0.3, 1.0])),
              with_corners=bool(r.integers(0, 2)) and cfg["n_cuboids"] > 0, with_vp_planes=int(r.integers(0, 3)) if cfg["n_planes"] > 1 else 0,
              fixed_observer_frac=float(r.choice([0.0, 0.0, 0.3])))
    return cfg, kw, int(r.integers(0, 2))


def rot_angle(qa, qb):
    qa, qb = np.asarray(qa), np.asarray(qb)
    if np.dot(qa, qb) < 0:
        qb = -qb
    return 2.0 * np.linalg.norm(qa - qb)


def run_case(hip, orc, seed):
    cfg, kw, elim = random_case(seed)
    P = synth.make_problem(cfg, **kw)
    o = B.default_options(eliminate_cuboids=elim)
    so, sh = orc.create(P, o), hip.create(P, o)
    to, th = so.run(), sh.run()
    ro, rh = so.read(), sh.read()
    so.close(); sh.close()
    sched = [(a["trials"], a["accepted"], a["result"]) for a in to["iters"]] == [(a["trials"], a["accepted"], a["result"]) for a in th["iters"]]
    n = ro.pose_qt.shape[0]
    m = {"t": float(np.sqrt(np.mean(np.sum((ro.pose_qt[:, 4:] - rh.pose_qt[:, 4:]) ** 2, axis=1)))),
         "r": max(rot_angle(ro.pose_qt[i, :4], rh.pose_qt[i, :4]) for i in range(n)),
         "p": float(np.sqrt(np.mean(np.sum((ro.point_xyz - rh.point_xyz) ** 2, axis=1)))) if ro.point_xyz.size else 0.0,
         "pl": float(np.abs(ro.plane_coef - rh.plane_coef).max()) if ro.plane_coef.size else 0.0,
         "cb": float(max(np.abs(ro.cuboid_qt - rh.cuboid_qt).max(), np.abs(ro.cuboid_scale - rh.cuboid_scale).max())) if ro.cuboid_qt.size else 0.0}
    chi = abs(to["iters"][-1]["chi2_after"] - th["iters"][-1]["chi2_after"]) / max(1.0, abs(to["iters"][-1]["chi2_after"])) if to["iters"] and th["iters"] else 0.0
    if ro.point_xyz.size:
        d = np.linalg.norm(ro.point_xyz - rh.point_xyz, axis=1)
        m["p99"] = float(np.quantile(d, 0.99))  # single landmarks left with one active observation after the culling are ill-conditioned
    else:
        m["p99"] = 0.0
    ok = sched and m["t"] < 1e-5 and m["r"] < 1e-6 and m["p"] < 1e-5 and m["pl"] < 1e-6 and m["cb"] < 1e-5
    return ok, sched, m, chi, cfg, kw, elim, len(to["iters"])


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    hip, orc = B.hip_backend(), oracle_py.backend()
    bad = 0
    worst = {"t": 0, "r": 0, "p": 0, "pl": 0, "cb": 0, "p99": 0}
    n_pose_bad = 0
    for seed in range(s0, s0 + n):
        try:
            ok, sched, m, chi, cfg, kw, elim, iters = run_case(hip, orc, seed)
        except Exception as e:  # noqa: BLE001
            print("seed %d EXCEPTION %r" % (seed, e))
            bad += 1
            continue
        if ok:
            for k in worst:
                worst[k] = max(worst[k], m[k])
        else:
            bad += 1
            n_pose_bad += int(not (sched and m["t"] < 1e-5 and m["r"] < 1e-6 and m["pl"] < 1e-6 and m["cb"] < 1e-5))
            print("seed %d %s elim=%d %s | schedule equal %s iters %d | %s | dchi2 %.1e" % (seed, cfg, elim, {k: v for k, v in kw.items() if k != "seed"}, sched, iters,
                                                                                 " ".join("%s %.1e" % kv for kv in m.items()), chi))
    print("%d cases, %d outside the bar (%d of them beyond the point RMSE alone); worst inside: %s" % (n, bad, n_pose_bad, " ".join("%s %.1e" % kv for kv in worst.items())))


if __name__ == "__main__":
    main()
