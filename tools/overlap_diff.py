#!/usr/bin/env python3
"""development aid (test-hooks library): per-iteration chi2 of the overlapped reduction against the plain one -- and of the plain one
against itself, the run-to-run spread of floating-point atomics -- on one BASELINE configuration.
  python tools/overlap_diff.py [cfg4]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "point-plane-object-slam_amd"))
from ppo_ba import binding as B  # noqa: E402
from ppo_ba import synth  # noqa: E402


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
    P = synth.make_problem(cfg, seed=1)
    hip = B.hip_backend(test_hooks=True)
    runs = {}
    for name, ov in (("plain", "0"), ("plain again", "0"), ("default", None), ("one block per launch", "1,0.0,0.0,0,2"), ("late, large slices", "3,0.4,0.4,0,2")):
        if ov is None:
            os.environ.pop("PPO_BA_OVERLAP", None)
        else:
            os.environ["PPO_BA_OVERLAP"] = ov
        s = hip.create(P, B.default_options())
        st = s.run()
        runs[name] = (st, s.read())
        s.close()
    ref_st, ref = runs["plain"]
    for name, (st, r) in runs.items():
        rel = [abs(a["chi2_after"] - b["chi2_after"]) / a["chi2_after"] for a, b in zip(ref_st["iters"], st["iters"])]
        same = [(a["trials"], a["accepted"]) for a in ref_st["iters"]] == [(a["trials"], a["accepted"]) for a in st["iters"]]
        print("%-22s same history %s | max pose diff %.1e point %.1e | chi2 rel diff per iteration: %s" % (
            name, same, np.abs(ref.pose_qt - r.pose_qt).max(), np.abs(ref.point_xyz - r.point_xyz).max(), " ".join("%.0e" % v for v in rel)))


if __name__ == "__main__":
    main()
