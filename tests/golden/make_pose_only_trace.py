#!/usr/bin/env python3
"""Regenerates tests/golden/pose_only_small.json: inputs and the oracle's outputs of the pose-only optimisation
(Optimizer::PoseOptimization restated in oracle/ppo_oracle.cpp) for two small frames.  A regression fixture of the
oracle itself; the leaf functions underneath are pinned separately by g2o_prebuilt.json (onlypose_mono / onlypose_stereo)."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(REPO, "oracle"))
sys.path.insert(0, os.path.join(REPO, "point-plane-object-slam_amd"))
import oracle_py  # noqa: E402
from ppo_ba import pose  # noqa: E402

fr = pose.make_frames(n_frames=2, n_obs=[60, 25], stereo_fraction=0.4, outlier_fraction=0.15, seed=11)
out = pose.optimize(oracle_py.load(), fr, prefix="oracle_")
doc = {
    "inputs": {"pose_qt": fr.pose_qt.tolist(), "cam": fr.cam.tolist(), "edge_offsets": fr.edge_offsets.tolist(), "xw": fr.xw.tolist(),
               "obs": fr.obs.tolist(), "inv_sigma2": fr.inv_sigma2.tolist()},
    "expected": {"pose_qt": out["pose_qt"].tolist(), "outlier": out["outlier"].tolist(), "n_inliers": out["n_inliers"].tolist(),
                 "chi2": out["chi2"].tolist(), "rounds_run": [s["rounds_run"] for s in out["stats"]]},
}
json.dump(doc, open(os.path.join(HERE, "pose_only_small.json"), "w"), indent=0)
print("wrote pose_only_small.json", out["n_inliers"])
