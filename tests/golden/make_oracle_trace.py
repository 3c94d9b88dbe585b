#!/usr/bin/env python3
"""Regenerates tests/golden/oracle_trace_small.json: the LM schedule and final estimates of the oracle on one small
seeded graph (a regression fixture of the oracle itself -- NOT a reference output)."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(REPO, "point-plane-object-slam_amd"))
sys.path.insert(0, os.path.join(REPO, "oracle"))
from ppo_ba import binding as B, synth  # noqa: E402
import oracle_py  # noqa: E402

cfg, seed = "small", 21
P = synth.make_problem(cfg, seed=seed)
s = oracle_py.backend().create(P, B.default_options())
st = s.run()
r = s.read()
out = dict(config=cfg, seed=seed, iters=st["iters"], pose_qt=r.pose_qt.tolist(), cuboid_scale=r.cuboid_scale.tolist(),
           n_mono_level1=int(r.mono_level.sum()))
json.dump(out, open(os.path.join(HERE, "oracle_trace_small.json"), "w"), indent=1)
print("wrote oracle_trace_small.json:", len(st["iters"]), "iterations")
