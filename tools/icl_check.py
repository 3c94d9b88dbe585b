#!/usr/bin/env python3
"""development aid: the two 15-keyframe windows the tools use (bench.py: 4 planes; small_window_trace.py: 5 planes), repeated runs on one handle each,
in this order and then reversed -- does the figure depend on the window or on what ran before it in the process?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "point-plane-object-slam_amd"))
from ppo_ba import binding as B, synth
cfgs = [("bench 15 KF (4 planes)", dict(n_kf=15, n_points=800, n_planes=4, n_cuboids=2)), ("trace 15 KF (5 planes)", dict(n_kf=15, n_points=800, n_planes=5, n_cuboids=2)),
        ("bench 30 KF", dict(n_kf=30, n_points=2000, n_planes=10, n_cuboids=3))]
for order in (cfgs, cfgs[::-1]):
    for label, cfg in order:
        P = synth.make_problem(cfg, seed=1)
        for lm in (0, 1):
            s = B.hip_backend().create(P, B.default_options(lm_control=lm))
            for _ in range(3):
                s.reset(); st = s.run()
            t0 = time.perf_counter()
            for _ in range(40):
                s.reset(); st = s.run()
            dt = time.perf_counter() - t0
            tr = [i["trials"] for i in st["iters"]]
            print("%-24s lm_control=%d: %.3f ms per call, %d iterations, %d solve attempts, %.1f us per attempt" % (label, lm, 1e3 * dt / 40, st["n_iters"], sum(tr), 1e6 * dt / 40 / sum(tr)))
            s.close()
