#!/usr/bin/env python3
"""development aid: where ppo_ba_create spends its time on an ICL-sized window (options.verbose phases on stderr of the LAST of several
creates, block cache warm), and the first run of a fresh handle against a repeated run on the same handle.
  python tools/create_breakdown.py [n_kf n_points]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "point-plane-object-slam_amd"))
from ppo_ba import binding as B, synth
n_kf = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n_pt = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
P = synth.make_problem(dict(n_kf=n_kf, n_points=n_pt, n_planes=max(2, n_kf // 3), n_cuboids=3 if n_kf >= 30 else 2), seed=1)
be = B.hip_backend()
for i in range(6):
    s = be.create(P, B.default_options()); s.run(); s.read(); s.close()
rows = []
for i in range(10):
    t0 = time.perf_counter(); s = be.create(P, B.default_options())
    t1 = time.perf_counter(); s.run()
    t2 = time.perf_counter(); s.reset(); t3 = time.perf_counter(); s.run()
    t4 = time.perf_counter(); s.read(); t5 = time.perf_counter(); s.close()
    rows.append((t1 - t0, t2 - t1, t4 - t3, t5 - t4, t3 - t2))
med = [1e3 * sorted(r[k] for r in rows)[len(rows) // 2] for k in range(5)]
print("%d KF / %d points: create %.3f ms, FIRST run %.3f ms, second run on the same handle %.3f ms, read %.3f ms, reset %.3f ms" % (n_kf, n_pt, *med[:4], med[4]))
sys.stdout.flush()
s = be.create(P, B.default_options(verbose=1)); s.run(); s.close()
