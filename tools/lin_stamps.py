"""Development aid: timeline of the fused linearisation's workgroups (test-hooks build, PPO_BA_LIN_STAMPS).

Prints, per part of the launch, when its workgroups enter and leave (µs after the first workgroup of the launch entered) and, for the
numerically differentiated classes, how long the phases inside a workgroup take (residual | wait for the others | H rows + atomics).
Usage (GPU box):  python tools/lin_stamps.py [config] [PPO_BA_LIN_PARTS mask]
"""
import os
import sys
import collections

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "point-plane-object-slam_amd"))
path = "/tmp/ppo_lin_stamps.txt"
os.environ["PPO_BA_LIN_STAMPS"] = path
if len(sys.argv) > 2:
    os.environ["PPO_BA_LIN_PARTS"] = sys.argv[2]
import numpy as np  # noqa: E402
from ppo_ba import binding as B, synth  # noqa: E402

P = synth.make_problem(sys.argv[1] if len(sys.argv) > 1 else "cfg4", seed=1)
s = B.hip_backend(test_hooks=True).create(P, B.default_options(jacobian_mode=int(os.environ.get("JAC", "0"))))  # JAC=1: analytic Jacobians
for _ in range(5):
    s.linearize()
s.close()
rows = collections.defaultdict(list)
for line in open(path):
    f = line.split()
    rows[f[0]].append([int(x) for x in f[1:]])


def q(v):
    v = np.sort(np.asarray(v, dtype=float)) * 0.01  # 10 ns ticks -> µs
    return "min %6.2f  med %6.2f  p90 %6.2f  max %6.2f" % (v[0], v[len(v) // 2], v[int(0.9 * (len(v) - 1))], v[-1])


for part, r in rows.items():
    a = np.array(r)
    print("%-7s %5d workgroups" % (part, len(a)))
    print("    entry      ", q(a[:, 0]))
    print("    exit       ", q(a[:, 4]))
    print("    inside     ", q(a[:, 4] - a[:, 0]))
    if part in ("plane", "bbox", "corner", "ptcub"):
        ok = a[:, 1] >= 0
        print("    residual   ", q((a[:, 1] - a[:, 0])[ok]))
        print("    barrier    ", q((a[:, 2] - a[:, 1])[ok]))
        print("    H + atomics", q((a[:, 3] - a[:, 2])[ok]))
        print("    chi2 sum   ", q((a[:, 4] - a[:, 3])[ok]))
