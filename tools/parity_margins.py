import sys
import numpy as np
sys.path.insert(0, "/root/repo/point-plane-object-slam_amd"); sys.path.insert(0, "/root/repo/oracle"); sys.path.insert(0, "/root/repo/tests")
from ppo_ba import binding as B, synth
import oracle_py
CASES = [("tiny", dict(seed=1)), ("small", dict(seed=1)), ("small", dict(seed=2, stereo_frac=0.3)), ("small", dict(seed=3, with_corners=True)),
         ("small", dict(seed=4, with_vp_planes=2)), ("small", dict(seed=5, fixed_observer_frac=0.3)),
         (dict(n_kf=12, n_points=400, n_planes=0, n_cuboids=0), dict(seed=6)), (dict(n_kf=12, n_points=400, n_planes=6, n_cuboids=0), dict(seed=7))]
for cfg, kw in CASES:
    for elim in (1, 0):
        P = synth.make_problem(cfg, **kw)
        o = B.default_options(eliminate_cuboids=elim)
        so = oracle_py.backend().create(P, o); to = so.run(); so.close()
        worst = 0.0; sched_bad = 0
        for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
            sh = B.hip_backend().create(P, o); th = sh.run(); sh.close()
            if [(i["trials"], i["accepted"], i["result"]) for i in to["iters"]] != [(i["trials"], i["accepted"], i["result"]) for i in th["iters"]]:
                sched_bad += 1; continue
            for a, b in zip(to["iters"], th["iters"]):
                rr = max(abs(a["chi2_after"] - b["chi2_after"]) / abs(a["chi2_after"]), abs(a["chi2_before"] - b["chi2_before"]) / abs(a["chi2_before"]))
                if rr > 6e-7:
                    print("   OUTLIER rep", rep, "pass", a["pass_"], "iter", a["iteration"], "rel %.2e" % rr, "trials", a["trials"], b["trials"], flush=True)
                worst = max(worst, abs(a["chi2_after"] - b["chi2_after"]) / abs(a["chi2_after"]), abs(a["chi2_before"] - b["chi2_before"]) / abs(a["chi2_before"]))
        print(cfg, kw, "elim", elim, "worst chi2 rel %.2e  schedule mismatches %d/N" % (worst, sched_bad), flush=True)
