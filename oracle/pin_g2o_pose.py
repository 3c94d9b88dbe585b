#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: Optimizer::PoseOptimization's protocol (src/Optimizer.cc:247-459) run INSIDE the reference's prebuilt
Thirdparty/g2o/lib/libg2o.so.

What runs inside the reference binary (nothing below is restated):
  g2o::SparseOptimizer::initializeOptimization(0) / optimize(10)         (sparse_optimizer.cpp:199-267, 354-419)
  g2o::OptimizationAlgorithmLevenberg::solve / computeLambdaInit / computeScale with ORB-SLAM2's nBad stop rule
  g2o::SparseOptimizer::computeActiveErrors / activeRobustChi2 / update / push / pop / discardTop
  EdgeSE3ProjectXYZOnlyPose / EdgeStereoSE3ProjectXYZOnlyPose: computeError, BaseEdge::chi2,
  BaseUnaryEdge::linearizeOplus(JacobianWorkspace&) (-> the analytic linearizeOplus), BaseUnaryEdge::constructQuadraticForm,
  RobustKernelHuber, VertexSE3Expmap::oplusImpl and its backup stack.
The edges are hand-laid-out objects on top of the exported OptimizableGraph::Edge constructor (oracle/pin_g2o_prebuilt.py,
make_unary_edge: the OnlyPose classes export no constructor), wired with the exported addVertex / addEdge.

What does NOT exist in libg2o.so is BlockSolver_6_3 + LinearSolverDense (header templates, instantiated only inside
lib/libORB_SLAM2.so, which does not load here): as in oracle/pin_g2o_lm.py the abstract g2o::Solver is supplied through a
ctypes vtable -- buildSystem calls every active edge's own linearizeOplus + constructQuadraticForm, setLambda / restoreDiagonal
touch the 6x6 diagonal, solve() is a numpy Cholesky solve of the 6x6 system.

The protocol on top is OPT:375-451 line by line: four rounds; each resets the pose to the frame's initial pose (:385),
initializeOptimization(0), optimize(10); then every observation is re-classified -- an outlier's error is re-evaluated at the
round's final pose (:397-400, :426-429), `const float chi2 = e->chi2()` against 5.991f / 7.815f, setLevel, the robust kernel
dropped at the end of round 2 (:415-416, :444-445) -- and the loop stops after one round with fewer than 10 edges (:448-449);
fewer than 3 correspondences return 0 without optimising (:372-373).

  python oracle/pin_g2o_pose.py            regenerate tests/golden/g2o_prebuilt_pose.json (needs /root/reference)
"""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import pin_g2o_prebuilt as PB  # noqa: E402
from pin_g2o_lm import Lib, aligned, rot, u64, vp, V_FIXED, V_HESS, V_B, V_ID, E_LEVEL, E_KERNEL  # noqa: E402

GOLD = os.path.join(REPO, "tests", "golden", "g2o_prebuilt_pose.json")
_keep = []


def make_frame(seed, n_obs=120, stereo_frac=0.3, outlier_frac=0.15, pose_noise=(0.02, 0.05), depth=(1.0, 9.0)):
    """One frame in plain arrays: initial pose, intrinsics, fixed map points (float32-representable: OPT:318-320 reads
    Xw.at<float>), observations (u, v, u_right; u_right < 0 marks a monocular observation, OPT:294), invSigma2 (float)."""
    rng = np.random.default_rng(seed)
    cam = [float(np.float32(rng.uniform(400, 600))), float(np.float32(rng.uniform(400, 600))), 320.0, 240.0, 40.0]
    q = rng.normal(size=4) * [0.1, 0.1, 0.1, 1.0]
    q /= np.linalg.norm(q)
    if q[3] < 0:
        q = -q
    t = rng.normal(size=3) * 0.5
    R = rot(q)
    xw, obs, invs = [], [], []
    for _ in range(n_obs):
        pc = np.array([rng.normal() * 1.5, rng.normal() * 1.0, rng.uniform(*depth)])
        X = (R.T @ (pc - t)).astype(np.float32).astype(np.float64)
        pc = R @ X + t
        octave = int(rng.integers(0, 8))
        sig = 1.2 ** octave
        u = cam[0] * pc[0] / pc[2] + cam[2] + rng.normal() * sig
        v = cam[1] * pc[1] / pc[2] + cam[3] + rng.normal() * sig
        ur = u - cam[4] / pc[2] + rng.normal() * sig
        if rng.uniform() < outlier_frac:
            u, v = rng.uniform(0, 640), rng.uniform(0, 480)
            ur = u - rng.uniform(2, 40)
        stereo = rng.uniform() < stereo_frac and ur >= 0
        xw.append(X.tolist())
        obs.append([float(np.float32(u)), float(np.float32(v)), float(np.float32(ur)) if stereo else -1.0])
        invs.append(float(np.float32(1.0 / (sig * sig))))
    # the frame's initial pose: the true pose perturbed (a tracked frame starts from the motion model's prediction)
    dq = np.concatenate([rng.normal(size=3) * pose_noise[0] / 2, [1.0]])
    dq /= np.linalg.norm(dq)
    x1, y1, z1, w1 = dq
    x2, y2, z2, w2 = q
    q0 = np.array([w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2, w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2, w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2])
    q0 /= np.linalg.norm(q0)
    if q0[3] < 0:
        q0 = -q0
    t0 = t + rng.normal(size=3) * pose_noise[1]
    return {"seed": seed, "cam": cam, "pose_qt": q0.tolist() + t0.tolist(), "xw": xw, "obs": obs, "inv_sigma2": invs,
            "huber_mono": float(np.float32(np.sqrt(5.991))), "huber_stereo": float(np.float32(np.sqrt(7.815)))}


class PoseDriver:
    def __init__(self, F):
        L = self.L = Lib()
        lib = L.lib
        self.F = F
        self.opt = aligned(1 << 16)
        L.fn("_ZN3g2o15SparseOptimizerC1Ev", vp)(self.opt.ctypes.data)
        self.log = {"lambda_trials": [], "chi2_at_trial_start": [], "solve_ok": [], "iter_first_trial": []}
        addV = L.fn("_ZN3g2o16OptimizableGraph9addVertexEPNS_10HyperGraph6VertexEPNS0_4DataE", vp, vp, vp, restype=C.c_bool)
        addE = L.fn("_ZN3g2o16OptimizableGraph7addEdgeEPNS_10HyperGraph4EdgeE", vp, vp, restype=C.c_bool)
        v = self.v = aligned(0x200)
        L.fn("_ZN3g2o15VertexSE3ExpmapC1Ev", vp)(v.ctypes.data)
        self.set_pose(F["pose_qt"])
        v.view(np.int32)[V_ID // 4] = 0
        v.view(np.uint8)[V_FIXED] = 0
        assert addV(self.opt.ctypes.data, v.ctypes.data, None)
        # mono edges first, then stereo (the oracle and the kernel evaluate them in that order; the reference adds them in feature
        # order, which only permutes the summation)
        self.edges = []  # (kind, D, buffer, J, workspace, observation index)
        for kind in ("mono", "stereo"):
            K = PB._UNARY[kind]
            for i, (X, ob, w) in enumerate(zip(F["xw"], F["obs"], F["inv_sigma2"])):
                if (ob[2] < 0) != (kind == "mono"):
                    continue
                e, J = PB.make_unary_edge(lib, kind, v, X, F["cam"], ob, w, F["huber_" + kind], _keep)
                ws = PB.fake_jacobian_workspace(J, _keep)
                assert addE(self.opt.ctypes.data, e.ctypes.data)
                self.edges.append((kind, K["D"], e, J, ws, i))
        self.f_err = {k: L.fn(PB._UNARY[k]["err_fn"], vp) for k in ("mono", "stereo")}
        self.f_chi2 = {k: L.fn(PB._UNARY[k]["chi2"], vp, restype=C.c_double) for k in ("mono", "stereo")}
        self.f_lin = {D: L.fn(PB._BU[D] + "14linearizeOplusERNS_17JacobianWorkspaceE", vp, vp) for D in (2, 3)}
        self.f_cqf = {D: L.fn(PB._BU[D] + "22constructQuadraticFormEv", vp) for D in (2, 3)}
        BOOL = C.c_bool
        self.cbs = []

        def cb(restype, *argtypes):
            def deco(f):
                c = C.CFUNCTYPE(restype, *argtypes)(f)
                self.cbs.append(c)
                return c
            return deco
        slots = [
            cb(None, vp)(lambda this: None), cb(None, vp)(lambda this: None),
            cb(BOOL, vp, vp, BOOL)(lambda this, o, online: True),
            cb(BOOL, vp, BOOL)(lambda this, z: self.build_structure()),
            cb(BOOL, vp, vp, vp)(lambda this, a, b: False),
            cb(BOOL, vp)(lambda this: self.build_system()),
            cb(BOOL, vp)(lambda this: self.solve()),
            cb(BOOL, vp, vp, vp)(lambda this, a, b: False),
            cb(BOOL, vp, C.c_double, BOOL)(lambda this, lam, backup: self.set_lambda(lam)),
            cb(None, vp)(lambda this: self.restore_diagonal()),
            cb(BOOL, vp)(lambda this: False), cb(BOOL, vp)(lambda this: False), cb(None, vp, BOOL)(lambda this, s: None),
            cb(None, vp, BOOL)(lambda this, s: None), cb(BOOL, vp)(lambda this: False), cb(BOOL, vp, vp)(lambda this, s: False),
        ]
        self.vtable = (C.c_void_p * len(slots))(*[C.cast(s, C.c_void_p).value for s in slots])
        self.solver = aligned(128)
        u64(self.solver)[0] = C.addressof(self.vtable)
        u64(self.solver)[1] = self.opt.ctypes.data
        u64(self.solver)[6] = 1
        self.H = aligned(8 * 36 + 64)
        self.x = aligned(8 * 6 + 64)
        self.b = aligned(8 * 6 + 64)
        u64(self.solver)[2] = self.x.ctypes.data
        u64(self.solver)[3] = self.b.ctypes.data
        u64(self.solver)[4] = 6
        u64(self.solver)[5] = 6
        self.algo = aligned(1 << 12)
        L.fn("_ZN3g2o30OptimizationAlgorithmLevenbergC1EPNS_6SolverE", vp, vp)(self.algo.ctypes.data, self.solver.ctypes.data)
        L.fn("_ZN3g2o15SparseOptimizer12setAlgorithmEPNS_21OptimizationAlgorithmE", vp, vp)(self.opt.ctypes.data, self.algo.ctypes.data)
        self.f_init = L.fn("_ZN3g2o15SparseOptimizer22initializeOptimizationEi", vp, C.c_int, restype=C.c_bool)
        self.f_optimize = L.fn("_ZN3g2o15SparseOptimizer8optimizeEib", vp, C.c_int, C.c_bool, restype=C.c_int)
        self.f_rchi2 = L.fn("_ZNK3g2o15SparseOptimizer16activeRobustChi2Ev", vp, restype=C.c_double)

    def set_pose(self, qt):
        self.v[0xb0 // 8:0xb0 // 8 + 4] = qt[:4]
        self.v[0xd0 // 8:0xd0 // 8 + 3] = qt[4:]

    def pose(self):
        return self.v[0xb0 // 8:0xb0 // 8 + 4].tolist() + self.v[0xd0 // 8:0xd0 // 8 + 3].tolist()

    # ---- the linear-solver side (BlockSolver_6_3 with no landmarks: one 6x6 block) ----
    def build_structure(self):
        u64(self.v)[V_HESS // 8] = self.H.ctypes.data
        self.act = [t for t in self.edges if t[2].view(np.int32)[E_LEVEL // 4] == 0]
        return True

    def build_system(self):
        self.log["iter_first_trial"].append(len(self.log["lambda_trials"]))
        self.H[:36] = 0.0
        self.v[V_B // 8:V_B // 8 + 6] = 0.0
        for kind, D, e, J, ws, i in self.act:
            self.f_lin[D](e.ctypes.data, ws.ctypes.data)
            self.f_cqf[D](e.ctypes.data)
        self.b[:6] = self.v[V_B // 8:V_B // 8 + 6]
        return True

    def set_lambda(self, lam):
        self.log["lambda_trials"].append(float(lam))
        self.log["chi2_at_trial_start"].append(float(self.f_rchi2(self.opt.ctypes.data)))  # the kept state's errors on a first attempt
        self.backup = self.H[:36:7].copy()
        self.H[:36:7] += lam
        return True

    def restore_diagonal(self):
        self.H[:36:7] = self.backup

    def solve(self):
        Hm = self.H[:36].reshape(6, 6).T.copy()
        ok = True
        try:
            Lc = np.linalg.cholesky(Hm)
            self.x[:6] = np.linalg.solve(Lc.T, np.linalg.solve(Lc, self.b[:6]))
        except np.linalg.LinAlgError:
            ok = False  # x keeps the previous solution, as g2o's _x does
        self.log["solve_ok"].append(bool(ok))
        return ok

    # ---- OPT:372-458 ----
    def run(self, rounds=4, iters=10):
        F = self.F
        n = len(F["obs"])
        out = {"n_correspondences": n, "rounds": []}
        outlier = [0] * n
        if n < 3:  # :372-373
            out["n_inliers"] = 0
            out["final_pose_qt"] = F["pose_qt"]
            out["outlier"] = outlier
            return out
        nBad = 0
        for it in range(rounds):
            self.set_pose(F["pose_qt"])                                   # :385
            assert self.f_init(self.opt.ctypes.data, 0) or True            # :386 (false when no edge is active)
            trials0 = len(self.log["lambda_trials"])
            n_it = self.f_optimize(self.opt.ctypes.data, iters, False)    # :387
            chi2_end = float(self.f_rchi2(self.opt.ctypes.data))          # of the errors the last attempt left behind
            nBad = 0
            chi2s = [0.0] * n
            for kind, D, e, J, ws, i in self.edges:                      # :390-417, :419-446
                if outlier[i]:
                    self.f_err[kind](e.ctypes.data)
                c = float(self.f_chi2[kind](e.ctypes.data))
                chi2 = np.float32(c)
                thr = np.float32(5.991) if kind == "mono" else np.float32(7.815)
                if chi2 > thr:
                    outlier[i] = 1
                    e.view(np.int32)[E_LEVEL // 4] = 1
                    nBad += 1
                else:
                    outlier[i] = 0
                    e.view(np.int32)[E_LEVEL // 4] = 0
                if it == 2:
                    u64(e)[E_KERNEL // 8] = 0
                chi2s[i] = c
            out["rounds"].append({"lm_iterations": int(n_it), "robust_chi2_after_optimize": chi2_end, "lm_trials": len(self.log["lambda_trials"]) - trials0, "pose_qt": self.pose(),
                                  "outlier": list(outlier), "chi2": chi2s, "n_bad": nBad})
            if len(self.edges) < 10:  # :448-449
                break
        out["lambda_trials"] = self.log["lambda_trials"]
        out["chi2_at_trial_start"] = self.log["chi2_at_trial_start"]
        out["solve_ok"] = self.log["solve_ok"]
        out["iter_first_trial"] = self.log["iter_first_trial"]
        out["final_pose_qt"] = self.pose()
        out["outlier"] = list(outlier)
        out["n_inliers"] = n - nBad  # :458
        return out


CASES = [dict(seed=11), dict(seed=12, n_obs=300, stereo_frac=0.0), dict(seed=13, n_obs=200, stereo_frac=1.0, outlier_frac=0.3),
         dict(seed=14, n_obs=80, outlier_frac=0.5, pose_noise=(0.1, 0.3)),       # far start, half the observations wrong: rejected trials
         dict(seed=15, n_obs=9, outlier_frac=0.0), dict(seed=16, n_obs=2), dict(seed=17, n_obs=40, stereo_frac=0.5, outlier_frac=0.1, depth=(0.4, 2.0)),
         dict(seed=18, n_obs=600, stereo_frac=0.2, outlier_frac=0.05, pose_noise=(0.005, 0.01))]


def generate():
    out = {"source": "Thirdparty/g2o/lib/libg2o.so (prebuilt): SparseOptimizer + OptimizationAlgorithmLevenberg + the OnlyPose edges under the protocol of "
                     "Optimizer.cc:375-451; 6x6 linear solve by numpy", "cases": []}
    for kw in CASES:
        F = make_frame(**kw)
        res = PoseDriver(F).run()
        out["cases"].append({"frame": F, "result": res})
        print("seed %d: %d observations, rounds %s, inliers %d, %d trials" %
              (F["seed"], len(F["obs"]), [(r["lm_iterations"], r["lm_trials"], r["n_bad"]) for r in res["rounds"]], res["n_inliers"], len(res.get("lambda_trials", []))))
    with open(GOLD, "w") as fh:
        json.dump(out, fh)
    print("wrote", GOLD)


if __name__ == "__main__":
    generate()
    sys.stdout.flush()
    os._exit(0)  # the library's objects live in numpy buffers: skip every destructor
