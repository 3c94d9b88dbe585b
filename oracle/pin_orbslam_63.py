#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: the points-only local BA and the pose-only optimisation run ENTIRELY inside the reference's prebuilt binaries, through
the SOLVER CLASSES THE REFERENCE INSTANTIATES FOR THEM (the last numpy solvers retired).

  Optimizer::LocalBundleAdjustment (src/Optimizer.cc:461-787)   BlockSolver_6_3 + LinearSolverEigen<BlockSolver_6_3::PoseMatrixType>   (:516-520)
  Optimizer::PoseOptimization      (src/Optimizer.cc:247-459)   BlockSolver_6_3 + LinearSolverDense<BlockSolver_6_3::PoseMatrixType>   (:250-254)

Both instantiations exist only inside lib/libORB_SLAM2.so (nm -DC: g2o::BlockSolver<g2o::BlockSolverTraits<6, 3> >::solve, ::buildSystem, ::setLambda, the
constructor; g2o::LinearSolverEigen<Eigen::Matrix<double, 6, 6, 0, 6, 6> >::solve / ::computeSymbolicDecomposition; g2o::LinearSolverDense<Eigen::Matrix<
double, 6, 6, 0, 6, 6> >::solve), which is loaded as oracle/orbslam_loader.py describes (trap armed; `armed hits: 0` goes into the fixture).  What runs
inside the binaries, beyond what oracle/pin_orbslam_lm.py lists: BlockSolver<6,3>::buildStructure / buildSystem / setLambda / solve (fixed-size 6 x 6 and
3 x 3 blocks, the Schur complement over SparseBlockMatrixCCS / the hash-map variant, restoreDiagonal, the back-substitution), Eigen::SimplicialLDLT's
analyzePattern (AMD ordering) / factorize / solve through LinearSolverEigen, Eigen::LDLT<MatrixXd> through LinearSolverDense.

The two linear solvers have inline constructors; the objects are completed exactly as the compiled call sites do it (disassembly of
Optimizer::LocalBundleAdjustment at +0x2e2: operator new(0x110), _init = 1, _blockOrdering = 0, _writeDebug = 0, the library's own exported
Eigen::SparseMatrix<double, 0, int>::SparseMatrix() on the two sparse matrices at +0x10 and +0x68, zeros, m_shiftScale = 1.0 at +0x108, vtable + 16;
Optimizer::PoseOptimization at +0x41: operator new(0x78), _reset = 1, zeros, Eigen::LDLT m_sign = 2, vtable + 16).  For the log, three slots of a COPY of
the BlockSolver's vtable (buildSystem, solve, setLambda) point at trampolines that record and then call the reference's own function.

What is restated here in Python is the host protocol around the solver, as in oracle/pin_orbslam_lm.py (5 iterations, classification, 10 more --
Optimizer.cc:461-787 is that protocol on mono / stereo edges only) and oracle/pin_g2o_pose.py (four rounds, OPT:375-451).

  python oracle/pin_orbslam_63.py            regenerate tests/golden/orbslam_prebuilt_63.json (needs /root/reference)
"""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(REPO, "point-plane-object-slam_amd"))
GOLD = os.path.join(REPO, "tests", "golden", "orbslam_prebuilt_63.json")

import orbslam_loader as loader  # noqa: E402

loader.load()  # BEFORE anything opens libg2o.so on its own: one instance of the library, under the trap

import orbslam_objects as O  # noqa: E402
import pin_g2o_prebuilt as PB  # noqa: E402
import pin_orbslam_lm as LM  # noqa: E402
import pin_g2o_pose as POSE  # noqa: E402

vp = C.c_void_p
BS63 = "g2o::BlockSolver<g2o::BlockSolverTraits<6, 3> >"
M66 = "Eigen::Matrix<double, 6, 6, 0, 6, 6>"
LSE66 = "g2o::LinearSolverEigen<%s >" % M66
LSD66 = "g2o::LinearSolverDense<%s >" % M66
BS63_CTOR = BS63 + "::BlockSolver(g2o::LinearSolver<%s >*)" % M66


def linear_solver_eigen(R):
    """new LinearSolverEigen<PoseMatrixType>() as Optimizer::LocalBundleAdjustment compiles it (Optimizer.cc:516)."""
    ls = O.aligned(0x110)
    O.u64(ls)[0] = R.vtable(LSE66)
    b = ls.view(np.uint8)
    b[0x8], b[0x9], b[0xa] = 1, 0, 0  # _init(true), _blockOrdering(false), _writeDebug(false)
    sm = R.fn("Eigen::SparseMatrix<double, 0, int>::SparseMatrix()", vp)
    sm(ls.ctypes.data + 0x10)   # _sparseMatrix
    sm(ls.ctypes.data + 0x68)   # SimplicialLDLT::m_matrix
    ls[0x108 // 8] = 1.0        # m_shiftScale (m_shiftOffset, the permutations, m_diag, the flags: zero)
    return ls


def linear_solver_dense(R):
    """new LinearSolverDense<PoseMatrixType>() as Optimizer::PoseOptimization compiles it (Optimizer.cc:250)."""
    ls = O.aligned(0x80)
    O.u64(ls)[0] = R.vtable(LSD66)
    ls.view(np.uint8)[0x8] = 1        # _reset(true)
    ls.view(np.int32)[0x68 // 4] = 2  # Eigen::LDLT m_sign(ZeroSign)
    return ls


class Window63(LM.RefRun):
    """A points-only window under BlockSolver_6_3 + LinearSolverEigen (Optimizer.cc:516-520)."""

    def _make_solver(self):
        R = self.W.R
        self.linear_solver = linear_solver_eigen(R)
        self.solver = O.aligned(0x100)
        R.fn(BS63_CTOR, vp, vp)(self.solver.ctypes.data, self.linear_solver.ctypes.data)


class Pose63(POSE.PoseDriver):
    """One frame under BlockSolver_6_3 + LinearSolverDense (Optimizer.cc:250-254); the protocol (run) is POSE.PoseDriver's."""

    def __init__(self, F):  # noqa: super().__init__ is not called: it builds the numpy stand-in this class replaces
        from pin_g2o_lm import Lib, aligned, u64, V_FIXED, V_ID
        L = self.L = Lib()
        lib = L.lib
        R = self.R = O.Ref()
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
        self.edges = []
        for kind in ("mono", "stereo"):
            K = PB._UNARY[kind]
            for i, (X, ob, w) in enumerate(zip(F["xw"], F["obs"], F["inv_sigma2"])):
                if (ob[2] < 0) != (kind == "mono"):
                    continue
                e, J = PB.make_unary_edge(lib, kind, v, X, F["cam"], ob, w, F["huber_" + kind], POSE._keep)
                assert addE(self.opt.ctypes.data, e.ctypes.data)
                self.edges.append((kind, K["D"], e, J, None, i))
        self.f_err = {k: L.fn(PB._UNARY[k]["err_fn"], vp) for k in ("mono", "stereo")}
        self.f_chi2 = {k: L.fn(PB._UNARY[k]["chi2"], vp, restype=C.c_double) for k in ("mono", "stereo")}
        self.f_rchi2 = L.fn("_ZNK3g2o15SparseOptimizer16activeRobustChi2Ev", vp, restype=C.c_double)
        # ---- Optimizer.cc:250-256 ----
        self.linear_solver = linear_solver_dense(R)
        bs = self.solver = O.aligned(0x100)
        R.fn(BS63_CTOR, vp, vp)(bs.ctypes.data, self.linear_solver.ctypes.data)
        self._trampolines()
        self.algo = aligned(1 << 12)
        L.fn("_ZN3g2o30OptimizationAlgorithmLevenbergC1EPNS_6SolverE", vp, vp)(self.algo.ctypes.data, bs.ctypes.data)
        L.fn("_ZN3g2o15SparseOptimizer12setAlgorithmEPNS_21OptimizationAlgorithmE", vp, vp)(self.opt.ctypes.data, self.algo.ctypes.data)
        self.f_init = L.fn("_ZN3g2o15SparseOptimizer22initializeOptimizationEi", vp, C.c_int, restype=C.c_bool)
        self.f_optimize = L.fn("_ZN3g2o15SparseOptimizer8optimizeEib", vp, C.c_int, C.c_bool, restype=C.c_int)

    def _trampolines(self):
        bs = self.solver
        vt0 = int(O.u64(bs)[0])
        n_slots = 17
        src = C.cast(vt0 - 16, C.POINTER(C.c_uint64))
        self.vt_copy = O.aligned(8 * (n_slots + 2) + 64)
        for i in range(n_slots + 2):
            O.u64(self.vt_copy)[i] = src[i]
        orig = {s: C.cast(vt0 + s, C.POINTER(C.c_uint64))[0] for s in (LM.SLOT_BUILD_SYSTEM, LM.SLOT_SOLVE, LM.SLOT_SET_LAMBDA)}
        f_build = C.CFUNCTYPE(C.c_bool, vp)(orig[LM.SLOT_BUILD_SYSTEM])
        f_solve = C.CFUNCTYPE(C.c_bool, vp)(orig[LM.SLOT_SOLVE])
        f_lambda = C.CFUNCTYPE(C.c_bool, vp, C.c_double, C.c_bool)(orig[LM.SLOT_SET_LAMBDA])

        def t_build(this):
            self.log["iter_first_trial"].append(len(self.log["lambda_trials"]))
            return f_build(this)

        def t_lambda(this, lam, backup):
            self.log["lambda_trials"].append(float(lam))
            self.log["chi2_at_trial_start"].append(float(self.f_rchi2(self.opt.ctypes.data)))
            return f_lambda(this, lam, backup)

        def t_solve(this):
            ok = bool(f_solve(this))
            self.log["solve_ok"].append(ok)
            return ok
        self._cbs = [C.CFUNCTYPE(C.c_bool, vp)(t_build), C.CFUNCTYPE(C.c_bool, vp)(t_solve), C.CFUNCTYPE(C.c_bool, vp, C.c_double, C.c_bool)(t_lambda)]
        for slot, cb in zip((LM.SLOT_BUILD_SYSTEM, LM.SLOT_SOLVE, LM.SLOT_SET_LAMBDA), self._cbs):
            O.u64(self.vt_copy)[2 + slot // 8] = C.cast(cb, C.c_void_p).value
        O.u64(bs)[0] = self.vt_copy.ctypes.data + 16


# points-only windows (Optimizer::LocalBundleAdjustment has no other vertex or edge class): the two of oracle/pin_orbslam_lm.py and three more
WINDOWS = [
    (dict(n_kf=9, n_points=150, n_planes=0, n_cuboids=0), dict(seed=8, stereo_frac=0.4)),
    (dict(n_kf=12, n_points=200, n_planes=0, n_cuboids=0), dict(seed=9, noise_scale=4.0, outlier_frac=0.1)),
    (dict(n_kf=6, n_points=60, n_planes=0, n_cuboids=0), dict(seed=21, stereo_frac=1.0)),
    (dict(n_kf=15, n_points=300, n_planes=0, n_cuboids=0), dict(seed=22, noise_scale=6.0, outlier_frac=0.2, fixed_observer_frac=0.2)),  # far start: rejected trials
    (dict(n_kf=20, n_points=400, n_planes=0, n_cuboids=0), dict(seed=23, stereo_frac=0.2, outlier_frac=0.05)),
]


def generate():
    from ppo_ba import binding as B
    from ppo_ba import synth
    out = {"source": "lib/libORB_SLAM2.so + Thirdparty/g2o/lib/libg2o.so (prebuilt): BlockSolver<BlockSolverTraits<6,3>> + LinearSolverEigen<Matrix<6,6>> "
                     "(Optimizer::LocalBundleAdjustment, OPT:516-520) and + LinearSolverDense<Matrix<6,6>> (Optimizer::PoseOptimization, OPT:250-254)",
           "windows": [], "frames": []}
    o = B.default_options()
    opt = {k: getattr(o, k) for k in LM.OPT_KEYS}
    for cfg, kw in WINDOWS:
        P = synth.make_problem(cfg, **kw)
        A = {k: v.tolist() for k, v in P.arrays.items()}
        Ashaped = {k: np.asarray(v).reshape(-1, B._WIDTH[k][1]) if (k in B._WIDTH and B._WIDTH[k][1] > 1) else np.asarray(v).reshape(-1) for k, v in P.arrays.items()}
        res = Window63(Ashaped, opt).run()
        out["windows"].append({"config": cfg, "kwargs": kw, "problem": A, "options": opt, "result": res})
        print("window seed %d: %d + %d iterations, %d trials, final chi2 %.6f, level 1: %d" %
              (kw["seed"], res["pass1_iterations"], res["pass2_iterations"], len(res["lambda_trials"]), res["final_chi2"], sum(sum(v) for v in res["levels_after_pass1"].values())))
    for kw in POSE.CASES:
        F = POSE.make_frame(**kw)
        res = Pose63(F).run()
        out["frames"].append({"frame": F, "result": res})
        print("frame seed %d: %d observations, rounds %s, inliers %d, %d trials" %
              (F["seed"], len(F["obs"]), [(r["lm_iterations"], r["lm_trials"], r["n_bad"]) for r in res["rounds"]], res["n_inliers"], len(res.get("lambda_trials", []))))
    info = loader.info()
    assert info["armed_hits"] == 0
    out["trap"] = {"symbols_trapped": info["trap_symbols"], "sink_hits_during_pinned_calls": info["armed_hits"]}
    with open(GOLD, "w") as fh:
        json.dump(out, fh)
    print("wrote", GOLD, out["trap"])


if __name__ == "__main__":
    generate()
    sys.stdout.flush()
    os._exit(0)  # the library's objects live in numpy buffers: skip every destructor
