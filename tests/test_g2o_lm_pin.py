"""The reference's OWN Levenberg-Marquardt loop as golden data.

tests/golden/g2o_prebuilt_lm.json was produced by oracle/pin_g2o_lm.py, which runs g2o::SparseOptimizer::optimize +
g2o::OptimizationAlgorithmLevenberg::solve + the reprojection edges INSIDE the reference's prebuilt
Thirdparty/g2o/lib/libg2o.so on four small points-only windows, following the two-pass protocol of
Optimizer.cc:2727-2838 (5 iterations, chi2 / depth classification, 10 iterations).  Only the linear solve is not the
reference's (BlockSolverX / LinearSolverDense exist only in lib/libORB_SLAM2.so, which does not load): it is numpy on the
same linear system.  What is pinned here, against the binary: active sets and index mapping, computeActiveErrors +
activeRobustChi2, buildSystem, computeLambdaInit, setLambda / restoreDiagonal, update, push / pop, the gain ratio and the
lambda schedule, accept / reject, ORB-SLAM2's nBad stop rule, the stale-_error semantics of the classification (A18-A27 of
SURVEY.md section 8a restricted to the reprojection edges).

CPU: the oracle against the fixture.  GPU (-m gpu): the HIP library against the same fixture -- the product against the
reference's own LM run, no oracle in between.
"""
import json
import os

import numpy as np
import pytest

from ppo_ba import binding as B

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = json.load(open(os.path.join(HERE, "golden", "g2o_prebuilt_lm.json")))["cases"]

_EMPTY = ("plane_coef", "cuboid_qt", "cuboid_scale", "pl_plane", "pl_pose", "pl_meas", "pl_info", "pl_huber", "bbox_pose", "bbox_cuboid", "bbox_meas",
          "bbox_info", "bbox_huber", "bbox_K", "corner_pose", "corner_cuboid", "corner_meas", "corner_info", "corner_huber", "corner_K", "ptcub_cuboid",
          "ptcub_margin_ratio", "ptcub_info", "cubpl_cuboid", "cubpl_plane", "cubpl_meas", "cubpl_info", "cubpl_huber")


def problem_of(G):
    P = B.Problem()
    n = len(G["pose_qt"])
    P.set("pose_qt", G["pose_qt"]); P.set("pose_fixed", G["pose_fixed"]); P.set("pose_cam", [G["cam"]] * n)
    P.set("point_xyz", G["point_xyz"])
    for kind, D in (("mono", 2), ("stereo", 3)):
        E = np.array(G[kind], dtype=np.float64).reshape(-1, 3 + D)
        P.set(kind + "_point", E[:, 0].astype(np.int32)); P.set(kind + "_pose", E[:, 1].astype(np.int32))
        P.set(kind + "_obs", E[:, 2:2 + D].copy()); P.set(kind + "_info", E[:, 2 + D].copy())
        P.set(kind + "_huber", np.full(len(E), G["huber_" + kind]))
    for name in _EMPTY:
        P.set(name, np.zeros(0))
    return P


def check_against_reference_run(backend, case, tol1, tol2, tol_pose, tol_point):
    """tol1 / tol2 = (lambda, chi2) relative tolerances in pass 1 / pass 2."""
    G, R = case["graph"], case["result"]
    opt = B.default_options()
    s = backend.create(problem_of(G), opt)
    st = s.run()
    r = s.read()
    it = st["iters"]
    first = R["iter_first_trial"] + [len(R["lambda_trials"])]
    n_ref = len(R["iter_first_trial"])
    assert len(it) == n_ref == R["pass1_iterations"] + R["pass2_iterations"]
    assert sum(1 for a in it if a["pass_"] == 0) == R["pass1_iterations"]
    for k, a in enumerate(it):
        tol_lambda, tol_chi2 = tol1 if a["pass_"] == 0 else tol2
        assert a["trials"] == first[k + 1] - first[k], (k, a["trials"], first[k + 1] - first[k])  # the same accept / reject sequence
        lam_ref = R["lambda_trials"][first[k + 1] - 1]   # lambda of the iteration's last trial
        assert abs(a["lambda_"] - lam_ref) <= tol_lambda * lam_ref, (k, a["lambda_"], lam_ref)
        chi_ref = R["chi2_at_trial_start"][first[k]]      # activeRobustChi2 of the state the iteration linearised
        assert abs(a["chi2_before"] - chi_ref) <= tol_chi2 * chi_ref, (k, a["chi2_before"], chi_ref)
    assert abs(it[-1]["chi2_after"] - R["final_chi2"]) <= tol2[1] * R["final_chi2"]
    n_mono = len(G["mono"])
    lv = np.array(R["levels_after_pass1"])
    assert np.array_equal(r.mono_level, lv[:n_mono]) and np.array_equal(r.stereo_level, lv[n_mono:])
    ref_pose, ref_pt = np.array(R["final_state"]["pose_qt"]), np.array(R["final_state"]["point_xyz"])
    assert np.abs(r.pose_qt[:, 4:] - ref_pose[:, 4:]).max() < tol_pose
    # (angle as twice the chord between the unit quaternions: the arccos of conftest.rot_angle bottoms out at 3e-8)
    assert max(2 * min(np.linalg.norm(a[:4] - b[:4]), np.linalg.norm(a[:4] + b[:4])) for a, b in zip(r.pose_qt, ref_pose)) < tol_pose
    # Points are compared where pass 2 determines them: at least 5 active residual rows (two observations).  A point the culling
    # left with ONE observation (2 or 3 rows for 3 unknowns) sits on a flat direction of the cost: there the reference's own
    # result moves by 1e-4 m with the rounding of the linear solve (numpy's Cholesky here against the oracle's Schur
    # complement + LDLT, everything else being the reference binary), which is why DESIGN.md section 6 excludes such points.
    E = np.concatenate([np.array(G["mono"], dtype=np.float64).reshape(-1, 5)[:, 0], np.array(G["stereo"], dtype=np.float64).reshape(-1, 6)[:, 0]]).astype(int)
    rows = np.concatenate([np.full(n_mono, 2), np.full(len(E) - n_mono, 3)])
    nrows = np.zeros(len(ref_pt))
    np.add.at(nrows, E[lv == 0], rows[lv == 0])
    well = nrows >= 5
    assert well.sum() >= 0.3 * len(ref_pt)
    assert np.abs(r.point_xyz[well] - ref_pt[well]).max() < tol_point
    chi = np.concatenate([r.mono_chi2, r.stereo_chi2])
    ref_chi = np.array(R["final_edge_chi2"])
    act = (lv == 0) & well[E]
    assert np.abs(chi[act] - ref_chi[act]).max() <= 1e-4 * max(1.0, ref_chi[act].max())
    s.close()


def test_fixture_holds_rejected_trials_and_an_early_stop():
    """The six windows exercise what the schedule must reproduce: rejected trials (more solve attempts than iterations), the
    nBad stop rule (a pass that ends before its 10 iterations) and 10-60 % of the edges culled between the passes."""
    trials = [len(c["result"]["lambda_trials"]) - len(c["result"]["iter_first_trial"]) for c in CASES]
    assert sum(t > 0 for t in trials) >= 2 and max(trials) >= 3
    assert any(c["result"]["pass2_iterations"] < 10 for c in CASES)
    for c in CASES:
        lv = c["result"]["levels_after_pass1"]
        assert 0.05 < sum(lv) / len(lv) < 0.65


@pytest.mark.parametrize("k", range(len(CASES)))
def test_oracle_follows_the_reference_lm_run(oracle, k):
    """Pass 1 (5 iterations with the Huber kernel): the same trial sequence, lambda BIT-IDENTICAL in every iteration of all six
    windows (asserted to 1e-12; lambda is a smooth function of the gain ratio, i.e. of chi2 before / after and of
    x' (lambda x + b): its agreement pins them all), chi2 to 1e-12.  Identical outlier levels.  Pass 2 (kernel off, 10-60 % of
    the edges culled): same trial sequence and stop iteration; lambda / chi2 at BASELINE.md's bars because windows whose
    culling leaves single-observation points amplify the rounding of the linear solve (see the comment in the checker)."""
    check_against_reference_run(oracle, CASES[k], tol1=(1e-12, 1e-12), tol2=(1e-3, 1e-6), tol_pose=5e-6, tol_point=2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("k", range(len(CASES)))
def test_hip_follows_the_reference_lm_run(hip, k):
    """The product against the reference's own LM run (no oracle in between) at BASELINE.md's bars."""
    # (pass-2 lambda: a cubic function of the gain ratio, itself a difference of two chi2 values that agree to 1e-6 near convergence -- the oracle's own
    #  pass-2 bar against this fixture is 1e-3; the deterministic device run measures 1.5e-3 at the last iteration of window 1)
    check_against_reference_run(hip, CASES[k], tol1=(1e-6, 1e-9), tol2=(3e-3, 1e-6), tol_pose=1e-5, tol_point=2e-5)
