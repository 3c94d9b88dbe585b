"""The points-only local BA and the pose-only optimisation against the reference's OWN runs through the solver classes it instantiates for them.

tests/golden/orbslam_prebuilt_63.json was produced by oracle/pin_orbslam_63.py inside the reference's prebuilt lib/libORB_SLAM2.so + libg2o.so
(trap armed, `sink_hits_during_pinned_calls` = 0):
  windows   five points-only windows, 5 + classify + 10 (Optimizer::LocalBundleAdjustment, Optimizer.cc:461-787) under
            BlockSolver<BlockSolverTraits<6,3>> + LinearSolverEigen<Matrix<6,6>> (OPT:516-520: fixed-size blocks, Eigen::SimplicialLDLT, AMD ordering)
  frames    the eight frames of tests/golden/g2o_prebuilt_pose.json, four rounds (Optimizer::PoseOptimization, OPT:247-459) under
            BlockSolver<BlockSolverTraits<6,3>> + LinearSolverDense<Matrix<6,6>> (OPT:250-254)
The earlier fixtures pinned these two paths through BlockSolverX + LinearSolverDense<MatrixXd> (windows) and a numpy 6 x 6 solve (frames); they stay as
regression data.  The checks are the ones of tests/test_orbslam_lm_pin.py and tests/test_g2o_pose_pin.py, pointed at this fixture.
CPU: the oracle.  GPU (-m gpu): the HIP library through the C ABI -- the product against the reference, no oracle in between.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

import test_g2o_pose_pin as TP
import test_orbslam_lm_pin as TL
from ppo_ba import binding as B
from ppo_ba import pose as PO

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "orbslam_prebuilt_63.json")) as _fh:
    _GOLD = json.load(_fh)
WINDOWS, FRAMES = _GOLD["windows"], _GOLD["frames"]
W_IDS = ["seed%d" % w["kwargs"]["seed"] for w in WINDOWS]
F_IDS = ["seed%d" % f["frame"]["seed"] for f in FRAMES]


def test_fixture_is_the_reference_through_the_6_3_solvers():
    assert _GOLD["trap"]["sink_hits_during_pinned_calls"] == 0
    assert "BlockSolverTraits<6,3>" in _GOLD["source"] and "LinearSolverEigen" in _GOLD["source"] and "LinearSolverDense<Matrix<6,6>>" in _GOLD["source"]
    assert len(WINDOWS) >= 5 and len(FRAMES) == 8
    assert all(TL.is_points_only(w) for w in WINDOWS) and all(all(w["result"]["solve_ok"]) for w in WINDOWS)
    rejected = [len(w["result"]["lambda_trials"]) - len(w["result"]["iter_first_trial"]) for w in WINDOWS]
    assert sum(r > 0 for r in rejected) >= 3 and max(rejected) >= 4
    assert any(w["result"]["pass2_iterations"] < 10 for w in WINDOWS)  # the nBad stop
    # the same frames as the numpy-solved fixture: the reference's dense 6 x 6 LDLT takes the same rounds, iteration for iteration
    old = {c["frame"]["seed"]: c["result"] for c in TP.CASES}
    for f in FRAMES:
        o, r = old[f["frame"]["seed"]], f["result"]
        assert [x["lm_iterations"] for x in o["rounds"]] == [x["lm_iterations"] for x in r["rounds"]]
        assert o["outlier"] == r["outlier"] and o["n_inliers"] == r["n_inliers"]


@pytest.mark.parametrize("k", range(len(WINDOWS)), ids=W_IDS)
def test_oracle_follows_the_reference_local_bundle_adjustment(oracle, k):
    """Analytic Jacobians, bit-identical blocks: lambda of every pass-1 iteration to 1e-11 and chi2 to 1e-12 against the reference's
    BlockSolver_6_3 + LinearSolverEigen (a sparse LDLT in another elimination order than the oracle's dense one: the solutions agree to rounding)."""
    TL.check_against_reference_run(oracle, WINDOWS[k], tol_lambda=1e-5, tol_chi2=1e-9, tol_pass1=(1e-11, 1e-12))


@pytest.mark.parametrize("k", range(len(WINDOWS)), ids=W_IDS)
def test_oracle_first_reduced_solve_matches_block_solver_6_3_and_simplicial_ldlt(oracle, oracle_lib, k):
    case = WINDOWS[k]
    sysm = case["result"]["first_system"][0]
    assert sysm["trial"] == 0 and sysm["ok"]
    dp = C.POINTER(C.c_double)
    s = oracle.create(TL.problem_of(case), B.default_options())
    chi = C.c_double()
    oracle_lib.oracle_ba_linearize(s.h, C.byref(chi))
    assert abs(chi.value - case["result"]["chi2_at_trial_start"][0]) <= 1e-12 * chi.value
    n = C.c_int32()
    oracle_lib.oracle_ba_reduced(s.h, sysm["lambda"], 0, -1, 1, None, None, C.byref(n))
    n = n.value
    S, bs, x = np.zeros((n, n)), np.zeros(n), np.zeros(n)
    oracle_lib.oracle_ba_reduced(s.h, sysm["lambda"], 0, -1, 1, S.ctypes.data_as(dp), bs.ctypes.data_as(dp), None)
    assert oracle_lib.oracle_ldlt_solve(S.ctypes.data_as(dp), n, bs.ctypes.data_as(dp), x.ctypes.data_as(dp)) == 1
    x_ref = np.array(sysm["x"])[:n]
    dev = np.abs(x - x_ref).max() / np.abs(x_ref).max()
    assert dev <= 1e-10, dev
    s.close()


@pytest.mark.parametrize("case", FRAMES, ids=F_IDS)
def test_oracle_follows_the_reference_pose_optimization_through_linear_solver_dense(oracle_lib, case):
    TP.check_case(oracle_lib, "oracle_", case, 1e-9, 1e-9, True)


def test_oracle_lambda_and_trial_schedule_match_the_reference_in_every_round(oracle_lib):
    """As tests/test_g2o_pose_pin.py, with Eigen's LDLT of the 6 x 6 system in place of numpy's Cholesky: every strict LM iteration takes the same number of
    solve attempts, lambda to 1e-6 (bit-identical in most), chi2 at entry to 1e-9."""
    n_strict = n_close = n_all = 0
    for case in FRAMES:
        F, R = case["frame"], case["result"]
        if not R["rounds"]:
            continue
        ref = TP.ref_iterations(R)
        got = TP.oracle_trace(oracle_lib, TP.frames_of(F), PO.default_options())
        assert len(got) == len(ref), F["seed"]
        for k, (a, b) in enumerate(zip(got, ref)):
            n_all += 1
            assert a.pass_ == b["round"]
            assert abs(a.chi2_before - b["chi2"]) <= 1e-9 * b["chi2"], (F["seed"], k)
            if not b["strict"]:
                continue
            n_strict += 1
            assert a.trials == b["trials"], (F["seed"], k, a.trials, b["trials"])
            assert abs(a.lambda_ - b["lam"]) <= 1e-6 * b["lam"], (F["seed"], k, a.lambda_, b["lam"])
            n_close += int(abs(a.lambda_ - b["lam"]) <= 1e-12 * b["lam"])
    assert n_strict >= 0.8 * n_all and n_close >= 0.8 * n_strict, (n_all, n_strict, n_close)


@pytest.mark.gpu
@pytest.mark.parametrize("k", range(len(WINDOWS)), ids=W_IDS)
def test_hip_follows_the_reference_local_bundle_adjustment(hip, k):
    """The product (default = deterministic mode: no run-to-run component, LM decisions on the device) against the reference's own LocalBundleAdjustment-shaped run: schedule, lambda, chi2 per
    iteration up to the first marginal one, levels, final chi2, final state at BASELINE.md's bar."""
    TL.check_against_reference_run(hip, WINDOWS[k], tol_lambda=1e-3, tol_chi2=1e-4, marginal_floor=1e-6, options=dict(deterministic=1))


@pytest.mark.gpu
@pytest.mark.parametrize("k", range(len(WINDOWS)), ids=W_IDS)
@pytest.mark.parametrize("opts", [dict(deterministic=1, lm_control=1), dict(deterministic=0, lm_control=0), dict(deterministic=0, lm_control=1)],
                         ids=["deterministic-lm-on-host", "atomics-lm-on-device", "atomics-lm-on-host"])
def test_hip_other_modes_follow_the_reference_schedule(hip, k, opts):
    """The strict comparison above runs the default mode (round 6: deterministic sums, k_decide behind every trial, batches queued ahead).  The same
    windows with the LM decisions on the host, and in the floating-point-atomics mode with either control, must walk the reference's trial schedule
    (accept / reject sequence, lambda, chi2) up to the first marginal iteration, classify the same edges and end at the same cost."""
    TL.check_against_reference_run(hip, WINDOWS[k], tol_lambda=1e-3, tol_chi2=1e-4, marginal_floor=1e-6, options=opts, states=False, tol_final_chi2=3e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("case", FRAMES, ids=F_IDS)
def test_hip_follows_the_reference_pose_optimization_through_linear_solver_dense(case):
    TP.check_case(B.load_library(), "ppo_", case, 1e-7, 1e-7, False)
