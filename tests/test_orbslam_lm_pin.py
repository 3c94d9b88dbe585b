"""The reference's OWN optimisation of point + plane + cuboid windows as golden data -- linear solver included.

tests/golden/orbslam_prebuilt_lm.json was produced by oracle/pin_orbslam_lm.py: six small windows (points mono / stereo, planes
incl. vertical / parallel, cuboid bbox / corner / point-cuboid / cuboid-plane edges) taken through the 5 + classify + 10 protocol
of Optimizer.cc:2727-2838 ENTIRELY inside the reference's prebuilt lib/libORB_SLAM2.so + libg2o.so -- SparseOptimizer,
OptimizationAlgorithmLevenberg, BlockSolver<BlockSolverTraits<-1,-1>> (buildStructure, buildSystem, the Schur complement of points and
planes, back-substitution), LinearSolverDense<MatrixXd> (Eigen::LDLT + isPositive), every vertex / edge class with its numeric
Jacobians -- with a trap armed on every symbol that is not reference code (`sink_hits_during_pinned_calls` = 0).

CPU: the oracle against that run (schedule, lambda, chi2, levels, final states; the first Schur-reduced solve directly).
GPU (-m gpu): the HIP library against the same run -- the product against the reference, no oracle in between.

Tolerances.  The chi2 of the initial state agrees to 1e-12.  From the first linearisation on, every plane / cuboid block carries the
noise of the reference's central differences (delta = 1e-9: one ulp of a residual moves a Jacobian entry by ~1e-7, measured in
tests/test_orbslam_pin.py), so lambda_0 = 1e-5 max diag(H) and everything downstream agree to ~1e-7 relative rather than bit for
bit (the points-only fixture of tests/test_g2o_lm_pin.py, whose Jacobians are analytic, IS bit-identical).  Final states are held to
BASELINE.md's parity gate.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

from ppo_ba import binding as B

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "orbslam_prebuilt_lm.json")) as _fh:
    _GOLD = json.load(_fh)
CASES = _GOLD["cases"]


def problem_of(case):
    return B.Problem(**{k: np.array(v) for k, v in case["problem"].items()})


def quat_angle(a, b):
    """Angle between unit quaternions as twice the chord (arccos bottoms out at 3e-8)."""
    return 2 * min(np.linalg.norm(np.asarray(a) - b), np.linalg.norm(np.asarray(a) + b))


def is_points_only(case):
    return len(case["problem"].get("pl_plane", [])) == 0 and len(case["problem"].get("bbox_pose", [])) == 0


def first_marginal_iteration(R, floor):
    """Index of the first LM iteration of the reference run whose accepted step improved chi2 by less than `floor` (relative)."""
    first = R["iter_first_trial"] + [len(R["lambda_trials"])]
    chis = [R["chi2_at_trial_start"][first[i]] for i in range(len(first) - 1)] + [R["final_chi2"]]
    for i in range(len(chis) - 1):
        if i + 1 < len(chis) - 1 and (i + 1) == R["pass1_iterations"]:
            continue  # (the next chi2 belongs to the re-classified edge set of pass 2)
        if abs(chis[i] - chis[i + 1]) < floor * chis[i]:
            return i
    return len(chis) - 1


class ReferenceState:
    """The reference's final state under the attribute names of a result (tests/sensitivity.py: metrics)."""

    def __init__(self, fs, like):
        self.pose_qt = np.array(fs["pose_qt"], dtype=float).reshape(like.pose_qt.shape)
        self.point_xyz = np.array(fs["point_xyz"], dtype=float).reshape(like.point_xyz.shape)
        self.plane_coef = np.array(fs["plane_coef"], dtype=float).reshape(like.plane_coef.shape)
        self.cuboid_qt = np.array(fs["cuboid_qt"], dtype=float).reshape(like.cuboid_qt.shape)
        self.cuboid_scale = np.array(fs["cuboid_scale"], dtype=float).reshape(like.cuboid_scale.shape)


def check_against_reference_run(backend, case, tol_lambda, tol_chi2, gate=1.0, tol_pass1=None, marginal_floor=0.0, options=None, states=True, tol_final_chi2=1e-6):
    """tol_pass1 = (lambda, chi2) overrides the tolerances during pass 1.
    marginal_floor: an iteration of the reference whose accepted step improved chi2 by less than this (relative) was decided by the
    noise of its own central differences -- the right-hand side b of two faithful evaluations differs by 2.5e-6 of its largest
    entry (tests/test_orbslam_pin.py) -- so its accept / reject sequence, and the iterations after it, are not compared; the final
    cost and states still are."""
    R = case["result"]
    s = backend.create(problem_of(case), B.default_options(**(options or {})))
    st = s.run()
    r = s.read()
    s.close()
    it = st["iters"]
    first = R["iter_first_trial"] + [len(R["lambda_trials"])]
    n_strict = first_marginal_iteration(R, marginal_floor) if marginal_floor > 0 else len(R["iter_first_trial"])
    assert n_strict >= R["pass1_iterations"] + 3, n_strict  # (only the tail of pass 2 can be marginal)
    if n_strict == len(R["iter_first_trial"]):
        assert len(it) == len(R["iter_first_trial"]) == R["pass1_iterations"] + R["pass2_iterations"]
    else:
        assert len(it) > n_strict
    assert sum(1 for a in it if a["pass_"] == 0) == R["pass1_iterations"]
    for k, a in enumerate(it[:n_strict]):
        assert a["trials"] == first[k + 1] - first[k], (k, a["trials"], first[k + 1] - first[k])  # the same accept / reject sequence
        tl, tc = tol_pass1 if (tol_pass1 and a["pass_"] == 0) else (tol_lambda, tol_chi2)
        lam_ref = R["lambda_trials"][first[k + 1] - 1]
        assert abs(a["lambda_"] - lam_ref) <= tl * lam_ref, (k, a["lambda_"], lam_ref)
        chi_ref = R["chi2_at_trial_start"][first[k]]
        assert abs(a["chi2_before"] - chi_ref) <= (min(1e-9, tc) if k == 0 else tc) * chi_ref, (k, a["chi2_before"], chi_ref)
    assert abs(it[-1]["chi2_after"] - R["final_chi2"]) <= tol_final_chi2 * R["final_chi2"]
    lv = R["levels_after_pass1"]
    for name, got in (("mono", r.mono_level), ("stereo", r.stereo_level), ("plane", r.pl_level), ("bbox", r.bbox_level), ("corner", r.corner_level),
                      ("cub3d", r.cub3d_level)):
        assert np.array_equal(got, np.array(lv.get(name, []), dtype=np.uint8)), name
    if not states:  # (the caller holds the final state and the per-edge chi2 to its own rule)
        return st, r
    fs = R["final_state"]
    ref_pose, ref_pt = np.array(fs["pose_qt"]), np.array(fs["point_xyz"])
    # BASELINE.md section 2, parity gate (translation 1e-5 m, rotation 1e-6 rad, points 1e-5 m, planes 1e-6, cuboids 1e-5).
    # gate=None: the caller holds the STATE to its own rule (sensitivity.BARS + the oracle-spread excuse); the per-edge chi2 below still is checked here.
    state_gate = gate is not None
    gate = 1.0 if gate is None else gate
    if state_gate:
        assert np.sqrt(np.mean(np.sum((r.pose_qt[:, 4:] - ref_pose[:, 4:]) ** 2, axis=1))) < 1e-5 * gate
        assert max(quat_angle(a[:4], b[:4]) for a, b in zip(r.pose_qt, ref_pose)) < 1e-6 * gate
    # points: where pass 2 determines them (two active observations: >= 4 residual rows for 3 unknowns)
    P = case["problem"]
    n_mono = len(P.get("mono_point", []))
    E = np.concatenate([np.array(P.get("mono_point", []), dtype=int).reshape(-1), np.array(P.get("stereo_point", []), dtype=int).reshape(-1)])
    rows = np.concatenate([np.full(n_mono, 2), np.full(len(E) - n_mono, 3)])
    lvl = np.concatenate([np.array(lv["mono"], dtype=int), np.array(lv["stereo"], dtype=int)])
    nrows = np.zeros(len(ref_pt))
    np.add.at(nrows, E[lvl == 0], rows[lvl == 0])
    well = nrows >= 4
    assert well.sum() >= 8
    if state_gate:
        assert np.sqrt(np.mean(np.sum((r.point_xyz[well] - ref_pt[well]) ** 2, axis=1))) < 1e-5 * gate
        if len(fs["plane_coef"]):
            assert np.abs(r.plane_coef - np.array(fs["plane_coef"])).max() < 1e-6 * gate
        if len(fs["cuboid_qt"]):
            assert np.abs(r.cuboid_qt[:, 4:] - np.array(fs["cuboid_qt"])[:, 4:]).max() < 1e-5 * gate
            assert max(quat_angle(a[:4], b[:4]) for a, b in zip(r.cuboid_qt, np.array(fs["cuboid_qt"]))) < 1e-5 * gate
            assert np.abs(r.cuboid_scale - np.array(fs["cuboid_scale"])).max() < 1e-5 * gate
    # per-edge chi2 of the custom edges at the final state (active edges).  Bar: a bbox residual of ~10 px with Omega = 0.49 turns a
    # state deviation of 1e-5 m (the gate above; f / z ~ 160 px per metre) into 2 e Omega de ~ 1.6e-2 of chi2 -- 1e-3 of the
    # largest chi2 (~50) is the tighter of the two statements
    for name, got in (("plane", r.pl_chi2), ("bbox", r.bbox_chi2), ("corner", r.corner_chi2), ("ptcub", r.ptcub_chi2), ("cubpl", r.cubpl_chi2),
                      ("cub3d", r.cub3d_chi2)):
        ref = np.array(R["final_edge_chi2"].get(name, []))
        if ref.size == 0:
            continue
        act = np.array(lv[name], dtype=int) == 0
        if not act.any():
            continue  # (every edge of the class was culled between the passes)
        assert np.abs(got[act] - ref[act]).max() <= 1e-3 * max(1.0, np.abs(ref[act]).max()), name
    return st, r


def test_fixture_is_the_reference_and_exercises_the_schedule():
    assert _GOLD["trap"]["sink_hits_during_pinned_calls"] == 0
    trials = [len(c["result"]["lambda_trials"]) - len(c["result"]["iter_first_trial"]) for c in CASES]
    assert sum(t > 0 for t in trials) >= 3 and max(trials) >= 8            # rejected trials
    assert any(c["result"]["pass2_iterations"] < 10 for c in CASES)         # the nBad stop
    kinds = set()
    for c in CASES:
        lv = c["result"]["levels_after_pass1"]
        kinds |= {k for k, v in lv.items() if len(v)}
        assert all(c["result"]["solve_ok"])
    assert kinds == {"plane", "mono", "stereo", "bbox", "corner", "ptcub", "cubpl", "cub3d"}
    assert sum(1 for c in CASES if len(c["problem"].get("cub3d_pose", []))) >= 2  # Optimizer::LocalBACameraPointCuboids2D windows (9-D edge)
    assert sum(1 for c in CASES if is_points_only(c)) >= 2
    assert any(sum(c["result"]["levels_after_pass1"]["plane"]) > 0 for c in CASES) or any(sum(c["result"]["levels_after_pass1"]["bbox"]) > 0 for c in CASES)
    assert any(1 in c["problem"].get("pl_kind", []) for c in CASES) and any(2 in c["problem"].get("pl_kind", []) for c in CASES)


@pytest.mark.parametrize("k", range(len(CASES)))
def test_oracle_follows_the_reference_run(oracle, k):
    """Windows with plane / cuboid edges: at the differencing noise.  Points-only windows (analytic Jacobians): lambda of every pass-1
    iteration BIT-IDENTICAL (asserted to 1e-12) and chi2 to 1e-12 with the reference's own BlockSolverX + LinearSolverDense doing
    the linear algebra -- what tests/test_g2o_lm_pin.py could only show with a numpy solver standing in."""
    if is_points_only(CASES[k]):
        check_against_reference_run(oracle, CASES[k], tol_lambda=1e-5, tol_chi2=1e-9, tol_pass1=(1e-12, 1e-12))
    else:
        check_against_reference_run(oracle, CASES[k], tol_lambda=2e-4, tol_chi2=2e-5)


@pytest.mark.parametrize("k", range(len(CASES)))
def test_oracle_first_reduced_solve_matches_block_solver_and_ldlt(oracle, oracle_lib, k):
    """BlockSolverX::solve of the very first trial: right-hand side b as buildSystem gathered it, and the pose / cuboid part of the
    solution x -- the Schur complement over points and planes (block_solver.hpp:354-460) followed by LinearSolverDense
    (linear_solver_dense.h:65-113) -- against the oracle's Schur complement + restated LDLT on the same initial state."""
    case = CASES[k]
    sysm = case["result"]["first_system"][0]
    assert sysm["trial"] == 0 and sysm["ok"]
    lib = oracle_lib
    dp = C.POINTER(C.c_double)
    s = oracle.create(problem_of(case), B.default_options())
    chi = C.c_double()
    lib.oracle_ba_linearize(s.h, C.byref(chi))
    assert abs(chi.value - case["result"]["chi2_at_trial_start"][0]) <= 1e-12 * chi.value
    n = C.c_int32()
    lib.oracle_ba_reduced(s.h, sysm["lambda"], 0, -1, 1, None, None, C.byref(n))
    n = n.value
    S, bs, x = np.zeros((n, n)), np.zeros(n), np.zeros(n)
    lib.oracle_ba_reduced(s.h, sysm["lambda"], 0, -1, 1, S.ctypes.data_as(dp), bs.ctypes.data_as(dp), None)
    assert lib.oracle_ldlt_solve(S.ctypes.data_as(dp), n, bs.ctypes.data_as(dp), x.ctypes.data_as(dp)) == 1
    x_ref = np.array(sysm["x"])[:n]
    # (measured 2e-9 ... 3e-5 of the largest component: the reference's central differences put ~1e-6 relative noise on every plane /
    #  cuboid block of H, and the solve amplifies it by the conditioning of the damped system -- lambda_0 is only 1e-5 max diag H)
    dev = np.abs(x - x_ref).max() / np.abs(x_ref).max()
    print("first reduced solve: n = %d, max |x - x_ref| / max |x_ref| = %.2e" % (n, dev))
    points_only = is_points_only(case)
    assert dev <= (1e-10 if points_only else 1e-3), dev
    # b of the reduced unknowns (poses, then cuboids, in index order)
    b_ref = np.array(sysm["b"])
    P = case["problem"]
    off, worst = 0, 0.0
    for vt, count, d in ((0, len(P["pose_qt"]), 6), (3, len(P.get("cuboid_qt", [])), 9)):
        for i in range(count):
            bv = np.zeros(9)
            if lib.oracle_ba_block(s.h, vt, i, -1, 0, None, bv.ctypes.data_as(dp)) > 0:
                worst = max(worst, np.abs(bv[:d] - b_ref[off:off + d]).max() / max(1.0, np.abs(b_ref[off:off + d]).max()))
                off += d
    assert off == n and worst <= (1e-12 if points_only else 1e-5), (off, n, worst)
    s.close()


def _oracle_spread(oracle, P, like):
    import sensitivity as S
    runs = S.oracle_runs(oracle, P, B.default_options(), variants=S.VARIANTS)
    return S.spread(runs, S.determined_points(P, like))[0], runs


@pytest.mark.gpu
@pytest.mark.parametrize("k", range(len(CASES)))
def test_hip_follows_the_reference_run(oracle, hip, k):
    """The product against the reference's own run of the complete path (no oracle in between), in its default mode -- since round 6 the
    DETERMINISTIC one (options.deterministic = 1: bit-reproducible, so this comparison has no run-to-run component; the mode bench.py times;
    the LM decisions are taken on the device, batches of solve attempts queued ahead): schedule, lambda and chi2 per
    iteration up to the first marginal one, levels, final chi2 -- and the final state at BASELINE.md's bar, quantity by quantity.  The only
    excuse above the bar is the reference algorithm's own sensitivity on that window: the spread of the oracle's four rounding-only
    re-runs must reach 0.3 of the deviation (tools/lm_pin_spread.py, MI355X: one of the eleven windows uses it -- the cuboid of window 3,
    1.2e-5 against a spread of 1.5e-5; everything else is inside the bar, the largest ratio 0.74 on a rotation)."""
    import sensitivity as S
    case = CASES[k]
    _, r = check_against_reference_run(hip, case, tol_lambda=1e-3, tol_chi2=1e-4, marginal_floor=1e-6, options=dict(deterministic=1), gate=None)
    P = problem_of(case)
    ref = ReferenceState(case["result"]["final_state"], r)
    m = S.metrics(ref, r, S.determined_points(P, r))
    beyond = [q for q, bar in S.BARS.items() if m[q] >= bar]
    if beyond:
        oo, _ = _oracle_spread(oracle, P, r)
        for q in beyond:
            assert oo[q] >= 0.3 * m[q], (k, q, m[q], oo[q])
    print("window %d, deterministic mode against the reference's final state: %s" % (k, {q: "%.1e" % v for q, v in m.items()}))


@pytest.mark.gpu
@pytest.mark.parametrize("k", range(len(CASES)))
def test_hip_atomics_mode_follows_the_reference_run(oracle, hip, k):
    """The same windows in the opt-in floating-point-atomics mode (options.deterministic = 0: the arrival order of the addends of a cross-wave sum
    changes from run to run).  Schedule, lambda, chi2 per iteration up to the first marginal one, levels and the final chi2 (1e-6) against the reference as
    above.  Final state: against the DETERMINISTIC result of the same library, per quantity inside max(3 x the bar, 5 x the oracle's own
    spread on the window) -- the rule of test_atomics_mode_stays_near_the_deterministic_result.  These windows are small (6 keyframes,
    40 points) and two of them end on a plateau: in about one run in ten the default mode takes one more or one fewer marginal iteration
    there than the deterministic run and stops elsewhere on the flat stretch (tools/lm_pin_spread.py: window 1, points 1.3e-4 apart at an
    identical cost).  Then -- and only when the two schedules do differ -- the state is held to the DIRECT check instead: the oracle's
    objective evaluated at it equals the oracle's own minimum to 1e-6, and nothing is further than 30 x the bar from the deterministic
    result."""
    import sensitivity as S
    case = CASES[k]
    P = problem_of(case)
    # (final chi2: 1e-6 in the deterministic test above; 1.01e-6 was seen once in 20 default-mode runs of window 4)
    st, r = check_against_reference_run(hip, case, tol_lambda=1e-3, tol_chi2=1e-4, marginal_floor=1e-6, states=False, tol_final_chi2=3e-6, options=dict(deterministic=0))
    sd = hip.create(P, B.default_options(deterministic=1))
    td = sd.run()
    rd = sd.read()
    sd.close()
    m = S.metrics(rd, r, S.determined_points(P, rd) & S.determined_points(P, r))
    beyond = [q for q, bar in S.BARS.items() if m[q] >= 3 * bar]
    if beyond:
        oo, runs = _oracle_spread(oracle, P, rd)
        if S.schedule(td) == S.schedule(st):
            for q in beyond:
                assert m[q] < 5 * oo[q], (k, q, m[q], oo[q])
        else:
            so = oracle.create(P, B.default_options())
            so.run()
            ro = so.read()
            gap, _, _ = S.optimality_gap(oracle, so, ro, r)
            so.close()
            assert gap <= 1e-6, (k, gap)
            for q in beyond:
                assert m[q] < max(30 * S.BARS[q], 5 * oo[q]), (k, q, m[q], oo[q])
    R = case["result"]
    for name, got in (("plane", r.pl_chi2), ("bbox", r.bbox_chi2), ("corner", r.corner_chi2), ("ptcub", r.ptcub_chi2), ("cubpl", r.cubpl_chi2),
                      ("cub3d", r.cub3d_chi2)):
        ref = np.array(R["final_edge_chi2"].get(name, []))
        act = np.array(R["levels_after_pass1"].get(name, []), dtype=int) == 0
        if ref.size and act.any() and not beyond:
            assert np.abs(got[act] - ref[act]).max() <= 1e-3 * max(1.0, np.abs(ref[act]).max()), name
