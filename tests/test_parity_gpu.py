"""Parity of the HIP path (through the C ABI of libppo_ba.so) against the CPU oracle on identical graphs.

Tolerances: pose translation RMSE < 1e-5 m, rotation < 1e-6 rad,
point RMSE < 1e-5 m, plane coefficients max-abs < 1e-6, cuboid 9-vector max-abs < 1e-5, final chi2 relative
< 1e-6, identical inlier/outlier classification except edges within 1e-6 relative of a threshold.
Floating-point caveat written into the tests: the plane / cuboid edges are differentiated numerically with
delta = 1e-9 on both sides, so their Jacobians carry ~1e-7 relative round-off noise that depends on libm and on
summation order; system-level comparisons therefore use 1e-5 relative where those edges contribute.
"""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import dptr, rot_angle
from ppo_ba import binding as B
from ppo_ba import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip_hooks():
    """libppo_ba_test.so: the same sources built with -DPPO_BA_TEST_HOOKS (csrc/ppo_dev.hpp) -- the A/B environment switches, the
    in-process loopback communicator and the one-rank RCCL self-test exist only there."""
    return B.hip_backend(test_hooks=True)


def both(oracle, hip, P, hip_only=None, **opt):
    """The same options on both sides; `hip_only` options (jacobian_mode: the oracle IS the numeric reference) go to the device only."""
    o = B.default_options(**opt)
    return oracle.create(P, o), hip.create(P, B.default_options(**dict(opt, **(hip_only or {}))))


def compare_states(ro, rh, tol_scale=1.0):
    n = ro.pose_qt.shape[0]
    t_rmse = np.sqrt(np.mean(np.sum((ro.pose_qt[:, 4:] - rh.pose_qt[:, 4:]) ** 2, axis=1)))
    r_max = max(rot_angle(ro.pose_qt[i, :4], rh.pose_qt[i, :4]) for i in range(n))
    assert t_rmse < 1e-5 * tol_scale, t_rmse
    assert r_max < 1e-6 * tol_scale, r_max
    if ro.point_xyz.size:
        p_rmse = np.sqrt(np.mean(np.sum((ro.point_xyz - rh.point_xyz) ** 2, axis=1)))
        assert p_rmse < 1e-5 * tol_scale, p_rmse
    if ro.plane_coef.size:
        assert np.abs(ro.plane_coef - rh.plane_coef).max() < 1e-6 * tol_scale
    if ro.cuboid_qt.size:
        assert np.abs(ro.cuboid_qt - rh.cuboid_qt).max() < 1e-5 * tol_scale
        assert np.abs(ro.cuboid_scale - rh.cuboid_scale).max() < 1e-5 * tol_scale


def S_schedule(st):
    return [(a["pass_"], a["iteration"], a["trials"], a["accepted"], a["result"]) for a in st["iters"]]


def compare_levels(lo, lh, chi, th):
    diff = np.nonzero(lo != lh)[0]
    for i in diff:  # only edges sitting on the threshold may flip
        assert abs(chi[i] - th) <= 1e-6 * th, (i, chi[i], th)


CASES = [
    ("tiny", dict(seed=1)),
    ("small", dict(seed=1)),
    ("small", dict(seed=2, stereo_frac=0.3)),
    ("small", dict(seed=3, with_corners=True)),
    ("small", dict(seed=4, with_vp_planes=2)),
    ("small", dict(seed=5, fixed_observer_frac=0.3)),
    (dict(n_kf=12, n_points=400, n_planes=0, n_cuboids=0), dict(seed=6)),  # points only (config 2 shape)
    (dict(n_kf=12, n_points=400, n_planes=6, n_cuboids=0), dict(seed=7)),  # + planes (config 3 shape)
    # the 9-D camera-object edge EdgeSE3Cuboid of Optimizer::LocalBACameraPointCuboids2D (Optimizer.cc:1767-1800)
    ("small", dict(seed=9, with_cuboid3d=True)),
    (dict(n_kf=12, n_points=300, n_planes=0, n_cuboids=3), dict(seed=10, with_cuboid3d=True, with_corners=True)),  # points + cuboids only
]


@pytest.mark.parametrize("elim", [1, 0])
@pytest.mark.parametrize("cfg,kw", CASES)
def test_linearisation_matches_oracle(oracle, hip, cfg, kw, elim):
    """Robust chi2 and the reduced (Schur) system of the first LM step.  With eliminate_cuboids the device system is
    poses only: the oracle's (pose + cuboid) system is reduced the same way in numpy before comparing."""
    P = synth.make_problem(cfg, **kw)
    so, sh = both(oracle, hip, P, eliminate_cuboids=elim)
    chi_o, chi_h = so.linearize(), sh.linearize()
    assert abs(chi_o - chi_h) <= 1e-9 * max(1.0, abs(chi_o))
    lam = 7.5
    n = C.c_int32()
    oracle.lib.oracle_ba_reduced(so.h, lam, 0, -1, 1, None, None, C.byref(n))
    n = n.value
    So, bo = np.zeros((n, n)), np.zeros(n)
    oracle.lib.oracle_ba_reduced(so.h, lam, 0, -1, 1, dptr(So), dptr(bo), None)
    nh = C.c_int32()
    hip.lib.ppo_ba_debug_set_lambda(sh.h, C.c_double(lam))
    hip.lib.ppo_ba_debug_reduced(sh.h, None, None, None, C.byref(nh))
    nh = nh.value
    Sh, bh, xh = np.zeros((nh, nh)), np.zeros(nh), np.zeros(nh)
    assert hip.lib.ppo_ba_debug_reduced(sh.h, dptr(Sh), dptr(bh), dptr(xh), None) == 0
    # both libraries order the reduced unknowns alike: free poses in input order, then cuboids
    if nh < n:
        assert elim == 1 and (n - nh) % 9 == 0
        Spc, Scc = So[:nh, nh:], So[nh:, nh:]
        K = np.linalg.solve(Scc, Spc.T).T
        So, bo = So[:nh, :nh] - K @ Spc.T, bo[:nh] - K @ bo[nh:]
    else:
        assert nh == n
    scale = np.abs(So).max()
    assert np.abs(So - Sh).max() <= 2e-6 * scale, np.abs(So - Sh).max() / scale
    assert np.abs(bo - bh).max() <= 2e-6 * np.abs(bo).max()
    # the device solve satisfies its own system
    assert np.linalg.norm(Sh @ xh - bh) <= 1e-8 * np.linalg.norm(bh)


# options.lm_control: the LM decisions taken by a kernel behind every trial (default wherever the configuration allows) or by the host;
# options.deterministic: 1 (default since round 6) fixed-point integer sums, bit-reproducible -- the mode bench.py times; 0 floating-point atomics
LM_CONTROL = pytest.mark.parametrize("lm,det", [(0, 1), (1, 1), (0, 0), (1, 0)], ids=["lm-on-device", "lm-on-host", "atomics-lm-on-device", "atomics-lm-on-host"])
ATOMICS = dict(deterministic=0)  # the opt-in floating-point-atomics mode


@LM_CONTROL
@pytest.mark.parametrize("elim", [1, 0])
@pytest.mark.parametrize("cfg,kw", CASES)
def test_full_protocol_matches_oracle(oracle, hip, cfg, kw, elim, lm, det):
    """5 LM iterations, outlier classification, 10 more: traces, final estimates, per-edge chi2, levels."""
    P = synth.make_problem(cfg, **kw)
    so, sh = both(oracle, hip, P, eliminate_cuboids=elim, lm_control=lm, deterministic=det)
    to, th = so.run(), sh.run()
    assert to["n_reduced"] == th["n_reduced"]
    # The accept / reject sequence is compared up to the first iteration of the ORACLE's own trace that is converged to rounding: a step that moved chi2
    # by less than 1e-9 relative was accepted or rejected by the sign of a difference of two numbers equal to nine digits (e.g. the terminating
    # iteration of pass 2 of the "tiny" window: the oracle accepts its only attempt and stops on nBad = 3, the deterministic device run -- bit-
    # reproducible, so this is a property of its summation order, not of a run -- finds rho < 0 by an ulp, retries twice and stops on rho == 0).
    # From there on: the cost reached and the states, below.
    k_strict = next((k for k, a in enumerate(to["iters"]) if abs(a["chi2_before"] - a["chi2_after"]) < 1e-9 * abs(a["chi2_before"])), len(to["iters"]))
    assert k_strict >= sum(1 for a in to["iters"] if a["pass_"] == 0) + 3 or k_strict == len(to["iters"]), k_strict  # (only the tail of pass 2 can be that flat)
    if k_strict == len(to["iters"]):
        assert len(to["iters"]) == len(th["iters"])
    else:
        assert len(th["iters"]) > k_strict and abs(len(to["iters"]) - len(th["iters"])) <= 2
    worst = 0.0
    for a, b in list(zip(to["iters"], th["iters"]))[:k_strict]:
        assert (a["pass_"], a["iteration"], a["trials"], a["accepted"], a["result"]) == (b["pass_"], b["iteration"], b["trials"], b["accepted"], b["result"])
        worst = max(worst, abs(a["chi2_before"] - b["chi2_before"]) / abs(a["chi2_before"]), abs(a["chi2_after"] - b["chi2_after"]) / abs(a["chi2_after"]))
        assert abs(a["lambda_"] - b["lambda_"]) <= 1e-3 * abs(a["lambda_"])  # lambda is a cubic function of the gain ratio
    assert abs(to["iters"][-1]["chi2_after"] - th["iters"][-1]["chi2_after"]) <= 1e-6 * abs(to["iters"][-1]["chi2_after"])
    # BASELINE.md's bar: 1e-6 per iteration (measured: worst 3.0e-7 over 320 runs, tools/r3_diag.py; 3.2e-7 over 6400 in round 1, one
    # excursion above 1e-6 in ~100 suite runs).  Above the bar only where the ORACLE's own trace moves as much when nothing but its
    # rounding changes (the 1e-7 noise of the delta = 1e-9 Jacobians, amplified over the iterations).
    if worst > 1e-6:
        import sensitivity as S
        own = 0.0
        for st2, _ in S.oracle_runs(oracle, P, B.default_options(eliminate_cuboids=elim), variants=S.VARIANTS)[1:]:
            for a, b in zip(to["iters"], st2["iters"]):
                own = max(own, abs(a["chi2_after"] - b["chi2_after"]) / abs(a["chi2_after"]))
        assert worst <= 5 * own, (worst, own)
    ro, rh = so.read(), sh.read()
    if k_strict == len(to["iters"]) or S_schedule(to) == S_schedule(th):
        compare_states(ro, rh)
    else:
        # The two runs stand on different sides of a step whose acceptance was decided by rounding (it moved chi2 by less than 1e-9 relative -- and, on the
        # weakly held "tiny" window, the plane coefficients by 1.1e-6): the device must stand where the oracle stood AFTER that step or BEFORE it (the
        # oracle stopped one iteration earlier), to the same bars.
        try:
            compare_states(ro, rh)
        except AssertionError:
            n1 = sum(1 for a in to["iters"] if a["pass_"] == 0)
            sb = oracle.create(P, B.default_options(eliminate_cuboids=elim, iters_pass2=k_strict - n1))
            sb.run()
            ro = sb.read()
            sb.close()
            compare_states(ro, rh)
    o = B.default_options()
    compare_levels(ro.mono_level, rh.mono_level, ro.mono_chi2, o.chi2_mono)
    compare_levels(ro.stereo_level, rh.stereo_level, ro.stereo_chi2, o.chi2_stereo)
    compare_levels(ro.pl_level, rh.pl_level, ro.pl_chi2, o.chi2_plane)
    assert np.array_equal(ro.bbox_level, rh.bbox_level)
    assert np.array_equal(ro.cub3d_level, rh.cub3d_level)
    np.testing.assert_array_equal(ro.mono_depth_positive, rh.mono_depth_positive)
    for name in ("mono_chi2", "stereo_chi2", "pl_chi2", "bbox_chi2", "corner_chi2", "ptcub_chi2", "cubpl_chi2", "cub3d_chi2"):
        a, b = getattr(ro, name), getattr(rh, name)
        if a.size:
            # cuboid-derived residuals inherit the 1e-5 tolerance of the cuboid state they are evaluated at; the plane
            # residual chi2 = 3282.8 * angle^2 + 1e4 * dist^2 magnifies the 1e-6 tolerance of the plane coefficients
            tol = 1e-5 if name in ("bbox_chi2", "corner_chi2", "ptcub_chi2", "pl_chi2", "cub3d_chi2") else 1e-6
            assert np.abs(a - b).max() <= tol * max(1.0, np.abs(a).max()), name


@pytest.mark.parametrize("cfg,kw", [("small", dict(seed=1)), (dict(n_kf=40, n_points=2000, n_planes=10, n_cuboids=5), dict(seed=2)),
                                    (dict(n_kf=33, n_points=1500, n_planes=0, n_cuboids=0), dict(seed=3))])
def test_native_cholesky_matches_rocsolver_and_oracle_ldlt(oracle_lib, hip, cfg, kw):
    """The reduced solve: native blocked Cholesky (solver=2) vs rocSOLVER potrf/potrs (solver=1) vs the oracle's
    restated pivoted LDLT on the same matrix.  n is deliberately not a multiple of the 64-wide panel."""
    P = synth.make_problem(cfg, **kw)
    xs = {}
    for kind in (1, 2):
        s = hip.create(P, B.default_options(solver=kind))
        n = C.c_int32()
        hip.lib.ppo_ba_debug_set_lambda(s.h, C.c_double(3.0))
        hip.lib.ppo_ba_debug_reduced(s.h, None, None, None, C.byref(n))
        n = n.value
        S, b, x = np.zeros((n, n)), np.zeros(n), np.zeros(n)
        assert hip.lib.ppo_ba_debug_reduced(s.h, dptr(S), dptr(b), dptr(x), None) == 0
        assert np.linalg.norm(S @ x - b) <= 1e-9 * np.linalg.norm(b)
        xs[kind] = (S, b, x)
        s.close()
    assert np.abs(xs[1][2] - xs[2][2]).max() <= 1e-8 * np.abs(xs[1][2]).max()
    S, b, x = xs[2]
    xo = np.zeros_like(x)
    assert oracle_lib.oracle_ldlt_solve(dptr(np.ascontiguousarray(S)), len(b), dptr(b), dptr(xo)) == 1
    assert np.abs(xo - x).max() <= 1e-8 * np.abs(xo).max()


@pytest.mark.parametrize("det", [1, 0], ids=["deterministic", "atomics"])
@pytest.mark.parametrize("n_kf", [3, 4, 6, 9, 12, 15, 17, 23, 30, 33])
def test_one_launch_solve_of_small_systems_against_the_panel_path(hip_hooks, monkeypatch, n_kf, det):
    """Reduced systems of at most 192 unknowns can be solved by ONE launch with the matrix in one compute unit (csrc/ppo_chol_small.inc; the product uses
    it up to 96 unknowns, where it is the faster path).  The same reduced system through that kernel (PPO_BA_CHOL_SMALL=192: wherever it can run) and
    through the panel path (=0), both against numpy on the matrix the library hands out; then a complete 5 + 10 run on either path."""
    P = synth.make_problem(dict(n_kf=n_kf, n_points=60 * n_kf, n_planes=min(4, n_kf), n_cuboids=2 if n_kf > 3 else 0), seed=20 + n_kf)
    xs, runs = {}, {}
    for mode in ("192", "0"):
        monkeypatch.setenv("PPO_BA_CHOL_SMALL", mode)
        monkeypatch.setenv("PPO_BA_DEBUG_OVERLAP", "1")  # (x of ppo_ba_debug_reduced through the path a run takes: in deterministic mode the kernel converts on load)
        s = hip_hooks.create(P, B.default_options(solver=2, deterministic=det))
        n = C.c_int32()
        hip_hooks.lib.ppo_ba_debug_set_lambda(s.h, C.c_double(2.0))
        hip_hooks.lib.ppo_ba_debug_reduced(s.h, None, None, None, C.byref(n))
        n = n.value
        assert n == 6 * (n_kf - 1)
        S, b, x = np.zeros((n, n)), np.zeros(n), np.zeros(n)
        assert hip_hooks.lib.ppo_ba_debug_reduced(s.h, dptr(S), dptr(b), dptr(x), None) == 0
        assert np.all(np.isfinite(x)) and np.linalg.norm(S @ x - b) <= 1e-9 * np.linalg.norm(b)
        assert np.abs(x - np.linalg.solve(S, b)).max() <= 1e-9 * np.abs(x).max()
        xs[mode] = x
        s.reset()
        runs[mode] = (s.run(), s.read())
        s.close()
    # (deterministic mode: on the panel path the trailing updates of column blocks that are still open go to the fixed-point array -- rounded to its 2^-24 grid like
    #  every other term there, csrc/ppo_chol.hip: StepFix -- which the one-launch kernel, with nothing open, does not do: measured 1.2e-10 / 2.0e-10 at 174 / 192 unknowns)
    assert np.abs(xs["192"] - xs["0"]).max() <= (1e-9 if det else 1e-10) * np.abs(xs["0"]).max()
    (ta, ra), (tb, rb) = runs["192"], runs["0"]
    assert [i["trials"] for i in ta["iters"]] == [i["trials"] for i in tb["iters"]]
    assert abs(ta["iters"][-1]["chi2_after"] - tb["iters"][-1]["chi2_after"]) <= 1e-6 * tb["iters"][-1]["chi2_after"]
    compare_states(rb, ra)


def test_warmup_entry_point(hip):
    """ppo_ba_warmup (include/ppo_ba.h; INTEGRATION.md: one line in System's constructor): succeeds, may be called again, and a create right after it
    costs what a steady-state create does (the first-use costs -- module load, stream, mapped block, host team -- are not left for it)."""
    import time
    hip.lib.ppo_ba_warmup.restype = C.c_int
    hip.lib.ppo_ba_warmup.argtypes = [C.c_int]
    assert hip.lib.ppo_ba_warmup(0) == 0
    t0 = time.perf_counter()
    assert hip.lib.ppo_ba_warmup(0) == 0
    assert time.perf_counter() - t0 < 0.1  # (wall clock on a shared box: generous -- the first call pays ~150 ms)
    P = synth.make_problem("small", seed=1)
    t0 = time.perf_counter()
    s = hip.create(P, B.default_options())
    dt = time.perf_counter() - t0
    s.close()
    assert dt < 0.05, dt  # (0.4 ms typical; 155 ms is what a create pays without the warm-up -- the bound only has to tell these apart)
    # the warm-up reserves a slab for the device block cache (csrc/ppo_pool.hpp): blocks carved from it live in the same free lists; giving the cache
    # back while a solver holds such blocks, and again when nobody does, must leave a working library behind
    a = hip.create(P, B.default_options())
    assert hip.lib.ppo_ba_release_cached() == 0
    ta = a.run()
    a.close()
    assert hip.lib.ppo_ba_release_cached() == 0
    b = hip.create(P, B.default_options())
    tb = b.run()
    b.close()
    assert [i["trials"] for i in ta["iters"]] == [i["trials"] for i in tb["iters"]]
    assert hip.lib.ppo_ba_warmup(0) == 0  # (reserves again)
    c = hip.create(synth.make_problem(dict(n_kf=40, n_points=3000, n_planes=10, n_cuboids=4), seed=3), B.default_options())
    assert c.run()["n_iters"] > 0
    c.close()


def test_one_launch_solve_reports_a_matrix_that_is_not_positive(hip_hooks, monkeypatch):
    """levenberg.cpp:126-127 treats a failed LDLT as a rejected trial; the one-launch kernel must raise the same flag as the panel path: the
    indefinite window of the failed-factorisation tests below walks the same schedule on either path."""
    P, kw = _not_positive_window()
    stats = {}
    for mode in ("192", "0"):
        monkeypatch.setenv("PPO_BA_CHOL_SMALL", mode)
        s = hip_hooks.create(P, B.default_options(solver=2, **kw))
        stats[mode] = s.run()
        s.close()
    sched = lambda t: [(i["trials"], i["accepted"], i["result"]) for i in t["iters"]]
    assert sched(stats["192"]) == sched(stats["0"]) and sched(stats["0"])[:2] == [(1, 1, 0), (2, 0, 1)]


def test_native_cholesky_beyond_the_resident_chain(hip):
    """A window with more 64-column blocks (34) than workgroups of the backward substitution fit on the one XCD they all run on (32): the chain
    must make progress with part of it resident (its head is the first workgroup dispatched) and its ring of x blocks must wrap.  Native
    solver against rocSOLVER on the same reduced system, and the residual of both."""
    P = synth.make_problem(dict(n_kf=360, n_points=6000, n_planes=20, n_cuboids=4), seed=11)
    xs = {}
    for kind in (1, 2):
        s = hip.create(P, B.default_options(solver=kind))
        n = C.c_int32()
        hip.lib.ppo_ba_debug_set_lambda(s.h, C.c_double(3.0))
        hip.lib.ppo_ba_debug_reduced(s.h, None, None, None, C.byref(n))
        n = n.value
        assert n > 32 * 64
        S, b, x = np.zeros((n, n)), np.zeros(n), np.zeros(n)
        assert hip.lib.ppo_ba_debug_reduced(s.h, dptr(S), dptr(b), dptr(x), None) == 0
        assert np.all(np.isfinite(x)) and np.linalg.norm(S @ x - b) <= 1e-9 * np.linalg.norm(b)
        xs[kind] = x
        s.close()
    assert np.abs(xs[1] - xs[2]).max() <= 1e-8 * np.abs(xs[1]).max()
    # ... and a complete run through the default path (overlapped reduction, 34 panel launches)
    s = hip.create(P, B.default_options())
    st = s.run()
    r = s.read()
    s.close()
    assert st["n_iters"] > 0 and np.all(np.isfinite(r.pose_qt)) and st["iters"][-1]["chi2_after"] < st["iters"][0]["chi2_before"]


def test_fixed_points_and_all_but_one_pose_fixed(oracle, hip):
    P = synth.make_problem("small", seed=8)
    pf = np.zeros(P.sizes()["n_points"], np.uint8)
    pf[::3] = 1  # fixPoint for a third of the points (Optimizer.cc:2344-2345)
    P.set("point_fixed", pf)
    so, sh = both(oracle, hip, P)
    to, th = so.run(), sh.run()
    assert len(to["iters"]) == len(th["iters"])
    ro, rh = so.read(), sh.read()
    compare_states(ro, rh)
    assert np.array_equal(rh.point_xyz[::3], P["point_xyz"].reshape(-1, 3)[::3])  # fixed points never move


def test_zero_noise_graph_stays_put(hip):
    P = synth.make_problem("small", seed=9, noise_scale=0.0, outlier_frac=0.0)
    s = hip.create(P, B.default_options())
    st = s.run()
    r = s.read()
    tr = P.meta["truth"]
    assert np.abs(r.point_xyz - tr["points"]).max() < 1e-4  # inputs were rounded through float32
    assert st["iters"][0]["chi2_before"] < 1.0


@LM_CONTROL
def test_stop_flag_aborts_before_and_between(hip, lm, det):
    P = synth.make_problem("small", seed=10)
    s = hip.create(P, B.default_options(lm_control=lm, deterministic=det))
    flag = np.ones(1, np.uint8)
    st = s.run(stop_flag=flag)
    assert s.last_rc == 1 and st["n_iters"] == 0
    r = s.read()  # nothing moved
    np.testing.assert_allclose(r.point_xyz, P["point_xyz"].reshape(-1, 3), atol=0)


def test_reset_makes_runs_repeatable(hip):
    P = synth.make_problem("small", seed=11)
    s = hip.create(P, B.default_options())
    a = s.run(); ra = s.read()
    s.reset()
    b = s.run(); rb = s.read()
    assert len(a["iters"]) == len(b["iters"])
    # atomics make the summation order free: equal to round-off, not bitwise
    # (up to 2.3e-7 on the poses over 1500 repeats, tools/stress_repeat.py; chi2 repeats to 1e-9)
    np.testing.assert_allclose(ra.pose_qt, rb.pose_qt, atol=2e-6)
    np.testing.assert_allclose(ra.point_xyz, rb.point_xyz, atol=1e-5)  # two-view points amplify round-off along the ray
    assert abs(a["iters"][-1]["chi2_after"] - b["iters"][-1]["chi2_after"]) <= 1e-7 * abs(a["iters"][-1]["chi2_after"])


def test_zero_iterations_and_empty_blocks(hip):
    P = synth.make_problem(dict(n_kf=6, n_points=60, n_planes=0, n_cuboids=0), seed=12)
    s = hip.create(P, B.default_options(iters_pass1=0, iters_pass2=0))
    st = s.run()
    assert st["n_iters"] == 0
    r = s.read()
    np.testing.assert_allclose(r.point_xyz, P["point_xyz"].reshape(-1, 3), atol=0)


@pytest.mark.parametrize("cfg", ["cfg2", "cfg4"])
def test_full_size_properties(hip, cfg):
    """BASELINE.json sizes, checked through size-independent properties (the oracle takes ~minutes there):
    chi2 never increases over accepted steps, the reduced matrix is symmetric positive definite and the device
    solve satisfies it, a second run from reset reproduces the first to round-off."""
    P = synth.make_problem(cfg, seed=1)
    s = hip.create(P, B.default_options())
    st = s.run()
    assert st["n_iters"] >= 5
    for it in st["iters"]:
        assert it["chi2_after"] <= it["chi2_before"] * (1 + 1e-12)
        if it["accepted"]:
            assert it["chi2_after"] < it["chi2_before"]
    first, last = st["iters"][0], st["iters"][-1]
    assert last["chi2_after"] < 0.2 * first["chi2_before"]
    nn = C.c_int32()
    hip.lib.ppo_ba_debug_set_lambda(s.h, C.c_double(1.0))
    hip.lib.ppo_ba_debug_reduced(s.h, None, None, None, C.byref(nn))
    n = nn.value  # dense system actually factorised (poses only when the cuboids are eliminated)
    assert n <= st["n_reduced"]
    S, b, x = np.zeros((n, n)), np.zeros(n), np.zeros(n)
    assert hip.lib.ppo_ba_debug_reduced(s.h, dptr(S), dptr(b), dptr(x), None) == 0
    assert np.abs(S - S.T).max() == 0.0
    w = np.linalg.eigvalsh(S)
    assert w.min() > 0
    assert np.linalg.norm(S @ x - b) <= 1e-7 * np.linalg.norm(b)
    r1 = s.read()
    s.reset()
    st2 = s.run()
    r2 = s.read()
    assert len(st2["iters"]) == len(st["iters"])
    assert abs(st2["iters"][-1]["chi2_after"] - last["chi2_after"]) <= 1e-6 * last["chi2_after"]
    assert np.sqrt(np.mean((r1.pose_qt[:, 4:] - r2.pose_qt[:, 4:]) ** 2)) < 1e-6


@pytest.mark.parametrize("det,lm", [(1, 0), (1, 1), (0, 0), (0, 1)], ids=["lm-on-device", "lm-on-host", "atomics-lm-on-device", "atomics-lm-on-host"])
def test_rejected_trials_follow_the_reference_schedule(oracle, hip, det, lm):
    """A badly initialised graph (20x the usual noise) makes LM reject trials: lambda grows by ni = 2, 4, 8 ..., the kept state's
    system is still in its buffer and the per-iteration trial counts must match the oracle's -- through BOTH passes (30 solve
    attempts for 15 iterations), in the deterministic (default) and in the floating-point-atomics mode.  Far from the optimum the 1e-7 noise of the numeric
    Jacobians is amplified from iteration to iteration: measured on MI355X (tools/r3_diag.py) chi2 per iteration agrees to 4e-5,
    lambda to 6e-3, and the final states differ by t 8e-5 m / r 2e-5 rad / planes 7e-5 -- a QUARTER of what the ORACLE's own result
    moves on this window when only its rounding changes (reversed summation, FMA contraction, compensated sums: t 3.5e-4, planes
    3e-4).  So the states are held to "inside the reference algorithm's own spread", the cost reached to the direct check: the
    oracle's objective evaluated at the device's final state equals its own minimum to 1e-6 (measured 3.5e-8)."""
    import sensitivity as S
    P = synth.make_problem("small", seed=2, noise_scale=20.0)
    so, sh = both(oracle, hip, P, deterministic=det, lm_control=lm)
    to, th = so.run(), sh.run()
    assert sum(i["trials"] for i in to["iters"]) >= len(to["iters"]) + 10 and len(to["iters"]) == 15, "the case no longer exercises rejections in both passes"
    assert [(i["pass_"], i["iteration"], i["trials"], i["accepted"], i["result"]) for i in to["iters"]] == \
           [(i["pass_"], i["iteration"], i["trials"], i["accepted"], i["result"]) for i in th["iters"]]
    # per-iteration bars.  Deterministic mode (reproducible): chi2 2e-4, lambda 3e-2 (measured 4e-5 / 6e-3).  Atomics mode: the same, or five times
    # what the ORACLE's own trace moves by on this window when only its rounding changes (its four re-runs: chi2 up to 1.7e-4, lambda 2.3e-2
    # per iteration) -- the order of the floating-point atomics is one more such change, and in about one run in twenty it is amplified
    # to 5.5e-4 at an iteration that needed three solve attempts (40 repeats on MI355X, profiles/r04_suite_runs.txt, round 4).
    runs = S.oracle_runs(oracle, P, B.default_options(), variants=S.VARIANTS)
    own_chi = own_lam = 0.0
    for st2, _ in runs[1:]:
        assert S.schedule(st2) == S.schedule(to)
        own_chi = max(own_chi, max(abs(a["chi2_after"] - b["chi2_after"]) / abs(a["chi2_after"]) for a, b in zip(to["iters"], st2["iters"])))
        own_lam = max(own_lam, max(abs(a["lambda_"] - b["lambda_"]) / abs(a["lambda_"]) for a, b in zip(to["iters"], st2["iters"])))
    bar_chi, bar_lam = (2e-4, 3e-2) if det else (max(2e-4, 5 * own_chi), max(3e-2, 5 * own_lam))
    for a, b in zip(to["iters"], th["iters"]):
        assert abs(a["chi2_after"] - b["chi2_after"]) <= bar_chi * abs(a["chi2_after"]), (a["chi2_after"], b["chi2_after"], bar_chi)
        assert abs(a["lambda_"] - b["lambda_"]) <= bar_lam * abs(a["lambda_"]), (a["lambda_"], b["lambda_"], bar_lam)
    ro, rh = so.read(), sh.read()
    gap, c_oracle, c_device = S.optimality_gap(oracle, so, ro, rh)
    if det:
        assert gap <= 1e-6, gap
    else:
        # Floating-point-atomics mode.  This window is NOT converged after 5 + 10 iterations (13 of its 30 solve attempts are rejected), so where a run stands
        # at the end hangs on every gain ratio on the way -- a small difference of two large chi2 values.  194 of 200 runs on MI355X
        # (tools/rejected_window_diag.py, profiles/r04_rejected_window_diag.txt, round 4) stay within 4e-8 of the oracle's final cost; in the other
        # six the gain ratio of pass 2 / iteration 3 comes out 0.15 % lower (chi2 19.00 instead of 19.03 gained), lambda follows 0.7 % higher
        # from there on, and the run ends 1.67e-5 BELOW the oracle's final cost -- the same accept / reject sequence, the same levels, a
        # slightly better iterate.  Held: never worse than the oracle's final cost by more than 1e-6, never further than 1e-4 from it.
        assert (c_device - c_oracle) <= 1e-6 * abs(c_oracle), (c_device, c_oracle)
        assert gap <= 1e-4, gap
    m = S.metrics(ro, rh)
    oo, _, _ = S.spread(runs)
    print("rejected-trials window (det=%d): deviation / oracle spread %s" % (det, {k: "%.1e / %.1e" % (m[k], oo[k]) for k in S.BARS}))
    for k, bar in S.BARS.items():
        if det:  # inside the bar, or inside what the reference algorithm's own rounding leaves open on this window
            assert m[k] <= max(bar, oo[k]), (k, m[k], oo[k])
        else:    # the atomics-mode rule of test_atomics_mode_stays_near_the_deterministic_result (measured: up to 2 x the oracle's spread)
            assert m[k] <= max(3 * bar, 5 * oo[k]), (k, m[k], oo[k])


def _not_positive_window():
    """The `small` window with ONE bbox observation carrying a NEGATIVE information weight (no robust kernel on it): the cuboid's
    block, and with it the reduced system, is indefinite by a wide margin wherever the damping is small.  With lm_tau = 3.566e-4
    and lm_max_trials = 2 the oracle (LDLT + `isPositive()`, linear_solver_dense.h:108-112) walks: iteration 0 -- lambda_0 = 1710,
    smallest eigenvalue +383: solved, accepted; iteration 1 -- lambda_0 / 3 and 2 lambda_0 / 3, smallest eigenvalue -9e3: BOTH
    factorisations fail, the trials are rejected, Terminate (found by scanning tau with the oracle; the pattern holds for a factor
    1.8 of tau around this value)."""
    P = synth.make_problem("small", seed=1)
    info = P["bbox_info"].copy().reshape(-1, 4)
    hub = P["bbox_huber"].copy()
    info[0, :] = -0.5
    hub[0] = 0.0
    P.set("bbox_info", info)
    P.set("bbox_huber", hub)
    return P, dict(lm_tau=1e-5 * 1.1 ** 37.5, lm_max_trials=2)


@LM_CONTROL
@pytest.mark.parametrize("elim", [1, 0])
def test_failed_factorisation_is_a_rejected_trial(oracle, hip, elim, lm, det):
    """The not-positive path (linear_solver_dense.h:108-112 -> levenberg.cpp:110-127: `if (!ok2) tempChi = max` -> rho < 0 -> lambda *= ni,
    pop) on the device: the native Cholesky reports the non-positive pivot (LAPACK-style info), the trial is rejected, lambda follows
    the reference's schedule, the kept state is untouched, and after ten... here two failures the iteration Terminates.  Schedule,
    lambda, chi2 against the oracle; then pass 2 runs on from the kept state."""
    P, kw = _not_positive_window()
    kw = dict(kw, lm_control=lm, deterministic=det)
    so, sh = both(oracle, hip, P, iters_pass2=0, eliminate_cuboids=elim, **kw)
    to, th = so.run(), sh.run()
    sched = [(i["pass_"], i["iteration"], i["trials"], i["accepted"], i["result"]) for i in to["iters"]]
    assert sched == [(0, 0, 1, 1, 0), (0, 1, 2, 0, 1)], "the case no longer produces: solved + accepted, then two failed factorisations"
    assert sched == [(i["pass_"], i["iteration"], i["trials"], i["accepted"], i["result"]) for i in th["iters"]]
    for a, b in zip(to["iters"], th["iters"]):
        assert abs(a["lambda_"] - b["lambda_"]) <= 1e-6 * a["lambda_"]
        assert abs(a["chi2_before"] - b["chi2_before"]) <= 1e-6 * a["chi2_before"] and abs(a["chi2_after"] - b["chi2_after"]) <= 1e-6 * a["chi2_after"]
    assert to["iters"][1]["chi2_after"] == to["iters"][1]["chi2_before"] and th["iters"][1]["chi2_after"] == th["iters"][1]["chi2_before"]  # nothing accepted
    ro, rh = so.read(), sh.read()
    compare_states(ro, rh)  # both kept the state iteration 0 produced
    # the system really is indefinite there: the device's own reduced matrix has a negative eigenvalue at the lambda of the failed trials
    hip.lib.ppo_ba_debug_set_lambda(sh.h, C.c_double(th["iters"][1]["lambda_"]))
    n = C.c_int32()
    hip.lib.ppo_ba_debug_reduced(sh.h, None, None, None, C.byref(n))
    Sm = np.zeros((n.value, n.value))
    assert hip.lib.ppo_ba_debug_reduced(sh.h, dptr(Sm), None, None, None) == 0
    assert not np.isfinite(Sm).all() or np.linalg.eigvalsh(Sm).min() < -1.0
    so.close(); sh.close()
    # the complete protocol: pass 1 as above, then the classification and pass 2 from the kept state
    so, sh = both(oracle, hip, P, eliminate_cuboids=elim, **kw)
    to, th = so.run(), sh.run()
    assert [(i["trials"], i["accepted"], i["result"]) for i in th["iters"][:2]] == [(1, 1, 0), (2, 0, 1)]
    # (pass 2 may differ from the oracle's: whether the negative-weight observation is culled in between hangs on the stored errors of
    #  the failed trials -- the one deliberate deviation, pinned by the next test; with it still active pass 2 fails and terminates at once)
    assert len(th["iters"]) > 2 and all(np.isfinite(i["chi2_after"]) and i["chi2_after"] <= i["chi2_before"] for i in th["iters"])
    so.close(); sh.close()


@LM_CONTROL
def test_failed_factorisation_leaves_the_kept_states_errors_behind__the_one_deliberate_deviation(oracle, hip, lm, det):
    """DESIGN.md section 2 lists ONE deviation of the product from the reference's control flow, and this test pins it.  When a
    factorisation fails the reference still applies `_solver->x()` -- the STALE solution of the last successful solve --, evaluates
    every edge there, and only then pops the vertices (levenberg.cpp:110-146): the edges' stored `_error` (what `e->chi2()` returns
    to the outlier classification, Optimizer.cc:2739-2814) are those of kept state (+) stale step.  The device rejects the trial
    without touching anything: its stored errors are the kept state's.  Same estimates, different per-edge chi2 after a pass that
    ENDS in failed factorisations -- which needs lm_max_trials consecutive failures right after a success, i.e. an indefinite
    problem like this one; a positive semi-definite system (every real window: J' W J + lambda I) cannot get there."""
    P, kw = _not_positive_window()
    so, sh = both(oracle, hip, P, iters_pass2=0, lm_control=lm, deterministic=det, **kw)
    so.run(); sh.run()
    ro, rh = so.read(), sh.read()
    compare_states(ro, rh)
    # per-edge chi2 AT the kept state, by the oracle, on a graph that starts there
    Pk = synth.make_problem("small", seed=1)
    for name in ("bbox_info", "bbox_huber"):
        Pk.set(name, P[name])
    Pk.set("pose_qt", ro.pose_qt); Pk.set("point_xyz", ro.point_xyz); Pk.set("plane_coef", ro.plane_coef)
    Pk.set("cuboid_qt", ro.cuboid_qt); Pk.set("cuboid_scale", ro.cuboid_scale)
    sk = oracle.create(Pk, B.default_options(**kw))
    sk.linearize()
    rk = sk.read()
    dev = np.abs(rh.mono_chi2 - rk.mono_chi2).max() / max(1.0, np.abs(rk.mono_chi2).max())
    ref = np.abs(ro.mono_chi2 - rk.mono_chi2).max() / max(1.0, np.abs(rk.mono_chi2).max())
    assert dev <= 1e-5, dev   # the device: the kept state's errors
    assert ref >= 1e-2, ref   # the reference algorithm: the errors of (kept state + stale step) -- a different set
    so.close(); sh.close(); sk.close()


@LM_CONTROL
def test_nan_residual_freezes_the_window_like_the_reference(oracle, hip, lm, det):
    """`ominus_ver` divides by |n x n_m| (G2O_Plane3D.h:97-106): a vertical-plane measurement whose normal is PARALLEL to the predicted one
    gives a NaN residual.  In the reference chi2 is then NaN, every comparison on it is false: each LM iteration makes one solve
    attempt, rejects it (`rho > 0` is false), leaves the loop (`rho < 0` is false too), never counts as bad -- 5 + 10 iterations that
    change nothing.  The device must walk the same schedule, keep every estimate, report NaN for that edge and neither hang nor
    fail (the parallel pair is made exact: identity keyframe pose, plane (0, 0, 1, 2), measurement (0, 0, 1, 3)).
    In the floating-point-atomics mode the NaN flows through the kernels as it does through g2o; in the deterministic mode (round 6) a non-finite
    addend is left out of the integer sums and flagged, the flag makes chi2 of that state NaN, and a NaN chi2 of the ESTIMATE freezes the pass
    exactly as the reference's comparisons do (csrc/ppo_solver.hip: frozen_iteration) -- it used to be an error."""
    P = synth.make_problem("small", seed=4, with_vp_planes=2)
    e = int(np.nonzero(P["pl_kind"] == 1)[0][0])
    pl, ps = int(P["pl_plane"][e]), int(P["pl_pose"][e])
    qt = P["pose_qt"].copy(); qt[ps] = [0, 0, 0, 1, 0, 0, 0]; P.set("pose_qt", qt)
    fx = P["pose_fixed"].copy(); fx[ps] = 1; P.set("pose_fixed", fx)
    pc = P["plane_coef"].copy(); pc[pl] = [0, 0, 1, 2]; P.set("plane_coef", pc)
    m = P["pl_meas"].copy().reshape(-1, 4); m[e] = [0, 0, 1, 3]; P.set("pl_meas", m)
    so, sh = both(oracle, hip, P, lm_control=lm, deterministic=det)
    to, th = so.run(), sh.run()
    assert [(i["pass_"], i["trials"], i["accepted"], i["result"]) for i in to["iters"]] == [(0, 1, 0, 0)] * 5 + [(1, 1, 0, 0)] * 10
    assert [(i["pass_"], i["trials"], i["accepted"], i["result"]) for i in th["iters"]] == [(0, 1, 0, 0)] * 5 + [(1, 1, 0, 0)] * 10
    assert all(np.isnan(i["chi2_before"]) for i in th["iters"])
    # lambda grows by ni = 2, 4, 8, 16 from one rejected iteration to the next on both sides.  (lambda_0 itself = tau * max |H_jj| is not
    # compared: with NaN diagonal entries g2o's running `std::max(fabs(h), m)` restarts behind every NaN it meets -- an accident of the
    # vertex order; the device takes the largest finite entry.)
    for tr in (to, th):
        lam = [i["lambda_"] for i in tr["iters"][:5]]
        assert np.isfinite(lam).all() and np.allclose(np.array(lam[1:]) / np.array(lam[:-1]), [2, 4, 8, 16], rtol=1e-12), lam
    ro, rh = so.read(), sh.read()
    assert np.isnan(ro.pl_chi2[e]) and np.isnan(rh.pl_chi2[e]) and rh.pl_level[e] == ro.pl_level[e] == 0
    for name in ("pose_qt", "point_xyz", "plane_coef", "cuboid_qt", "cuboid_scale"):
        np.testing.assert_allclose(getattr(rh, name), getattr(ro, name), rtol=0, atol=1e-12)
    np.testing.assert_allclose(rh.point_xyz, P["point_xyz"], rtol=0, atol=0)
    so.close(); sh.close()


def _window_in_other_units():
    """A points-only window, and the same window with every information matrix times 2^28 and every Huber width times 2^14 (chi2, H, b and lambda all scale by
    exactly 2^28: the same problem, as well conditioned, with terms J^2 Omega ~ 1e13 -- beyond the fixed-point range); the options that go with the second."""
    P0 = synth.make_problem(dict(n_kf=8, n_points=300, n_planes=0, n_cuboids=0), seed=4)
    P = synth.make_problem(dict(n_kf=8, n_points=300, n_planes=0, n_cuboids=0), seed=4)
    k = 2.0 ** 28
    P.set("mono_info", P["mono_info"] * k)
    P.set("mono_huber", P["mono_huber"] * 2.0 ** 14)
    return P0, P, k, dict(chi2_mono=5.991 * k, chi2_stereo=7.815 * k)


@pytest.mark.parametrize("lm", [0, 1], ids=["lm-on-device", "lm-on-host"])
def test_out_of_range_term_leaves_the_deterministic_mode(hip, lm):
    """A FINITE term that does not fit the 2^-24 fixed-point range (+-4e11) cannot enter the integer sums.  The reference sums doubles and takes such a window,
    so the handle leaves the deterministic mode at that linearisation and goes on with floating-point sums (ppo_ba_stats.deterministic_fell_back, include/ppo_ba.h).
    The window: points only, every information matrix times 2^28 and every Huber width times 2^14 -- the SAME problem in other units (chi2, H, b, lambda all
    scale by exactly 2^28), so it is as well conditioned as the original and its terms (J^2 Omega ~ 1e13) are out of range.  The handle that fell back follows
    a handle created with deterministic = 0 (same schedule, chi2 history to 1e-7, poses to 1e-7), also on a second run after a reset, and ends where the
    deterministic run of the unscaled window ends; a window whose terms fit never leaves."""
    P0, P, k, scaled = _window_in_other_units()
    scaled["lm_control"] = lm
    sa = hip.create(P, B.default_options(**scaled, **ATOMICS))
    sta = sa.run()
    ra = sa.read()
    sa.close()
    assert sta["deterministic_fell_back"] == 0 and len(sta["iters"]) > 0
    s0 = hip.create(P0, B.default_options(lm_control=lm))
    st0 = s0.run()
    r0 = s0.read()
    s0.close()
    assert st0["deterministic_fell_back"] == 0
    sd = hip.create(P, B.default_options(**scaled))
    for rep in range(2):
        std = sd.run()
        rd = sd.read()
        assert std["deterministic_fell_back"] == 1
        assert [(i["trials"], i["accepted"]) for i in std["iters"]] == [(i["trials"], i["accepted"]) for i in sta["iters"]]
        np.testing.assert_allclose([i["chi2_after"] for i in std["iters"]], [i["chi2_after"] for i in sta["iters"]], rtol=1e-7)
        np.testing.assert_allclose(rd.pose_qt, ra.pose_qt, atol=1e-7)
        np.testing.assert_allclose(rd.point_xyz, ra.point_xyz, atol=1e-6)
        # ... and the unscaled window in deterministic mode: the same problem
        assert [(i["trials"], i["accepted"]) for i in std["iters"]] == [(i["trials"], i["accepted"]) for i in st0["iters"]]
        np.testing.assert_allclose([i["chi2_after"] / k for i in std["iters"]], [i["chi2_after"] for i in st0["iters"]], rtol=1e-7)
        np.testing.assert_allclose(rd.pose_qt, r0.pose_qt, atol=1e-7)
        np.testing.assert_array_equal(rd.mono_level, r0.mono_level)
        sd.reset()
    sd.close()


def test_block_cache_survives_changing_shapes(oracle, hip):
    """ppo_ba_destroy() parks device blocks, streams and the mapped host block in a process-wide cache; a solver that
    receives recycled (stale) blocks of another shape must give the same answer as the first one did."""
    PA = synth.make_problem("small", seed=1)
    PB = synth.make_problem(dict(n_kf=12, n_points=400, n_planes=6, n_cuboids=0), seed=7)
    first = None
    for P in (PA, PB, PA, PB, PA):
        s = hip.create(P, B.default_options())
        s.run()
        r = s.read()
        s.close()
        if P is PA:
            if first is None:
                first = r
            else:
                # run-to-run differences come from the order of the atomic sums: up to 2.3e-7 on the poses over 1500 repeats
                # (tools/stress_repeat.py; the monocular gauge is only weakly held), chi2 itself repeats to 1e-11
                np.testing.assert_allclose(r.pose_qt, first.pose_qt, atol=2e-6)
                np.testing.assert_allclose(r.point_xyz, first.point_xyz, atol=2e-6)
                np.testing.assert_allclose(r.cuboid_scale, first.cuboid_scale, atol=1e-5)
    assert hip.lib.ppo_ba_release_cached() == 0
    s = hip.create(PA, B.default_options())
    s.run()
    np.testing.assert_allclose(s.read().pose_qt, first.pose_qt, atol=2e-6)
    s.close()


def test_kernel_timing_hook_reports_every_kernel(hip):
    """ppo_ba_time_kernel: each kernel of the path launches alone, takes a positive time and states its algorithmic bytes."""
    P = synth.make_problem("small", seed=1)
    s = hip.create(P, B.default_options())
    s.linearize()
    n = C.c_int32()
    hip.lib.ppo_ba_debug_reduced(s.h, None, None, None, C.byref(n))
    b = np.zeros(n.value)
    assert hip.lib.ppo_ba_debug_reduced(s.h, None, dptr(b), None, None) == 0
    for which in range(7):
        ms, by = C.c_double(), C.c_int64()
        assert hip.lib.ppo_ba_time_kernel(s.h, which, 3, C.byref(ms), C.byref(by)) == 0
        if which == 2:  # PPO_BA_KERNEL_SCHUR_Z is retired (folded into the pair kernel)
            assert ms.value == 0
            continue
        assert ms.value > 0 and by.value > 0, which
    assert hip.lib.ppo_ba_time_kernel(s.h, 99, 3, C.byref(ms), C.byref(by)) != 0
    s.close()


def test_phase_times_are_collected_in_verbose_mode_and_do_not_change_the_result(hip):
    """ppo_ba_stats.ms_*: device time per phase from events on the stream, only when options.verbose != 0."""
    P = synth.make_problem("small", seed=1)
    quiet = hip.create(P, B.default_options())
    tq = quiet.run()
    rq = quiet.read()
    quiet.close()
    assert tq["ms_linearize"] == 0 and tq["ms_factor"] == 0
    loud = hip.create(P, B.default_options(verbose=1))
    tl = loud.run()
    rl = loud.read()
    loud.close()
    phases = [tl[k] for k in ("ms_linearize", "ms_schur", "ms_factor", "ms_backsub", "ms_chi2")]
    assert all(v > 0 for v in phases), phases
    assert sum(phases) <= tl["wall_ms_total"] * 1.05
    assert [i["trials"] for i in tl["iters"]] == [i["trials"] for i in tq["iters"]]
    assert np.abs(rl.pose_qt - rq.pose_qt).max() < 1e-5


def _schedules_may_differ(to, th, runs):
    """When the device and the oracle walk different accept / reject sequences: is that a property of the window?  Yes if the
    oracle's own re-runs (summation order, FMA contraction, compensated sums) flip among themselves; or the FIRST iteration that
    differs is a marginal one in the oracle's own trace -- it needed more than one solve attempt there (rejections: the gain
    ratio's sign is being decided near zero) or moved chi2 by less than 1e-7 relative (convergence plateau)."""
    import sensitivity as S
    if any(S.schedule(st) != S.schedule(runs[0][0]) for st, _ in runs[1:]):
        return True
    so_, sh_ = S.schedule(to), S.schedule(th)
    first = next((k for k, (x, y) in enumerate(zip(so_, sh_)) if x != y), min(len(so_), len(sh_)) - 1)
    for k in (first, max(first - 1, 0)):
        a = to["iters"][k]
        if a["trials"] > 1 or abs(a["chi2_before"] - a["chi2_after"]) < 1e-7 * a["chi2_before"]:
            return True
    # (round 5: the clause "the DEVICE needed more than one attempt there" is gone -- a kernel that wrongly rejected a trial excused itself with it;
    #  every remaining excuse is derived from the oracle's own runs)
    return False


def _random_sweep_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("random_sweep", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "random_sweep.py"))
    rs = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rs)
    return rs


N_RANDOM_WINDOWS = 60


def test_randomly_structured_windows_against_the_oracle(oracle, hip):
    """60 windows of random size and composition (3-23 keyframes, 30-700 points, 0-8 planes, 0-4 cuboids, stereo share, corner
    edges, vertical / parallel planes, fixed observers, 0-20 % gross outliers, both elimination modes; tools/random_sweep.py), the
    device in its DEFAULT mode, which since round 6 is the deterministic one (options.deterministic = 1: bit-reproducible, so this test has no
    run-to-run component, and it is the mode bench.py times; the opt-in floating-point-atomics mode is held to this result by the next test).  The only excuse for a deviation is the ORACLE's own spread -- how far
    the reference algorithm's result moves on that window when nothing but its rounding changes (summation order, FMA contraction,
    compensated sums: tests/sensitivity.py).  Three requirements per window (numbers: tools/strict_sweep.py over 300 windows,
    profiles/r04_strict_sweep.txt):
      1. DIRECT OPTIMALITY -- the oracle's objective (its pass-2 active set, levels and kernels) evaluated at the DEVICE's final
         state equals the oracle's own minimum: relative gap <= max(5e-7, 5 x the gap of the oracle's own four re-runs) when the device
         walked the oracle's accept / reject sequence (300 windows: never violated), <= 1e-4 when it did not and requirement 3 holds.
      2. every quantity of BASELINE.md's bar is inside the bar, OR the oracle's own spread of that quantity on that window reaches 0.3 of
         the deviation (4 of 300 windows: plane coefficients 1.1e-6 against a spread of 3.6e-7, a rotation, a cuboid).  The point RMSE is
         taken over the points the second pass determines (>= 2 active observations, sensitivity.determined_points); the others -- on a
         flat direction of the cost -- must stay within 1e-2 m.
      3. the same accept / reject sequence, or a demonstrably marginal one (_schedules_may_differ; 2 of 300)."""
    import sensitivity as S
    rs = _random_sweep_module()
    n = N_RANDOM_WINDOWS
    n_out = n_sched = 0
    for seed in range(3000, 3000 + n):
        cfg, kw, elim = rs.random_case(seed)
        P = synth.make_problem(cfg, **kw)
        opt = B.default_options(eliminate_cuboids=elim)
        so, sh = oracle.create(P, opt), hip.create(P, B.default_options(eliminate_cuboids=elim, deterministic=1))
        to, th = so.run(), sh.run()
        ro, rh = so.read(), sh.read()
        pts = S.determined_points(P, ro)
        m = S.metrics(ro, rh, pts)
        assert m["t"] < 1e-3 and m["r"] < 1e-3, (seed, cfg, kw, m)  # nothing diverges
        if ro.point_xyz.size:
            assert np.linalg.norm(ro.point_xyz - rh.point_xyz, axis=1).max() < 1e-2, (seed, cfg, kw)
        runs = S.oracle_runs(oracle, P, opt, variants=S.VARIANTS)
        own_gap = max(S.optimality_gap(oracle, so, ro, r)[0] for _, r in runs[1:])
        gap, _, _ = S.optimality_gap(oracle, so, ro, rh)
        same_schedule = S.schedule(to) == S.schedule(th)
        if same_schedule:
            assert gap <= max(5e-7, 5 * own_gap), (seed, gap, own_gap, cfg, kw)
        else:
            assert gap <= 1e-4, (seed, gap, own_gap, cfg, kw)
        oo, _, _ = S.spread(runs, pts)
        out = [k for k, bar in S.BARS.items() if m[k] >= bar]
        for k in out:
            assert oo[k] >= 0.3 * m[k], (seed, k, m[k], oo[k], cfg, kw)
        n_out += bool(out)
        if not same_schedule:
            n_sched += 1
            assert _schedules_may_differ(to, th, runs), (seed, cfg, kw, S.schedule(to), S.schedule(th))
        so.close(); sh.close()
    print("random windows, deterministic mode: %d of %d used the oracle-spread excuse, %d walked another (marginal) schedule" % (n_out, n, n_sched))
    # drift guard: measured 0 windows outside the bar and 1 differing schedule among these 60 (4 and 2 among 300)
    assert n_out <= 3 and n_sched <= 3, (n_out, n_sched)


def test_atomics_mode_stays_near_the_deterministic_result(oracle, hip):
    """The same 60 windows in the opt-in floating-point-atomics mode (options.deterministic = 0: the order in which the addends of a cross-wave sum
    arrive changes from run to run; ~1 % faster, NOT what bench.py times) against the deterministic result of the same library: that isolates the noise of the atomic order from any error of
    a kernel, which the previous test would have caught.  Bound per quantity of BASELINE.md's bar: max(3 x the bar, 5 x the oracle's own
    spread on that window).  Measured over 300 windows (tools/strict_sweep.py): never violated; the largest ratios to the bar are
    translation 3.8, rotation 11, point RMSE 10, plane 14, cuboid 1.2 -- all on windows whose oracle spread is as large (seed 3178: plane
    spread 9.2e-6) or where the two runs walk different, marginal schedules (seed 3220)."""
    import sensitivity as S
    rs = _random_sweep_module()
    n_sched = n_beyond_bar = 0
    for seed in range(3000, 3000 + N_RANDOM_WINDOWS):
        cfg, kw, elim = rs.random_case(seed)
        P = synth.make_problem(cfg, **kw)
        sd = hip.create(P, B.default_options(eliminate_cuboids=elim, deterministic=1))
        s0 = hip.create(P, B.default_options(eliminate_cuboids=elim, **ATOMICS))
        td, t0 = sd.run(), s0.run()
        rd, r0 = sd.read(), s0.read()
        sd.close(); s0.close()
        pts = S.determined_points(P, rd)
        m = S.metrics(rd, r0, pts)
        beyond = [k for k, bar in S.BARS.items() if m[k] >= bar]
        if beyond or S.schedule(td) != S.schedule(t0):
            runs = S.oracle_runs(oracle, P, B.default_options(eliminate_cuboids=elim), variants=S.VARIANTS)
            oo, _, _ = S.spread(runs, S.determined_points(P, runs[0][1]))
            for k in beyond:
                assert m[k] < max(3 * S.BARS[k], 5 * oo[k]), (seed, k, m[k], oo[k], cfg, kw)
            if S.schedule(td) != S.schedule(t0):
                n_sched += 1
                assert _schedules_may_differ(td, t0, runs) or _schedules_may_differ(runs[0][0], t0, runs), (seed, cfg, kw, S.schedule(td), S.schedule(t0))
        n_beyond_bar += bool(beyond)
    print("random windows, atomics vs deterministic mode: %d of %d beyond the bar somewhere, %d with another (marginal) schedule" % (n_beyond_bar, N_RANDOM_WINDOWS, n_sched))
    assert n_beyond_bar <= 8 and n_sched <= 4, (n_beyond_bar, n_sched)


@pytest.mark.parametrize("on_owner", [0, 1])
def test_sharded_code_path_with_a_one_rank_rccl_communicator(hip_hooks, monkeypatch, on_owner):
    """PPO_BA_COMM_SELFTEST=1 makes the library open librccl.so, build a ONE-rank communicator and take every branch of the
    N > 1 path (all-reduce of the pose blocks after each linearisation, of (S, b) per solve attempt, of the chi2 / scale /
    max-diagonal scalars, the landmark gather in read) on a single GPU.  Multi-GPU boxes are not available to these tests;
    the arithmetic of the sharding itself is covered by tests/test_sharding_gloo.py."""
    hip = hip_hooks
    P = synth.make_problem("small", seed=1)
    s = hip.create(P, B.default_options())
    ref_stats, ref = s.run(), s.read()
    s.close()
    monkeypatch.setenv("PPO_BA_COMM_SELFTEST", "1")
    # on_owner = 1: options.solve_on_owner -- ncclReduce of (S, b) to rank 0 + ncclBroadcast of x and the flag instead of the all-reduce
    s = hip.create(P, B.default_options(solve_on_owner=on_owner))
    st, r = s.run(), s.read()
    out = np.zeros(4)
    hip.lib.ppo_ba_time_collectives.argtypes = [C.c_void_p, C.c_int, B.c_dp]
    assert hip.lib.ppo_ba_time_collectives(s.h, 5, out.ctypes.data_as(B.c_dp)) == 0
    assert out[0] > 0 and out[1] > 0 and out[2] > 0 and out[3] > 8 * 4096, out  # the three exchange steps, timed alone through RCCL
    s.close()
    assert [(i["trials"], i["accepted"]) for i in st["iters"]] == [(i["trials"], i["accepted"]) for i in ref_stats["iters"]]
    assert abs(st["iters"][-1]["chi2_after"] - ref_stats["iters"][-1]["chi2_after"]) <= 1e-7 * ref_stats["iters"][-1]["chi2_after"]
    np.testing.assert_allclose(r.pose_qt, ref.pose_qt, atol=2e-6)
    np.testing.assert_allclose(r.point_xyz, ref.point_xyz, atol=1e-5)
    np.testing.assert_allclose(r.plane_coef, ref.plane_coef, atol=1e-6)


def _run_sharded_in_process(hip, P, world, key, **opt):
    """`world` ranks of ONE process on ONE device, each on its own host thread, joined by the library's loopback communicator
    (comm_id "PPOLOOP0" + key; csrc/ppo_comm.hpp).  Returns the per-rank (stats, result)."""
    import threading
    out, err = [None] * world, [None] * world

    def worker(rank):
        try:
            o = B.default_options(rank=rank, world_size=world, **opt)
            ident = b"PPOLOOP0" + int(key).to_bytes(8, "little")
            for i, b in enumerate(ident):
                o.comm_id[i] = b
            s = hip.create(P, o)
            st = s.run()
            out[rank] = (st, s.read())
            s.close()
        except Exception as e:  # noqa: BLE001
            err[rank] = e

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
        assert not t.is_alive(), "a rank is stuck in a collective"
    for e in err:
        if e is not None:
            raise e
    return out


@pytest.mark.parametrize("on_owner", [0, 1])
@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("cfg,kw", [("small", dict(seed=1)), ("small", dict(seed=2, stereo_frac=0.3)),
                                    (dict(n_kf=12, n_points=400, n_planes=6, n_cuboids=0), dict(seed=7))])
def test_landmark_sharding_gives_the_single_rank_answer(hip_hooks, cfg, kw, world, on_owner):
    """The N > 1 path for real: landmark ranges per rank, partial reduced systems summed by a collective, every rank
    factoring redundantly (on_owner = 0) or rank 0 alone factoring and broadcasting the solution (options.solve_on_owner = 1),
    landmarks gathered in read -- against the same graph solved by one rank."""
    hip = hip_hooks
    P = synth.make_problem(cfg, **kw)
    s = hip.create(P, B.default_options())
    st1, r1 = s.run(), s.read()
    s.close()
    res = _run_sharded_in_process(hip, P, world, key=1000 + world * 17 + kw["seed"] + 500 * on_owner, solve_on_owner=on_owner)
    for st, r in res:
        assert [(i["trials"], i["accepted"], i["result"]) for i in st["iters"]] == [(i["trials"], i["accepted"], i["result"]) for i in st1["iters"]]
        for a, b in zip(st["iters"], st1["iters"]):
            assert abs(a["chi2_after"] - b["chi2_after"]) <= 1e-6 * abs(b["chi2_after"])  # summation order differs
        compare_states(r1, r)  # the same bar as against the oracle (RMSE: single two-view points move more along their ray)
        assert np.array_equal(r.mono_level, r1.mono_level)
        assert np.array_equal(r.pl_level, r1.pl_level)
        np.testing.assert_allclose(r.mono_chi2, r1.mono_chi2, atol=1e-5 * max(1.0, np.abs(r1.mono_chi2).max()))
    # every rank holds the same poses bit for bit (identical inputs to the redundant factorisation)
    for st, r in res[1:]:
        np.testing.assert_array_equal(r.pose_qt, res[0][1].pose_qt)


def test_sharded_ranks_leave_the_deterministic_mode_together(hip_hooks):
    """The out-of-range flags are summed over the ranks before any rank looks at them (csrc/ppo_solver.hip: fetch_acc_enqueue), so every rank leaves the fixed-point
    sums at the same linearisation (leave_fixed_point) and the collectives stay matched: two ranks on the window of
    test_out_of_range_term_leaves_the_deterministic_mode end where one rank ends, with the same schedule."""
    hip = hip_hooks
    _, P, _, scaled = _window_in_other_units()
    s = hip.create(P, B.default_options(**scaled))
    st1, r1 = s.run(), s.read()
    s.close()
    assert st1["deterministic_fell_back"] == 1
    res = _run_sharded_in_process(hip, P, 2, key=7311, **scaled)
    for st, r in res:
        assert st["deterministic_fell_back"] == 1
        assert [(i["trials"], i["accepted"], i["result"]) for i in st["iters"]] == [(i["trials"], i["accepted"], i["result"]) for i in st1["iters"]]
        for a, b in zip(st["iters"], st1["iters"]):
            assert abs(a["chi2_after"] - b["chi2_after"]) <= 1e-6 * abs(b["chi2_after"])
        np.testing.assert_allclose(r.pose_qt, r1.pose_qt, atol=1e-7)
        assert np.array_equal(r.mono_level, r1.mono_level)
    np.testing.assert_array_equal(res[1][1].pose_qt, res[0][1].pose_qt)


def _full_size_against_oracle(oracle, hip, P, with_spread, hip_only=None):
    """Complete 5 + 10 protocol, the device in its default (DETERMINISTIC, round 6: the timed) mode against the oracle: same accept / reject schedule, chi2 per iteration at
    1e-6, final estimates at BASELINE.md's bars, same outlier sets, and the DIRECT check: the oracle's objective evaluated at the
    device's final state equals its own minimum (relative gap <= max(1e-7, 5 x the gap between the oracle's own four rounding-only
    re-runs; measured <= 2 x, tools/optimality_sweep.py full).  Cuboids and points: each one inside the bar, OR within FIVE times that same
    cuboid's / point's spread between those re-runs of the ORACLE (measured <= 1.5 x) -- a deviation is only accepted where the reference
    algorithm's own result is not defined to the bar (tests/test_oracle_sensitivity.py).  No excuse comes from the device.
    Then the opt-in floating-point-atomics mode (options.deterministic = 0; not what bench.py times) is held to the deterministic result:
    same schedule, every quantity within max(3 x the bar, 5 x the oracle's spread of that cuboid / point / quantity)."""
    import sensitivity as S
    opt_det = B.default_options(deterministic=1, **(hip_only or {}))
    so, sh = oracle.create(P, B.default_options()), hip.create(P, opt_det)
    to, th = so.run(), sh.run()
    assert [(i["pass_"], i["iteration"], i["trials"], i["accepted"]) for i in to["iters"]] == [(i["pass_"], i["iteration"], i["trials"], i["accepted"]) for i in th["iters"]]
    for a, b in zip(to["iters"], th["iters"]):
        assert abs(a["chi2_after"] - b["chi2_after"]) <= 1e-6 * abs(a["chi2_after"])
    ro, rh = so.read(), sh.read()
    n = ro.pose_qt.shape[0]
    assert np.sqrt(np.mean(np.sum((ro.pose_qt[:, 4:] - rh.pose_qt[:, 4:]) ** 2, axis=1))) < 1e-5
    assert max(rot_angle(ro.pose_qt[i, :4], rh.pose_qt[i, :4]) for i in range(n)) < 1e-6
    if ro.plane_coef.size:
        assert np.abs(ro.plane_coef - rh.plane_coef).max() < 1e-6
    d_pt = S.per_point(ro, rh)
    d_cb = S.per_cuboid(ro, rh)
    p_rmse = np.sqrt(np.mean(d_pt ** 2))
    gap, _, _ = S.optimality_gap(oracle, so, ro, rh)
    # the floating-point-atomics mode next to it
    s0 = hip.create(P, B.default_options(**dict(hip_only or {}, **ATOMICS)))
    t0 = s0.run()
    r0 = s0.read()
    s0.close()
    m0 = S.metrics(rh, r0)
    d0_cb, d0_pt = S.per_cuboid(rh, r0), S.per_point(rh, r0)
    default_beyond = [k for k, bar in S.BARS.items() if m0[k] >= bar]
    # per-edge chi2 of the 9-D camera-object edges: 1e-5 of the largest -- or, edge by edge, five times what the ORACLE's own rounding-only re-runs move
    # that edge's chi2 (the edge hangs on a cuboid: where the cuboid sits on a flat direction, e.g. edge 127 of config 4 / seed 1, the oracle's re-runs
    # differ by 2.0e-5 there and every faithful implementation lands somewhere in that neighbourhood; same rule as for the cuboids themselves below)
    d_c3 = np.abs(ro.cub3d_chi2 - rh.cub3d_chi2) if ro.cub3d_chi2.size else np.zeros(0)
    bar_c3 = 1e-5 * max(1.0, np.abs(ro.cub3d_chi2).max()) if d_c3.size else 0.0
    need_spread = p_rmse >= 1e-5 or (d_cb.size and d_cb.max() >= 1e-5) or gap > 1e-7 or bool(default_beyond) or (d_c3.size and d_c3.max() > bar_c3)
    s_cb = s_pt = oo = None
    if need_spread or with_spread:
        # measured on MI355X (tools/optimality_sweep.py full, config 4 and its option sets): point RMSE 3e-8 .. 1.4e-5, one to five
        # of the 50 cuboids beyond 1e-5 (worst 2.6e-4 where the oracle's own re-runs differ by 4.1e-4), gap 1e-10 .. 1.6e-7
        runs = S.oracle_runs(oracle, P, B.default_options(), variants=S.VARIANTS)
        own_gap = max(S.optimality_gap(oracle, so, ro, r)[0] for _, r in runs[1:])
        assert gap <= max(1e-7, 5 * own_gap), (gap, own_gap)
        oo, s_cb, s_pt = S.spread(runs)
        if d_cb.size:
            assert (d_cb < np.maximum(1e-5, 5 * s_cb)).all(), (np.sort(d_cb)[::-1][:5], np.sort(s_cb)[::-1][:5])
            assert int((d_cb >= 1e-5).sum()) <= int((s_cb > 2e-6).sum())
            assert np.mean(d_cb < 1e-5) >= 0.9
        if p_rmse >= 1e-5:
            assert (d_pt < np.maximum(1e-5, 5 * s_pt)).all()
            assert np.sqrt(np.mean(d_pt[s_pt < 1e-6] ** 2)) < 1e-5
        if d_c3.size:
            s_c3 = np.zeros_like(d_c3)
            for _, r_v in runs[1:]:
                s_c3 = np.maximum(s_c3, np.abs(r_v.cub3d_chi2 - runs[0][1].cub3d_chi2))
            assert (d_c3 <= np.maximum(bar_c3, 5 * s_c3)).all(), (np.sort(d_c3)[::-1][:5], np.sort(s_c3)[::-1][:5])
            assert int((d_c3 > bar_c3).sum()) <= int((s_c3 > 0.2 * bar_c3).sum())
    assert abs(to["iters"][-1]["chi2_after"] - th["iters"][-1]["chi2_after"]) <= 1e-6 * to["iters"][-1]["chi2_after"]
    o = B.default_options()
    compare_levels(ro.mono_level, rh.mono_level, ro.mono_chi2, o.chi2_mono)
    compare_levels(ro.stereo_level, rh.stereo_level, ro.stereo_chi2, o.chi2_stereo)
    compare_levels(ro.pl_level, rh.pl_level, ro.pl_chi2, o.chi2_plane)
    assert np.array_equal(ro.bbox_level, rh.bbox_level)
    assert np.array_equal(ro.corner_level, rh.corner_level)
    assert np.array_equal(ro.cub3d_level, rh.cub3d_level)
    if d_c3.size and s_cb is None:
        assert d_c3.max() <= bar_c3
    # ---- floating-point-atomics mode against the deterministic result ----
    assert S.schedule(t0) == S.schedule(th), (S.schedule(t0), S.schedule(th))
    assert abs(t0["iters"][-1]["chi2_after"] - th["iters"][-1]["chi2_after"]) <= 1e-6 * th["iters"][-1]["chi2_after"]
    for k in default_beyond:
        if k == "cb":
            assert (d0_cb < np.maximum(3 * S.BARS["cb"], 5 * s_cb)).all(), (np.sort(d0_cb)[::-1][:5], np.sort(s_cb)[::-1][:5])
        elif k == "p":
            assert (d0_pt < np.maximum(3 * S.BARS["p"], 5 * s_pt)).all()
        else:
            assert m0[k] < max(3 * S.BARS[k], 5 * oo[k]), (k, m0[k], oo[k])
    print("full size: deterministic vs oracle gap %.1e, point RMSE %.1e, worst cuboid %.1e; atomics vs deterministic %s" %
          (gap, p_rmse, d_cb.max() if d_cb.size else 0.0, {k: "%.1e" % v for k, v in m0.items()}))
    so.close(); sh.close()


@pytest.mark.parametrize("cfg", ["cfg2", "cfg3", "cfg4"])
def test_baseline_configurations_match_the_oracle_at_full_size(oracle, hip, cfg):
    """BASELINE.json configs 2-4 at their full sizes (50 KF / 10k points; 100 KF / 30k points / 100 planes; 200 KF / 50k points /
    200 planes / 50 cuboids).  (The single-threaded oracle needs ~2 s for config 4.)"""
    _full_size_against_oracle(oracle, hip, synth.make_problem(cfg, seed=1), with_spread=(cfg == "cfg4"))


@pytest.mark.parametrize("kw", [dict(stereo_frac=0.2), dict(with_corners=True), dict(with_vp_planes=1000), dict(fixed_observer_frac=0.1),
                                dict(stereo_frac=0.2, with_corners=True, with_vp_planes=1000, fixed_observer_frac=0.1), dict(with_cuboid3d=True)],
                         ids=["stereo20", "corners", "vp1000", "fixedobs10", "all", "cuboid3d"])
def test_config4_edge_class_options_match_the_oracle_at_full_size(oracle, hip, kw):
    """The config-4 options of SURVEY.md section 8(d) at FULL size (they met the oracle only on 10-keyframe graphs before): 20 % stereo
    observations (multi-chunk StereoEdgeP segments, Optimizer.cc:2393-2422), the 16-D corner edges on all 500 cuboid observations
    (the second kernel variant, Optimizer.cc:2493-2551), 1 000 vertical + 1 000 parallel plane edges (Optimizer.cc:2262-2307),
    10 % additional fixed observers, and all of them together; and the 9-D camera-object edge of Optimizer::LocalBACameraPointCuboids2D on all
    500 cuboid observations (Optimizer.cc:1767-1800)."""
    _full_size_against_oracle(oracle, hip, synth.make_problem("cfg4", seed=1, **kw), with_spread=False)


ANALYTIC = dict(jacobian_mode=1)  # PPO_JAC_ANALYTIC


@pytest.mark.parametrize("cfg,kw", [c for c in CASES if c[0] != "tiny"][:6] + [("small", dict(seed=8, stereo_frac=0.2, with_corners=True, with_vp_planes=2))])
def test_analytic_jacobians_give_the_reference_result(oracle, hip, cfg, kw):
    """options.jacobian_mode = PPO_JAC_ANALYTIC (include/ppo_ba.h): exact derivatives for the plane and the
    cuboid-projection edges instead of the reference's delta = 1e-9 central differences.  Against the NUMERIC oracle: the first
    reduced system at the tolerance of the numeric Jacobians' own noise, then the whole protocol -- the same accept / reject sequence,
    chi2 per iteration to 1e-6, final states at BASELINE.md's bars, identical outlier sets."""
    P = synth.make_problem(cfg, **kw)
    so, sh = both(oracle, hip, P, hip_only=ANALYTIC, eliminate_cuboids=0)
    assert abs(so.linearize() - sh.linearize()) <= 1e-9 * max(1.0, abs(so.linearize()))
    lam = 7.5
    n = C.c_int32()
    oracle.lib.oracle_ba_reduced(so.h, lam, 0, -1, 1, None, None, C.byref(n))
    So, bo = np.zeros((n.value, n.value)), np.zeros(n.value)
    oracle.lib.oracle_ba_reduced(so.h, lam, 0, -1, 1, dptr(So), dptr(bo), None)
    hip.lib.ppo_ba_debug_set_lambda(sh.h, C.c_double(lam))
    Sh, bh = np.zeros((n.value, n.value)), np.zeros(n.value)
    assert hip.lib.ppo_ba_debug_reduced(sh.h, dptr(Sh), dptr(bh), None, None) == 0
    assert np.abs(So - Sh).max() <= 2e-6 * np.abs(So).max() and np.abs(bo - bh).max() <= 2e-6 * np.abs(bo).max()
    so.close(); sh.close()
    so, sh = both(oracle, hip, P, hip_only=ANALYTIC)
    to, th = so.run(), sh.run()
    assert [(i["pass_"], i["iteration"], i["trials"], i["accepted"], i["result"]) for i in to["iters"]] == \
           [(i["pass_"], i["iteration"], i["trials"], i["accepted"], i["result"]) for i in th["iters"]]
    for a, b in zip(to["iters"], th["iters"]):
        assert abs(a["chi2_after"] - b["chi2_after"]) <= 1e-6 * abs(a["chi2_after"])
    ro, rh = so.read(), sh.read()
    compare_states(ro, rh)
    assert np.array_equal(ro.pl_level, rh.pl_level) and np.array_equal(ro.bbox_level, rh.bbox_level) and np.array_equal(ro.corner_level, rh.corner_level)
    compare_levels(ro.mono_level, rh.mono_level, ro.mono_chi2, B.default_options().chi2_mono)
    so.close(); sh.close()


@pytest.mark.parametrize("cfg,kw", [("cfg3", {}), ("cfg4", {}), ("cfg4", dict(with_corners=True, with_vp_planes=1000))], ids=["cfg3", "cfg4", "cfg4-corners-vp"])
def test_analytic_jacobians_at_full_size(oracle, hip, cfg, kw):
    """The analytic mode at the full BASELINE sizes against the (numeric) oracle, by the same rules as the numeric mode."""
    _full_size_against_oracle(oracle, hip, synth.make_problem(cfg, seed=1, **kw), with_spread=False, hip_only=ANALYTIC)


def test_analytic_mode_refuses_a_fix_rotation_cuboid(hip):
    P = synth.make_problem("small", seed=1)
    f = P["cuboid_flags"].copy()
    f[0] = 4 | 1
    P.set("cuboid_flags", f)
    with pytest.raises(RuntimeError):
        hip.create(P, B.default_options(jacobian_mode=1))
    hip.create(P, B.default_options()).close()  # the numeric mode takes it


def test_product_library_ignores_the_development_switches(hip, monkeypatch):
    """libppo_ba.so is built without PPO_BA_TEST_HOOKS: environment switches and the loopback comm_id prefix change nothing."""
    P = synth.make_problem("small", seed=1)
    s = hip.create(P, B.default_options())
    ref = s.run()
    s.close()
    for k in ("PPO_BA_LIN_PARTS", "PPO_BA_CHI2_PASS", "PPO_BA_COMM_SELFTEST", "PPO_BA_CHOL_ONELAUNCH"):
        monkeypatch.setenv(k, "1")
    s = hip.create(P, B.default_options())
    st = s.run()
    s.close()
    assert [(i["trials"], i["accepted"]) for i in st["iters"]] == [(i["trials"], i["accepted"]) for i in ref["iters"]]
    assert abs(st["iters"][-1]["chi2_after"] - ref["iters"][-1]["chi2_after"]) <= 1e-7 * ref["iters"][-1]["chi2_after"]
    o = B.default_options(rank=0, world_size=2)
    for i, b in enumerate(b"PPOLOOP0" + (77).to_bytes(8, "little")):
        o.comm_id[i] = b
    with pytest.raises(Exception):  # a two-rank communicator with a made-up id: RCCL refuses it, nothing replaces RCCL
        hip.create(P, o)
    for bad in (dict(rank=2, world_size=2), dict(rank=-1, world_size=1), dict(rank=0, world_size=0)):
        with pytest.raises(Exception):
            hip.create(P, B.default_options(**bad))


def test_stop_flag_is_a_collective_decision_in_sharded_runs(hip_hooks):
    """Two ranks (loopback communicator), the stop flag raised on ONE of them only, before the run and in the middle of it: both
    ranks must return PPO_BA_ABORTED after the same number of LM iterations -- nobody is left waiting in a collective
    (the ranks vote: every exchange of the chi2 / scale scalars carries each rank's reading of its flag, summed)."""
    import threading
    import time
    P = synth.make_problem("cfg3", seed=1)  # ~0.25 ms per LM iteration: the 5 + 10 iterations leave time to raise the flag in the middle
    full = None
    for when in ("never", "before", "during"):
        world, key = 2, 4242 + len(when)
        flags = [np.zeros(1, np.uint8) for _ in range(world)]
        out, err = [None] * world, [None] * world
        started = [threading.Event() for _ in range(world)]

        def worker(rank):
            try:
                o = B.default_options(rank=rank, world_size=world)
                for i, b in enumerate(b"PPOLOOP0" + int(key).to_bytes(8, "little")):
                    o.comm_id[i] = b
                s = hip_hooks.create(P, o)
                started[rank].set()
                st = s.run(stop_flag=flags[rank])
                out[rank] = (s.last_rc, len(st["iters"]))
                s.close()
            except Exception as e:  # noqa: BLE001
                started[rank].set()
                err[rank] = e
        if when == "before":
            flags[1][0] = 1
        th = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
        for t in th:
            t.start()
        if when == "during":
            for ev in started:
                ev.wait(timeout=60)
            time.sleep(0.0015)
            flags[1][0] = 1
        for t in th:
            t.join(timeout=120)
            assert not t.is_alive(), "a rank is stuck in a collective"
        for e in err:
            if e is not None:
                raise e
        assert out[0] == out[1], (when, out)
        if when == "never":
            full = out[0][1]
            assert out[0][0] == 0 and full >= 10
        elif when == "before":
            assert out[0] == (1, 0), out  # PPO_BA_ABORTED before the first iteration
        else:
            # (a run that finished before the flag could be raised is not a failure of the library: only the agreement is asserted)
            assert out[0][0] in (0, 1) and (out[0][0] == 0 or out[0][1] < full), (out, full)


@pytest.mark.parametrize("cfg,kw", [("small", dict(seed=3, with_corners=True, with_vp_planes=2, stereo_frac=0.3)), ("cfg3", dict(seed=1))])
def test_deterministic_mode_is_bit_reproducible_and_inside_the_bar(oracle, hip, cfg, kw):
    """options.deterministic = 1: every cross-wave sum is an integer atomic on a 2^-24 fixed-point image of the addend, so
    the result cannot depend on the order in which workgroups arrive.  Three runs (two handles, a reset in between) must agree BIT
    FOR BIT in every output -- poses, landmarks, per-edge chi2, the chi2 / lambda trace -- and sit inside the same parity bar
    against the oracle as the floating-point-atomics mode, from which they differ only by that rounding."""
    P = synth.make_problem(cfg, **kw)
    outs = []
    for rep in range(2):
        s = hip.create(P, B.default_options(deterministic=1))
        for again in range(2 if rep == 0 else 1):
            if again:
                s.reset()
            st, r = s.run(), s.read()
            outs.append((st, r))
        s.close()
    st0, r0 = outs[0]
    for st, r in outs[1:]:
        assert [(i["trials"], i["accepted"], i["chi2_before"], i["chi2_after"], i["lambda_"]) for i in st["iters"]] == \
               [(i["trials"], i["accepted"], i["chi2_before"], i["chi2_after"], i["lambda_"]) for i in st0["iters"]]
        for name in ("pose_qt", "point_xyz", "plane_coef", "cuboid_qt", "cuboid_scale", "mono_chi2", "stereo_chi2", "pl_chi2", "bbox_chi2", "corner_chi2",
                     "mono_level", "pl_level"):
            assert np.array_equal(getattr(r, name), getattr(r0, name)), name
    so = oracle.create(P, B.default_options())
    to, ro = so.run(), so.read()
    so.close()
    assert [(i["trials"], i["accepted"]) for i in to["iters"]] == [(i["trials"], i["accepted"]) for i in st0["iters"]]
    assert abs(to["iters"][-1]["chi2_after"] - st0["iters"][-1]["chi2_after"]) <= 1e-6 * to["iters"][-1]["chi2_after"]
    compare_states(ro, r0)
    # the floating-point-atomics mode lands on the same answer to the bar
    sd = hip.create(P, B.default_options(**ATOMICS))
    sd.run()
    compare_states(sd.read(), r0)
    sd.close()


def test_cpp_caller_through_the_c_abi(hip, tmp_path):
    """examples/replay.cpp (plain g++, links libppo_ba.so) replays a dumped window: the boundary as a C++ host would use it."""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "replay")
    if not os.path.exists(exe):
        pytest.skip("examples/replay not built (run __graft_entry__.build())")
    P = synth.make_problem("small", seed=1)
    path = tmp_path / "w.ppoba"
    P.save(path)
    out = subprocess.run([exe, str(path), "2"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    s = hip.create(P, B.default_options())
    st = s.run()
    s.close()
    lines = [l for l in out.stdout.splitlines() if l.startswith("  pass")]
    assert len(lines) == len(st["iters"])
    last = float(lines[-1].split("->")[1].split()[0])
    assert abs(last - st["iters"][-1]["chi2_after"]) <= 1e-5 * st["iters"][-1]["chi2_after"]
    assert "call 1:" in out.stdout


def test_linearisation_standing_in_for_the_residual_pass_changes_nothing(hip):
    """By default the speculated linearisation of a trial state supplies that trial's chi2; PPO_BA_CHI2_PASS=1
    keeps the separate residual-only launch.  Both orders must walk the same LM schedule to the same estimates (the switch
    is read once per process, hence the child processes)."""
    import json
    import subprocess
    import sys
    code = (
        "import sys, json, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from ppo_ba import binding as B, synth\n"
        "P = synth.make_problem('small', seed=3, outlier_frac=0.2)\n"
        "s = B.hip_backend(test_hooks=True).create(P, B.default_options())\n"
        "t = s.run(); r = s.read()\n"
        "print(json.dumps({'iters': [[i['trials'], i['accepted'], i['chi2_after']] for i in t['iters']],\n"
        "                  'pose': r.pose_qt.tolist(), 'lvl': int(r.mono_level.sum())}))\n" % os.path.join(B.PKG_ROOT))
    out = {}
    for flag in ("0", "1"):
        env = dict(os.environ, PPO_BA_CHI2_PASS=flag)
        res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stderr[-2000:]
        out[flag] = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    a, b = out["0"], out["1"]
    assert len(a["iters"]) > 0 and len(b["iters"]) > 0
    # Two runs of the SAME order differ by the atomic summation order (chi2 ~1e-9 relative), which can flip the accept test of a
    # trial at convergence (delta chi2 ~1e-13 relative): the schedule is compared up to the first such
    # flip, the results everywhere.
    for x, y in zip(a["iters"], b["iters"]):
        if x[:2] != y[:2]:
            break
        assert abs(x[2] - y[2]) <= 3e-6 * abs(y[2])  # (the per-iteration chi2 bar of the parity tests: the delta = 1e-9 Jacobians amplify summation-order noise)
    assert abs(a["iters"][-1][2] - b["iters"][-1][2]) <= 1e-6 * abs(b["iters"][-1][2])
    assert abs(a["lvl"] - b["lvl"]) <= 2
    assert np.abs(np.array(a["pose"]) - np.array(b["pose"])).max() < 1e-6


def test_the_order_of_the_callers_edge_list_does_not_matter(hip):
    """ppo_ba_create sorts the reprojection edges by point itself.  A list that arrives sorted (the reference adds the observations
    point by point, Optimizer.cc:2350-2422) takes a short cut -- identity permutation, offsets from the key changes; a shuffled
    list takes the counting sort.  Both must build the same graph: same iteration history, same estimates (to the round-off the
    order of the edges inside a landmark leaves), and the per-edge outputs come back in the CALLER's order either way."""
    P = synth.make_problem("cfg3", seed=4, stereo_frac=0.2)
    n_mono, n_stereo = P.count("n_mono"), P.count("n_stereo")
    assert n_mono >= 1 << 16 and np.all(np.diff(P["mono_point"]) >= 0)  # the sorted short cut applies to this one
    rng = np.random.default_rng(5)
    pm, ps = rng.permutation(n_mono), rng.permutation(n_stereo)
    Q = B.Problem(**P.arrays)
    for name, w in (("mono_point", 1), ("mono_pose", 1), ("mono_obs", 2), ("mono_info", 1), ("mono_huber", 1)):
        Q.set(name, P[name].reshape(n_mono, w)[pm].reshape(-1))
    for name, w in (("stereo_point", 1), ("stereo_pose", 1), ("stereo_obs", 3), ("stereo_info", 1), ("stereo_huber", 1)):
        Q.set(name, P[name].reshape(n_stereo, w)[ps].reshape(-1))
    sp, sq = hip.create(P, B.default_options()), hip.create(Q, B.default_options())
    a, b = sp.run(), sq.run()
    rp, rq = sp.read(), sq.read()
    assert [it["trials"] for it in a["iters"]] == [it["trials"] for it in b["iters"]]
    assert abs(a["iters"][-1]["chi2_after"] - b["iters"][-1]["chi2_after"]) <= 1e-7 * a["iters"][-1]["chi2_after"]
    np.testing.assert_allclose(rp.pose_qt, rq.pose_qt, atol=2e-6)
    dp = np.linalg.norm(rp.point_xyz - rq.point_xyz, axis=1)  # (two-view points amplify round-off along the ray: RMSE bar, loose maximum)
    assert np.sqrt(np.mean(dp ** 2)) < 1e-5 and dp.max() < 1e-4, (np.sqrt(np.mean(dp ** 2)), dp.max())
    # (the estimates differ by the round-off above, and a residual of 0.1 px moves by 1e-3 relative when its point moves by 1e-6 m)
    np.testing.assert_allclose(rp.mono_chi2[pm], rq.mono_chi2, rtol=2e-3, atol=1e-4)
    np.testing.assert_allclose(rp.stereo_chi2[ps], rq.stereo_chi2, rtol=2e-3, atol=1e-4)
    assert np.mean(rp.mono_level[pm] != rq.mono_level) < 1e-4 and np.mean(rp.stereo_level[ps] != rq.stereo_level) < 1e-4
    sp.close(); sq.close()


def test_device_built_plans_equal_the_host_built_ones(hip_hooks, monkeypatch):
    """ppo_ba_create builds the reprojection edge records, the pose-major copy and the Schur pair plan on the device when the edge
    lists arrive sorted by point (csrc/ppo_plan.hpp), on the host otherwise.  Same window through both builders (the test-hooks
    library takes PPO_BA_HOST_PLAN=1), deterministic mode -- every cross-wave sum is an exact integer sum there, so the two runs are
    BIT-IDENTICAL exactly when both builders produced the same records, slots and pairs -- and once more in the floating-point-atomics
    mode, where the order inside a chunk may show as round-off."""
    P = synth.make_problem("cfg3", seed=3, stereo_frac=0.25, with_corners=True, fixed_observer_frac=0.1)
    assert P.count("n_mono") + P.count("n_stereo") >= 40000
    for det in (1, 0):
        res = []
        for host in ("", "1"):
            if host:
                monkeypatch.setenv("PPO_BA_HOST_PLAN", host)
            else:
                monkeypatch.delenv("PPO_BA_HOST_PLAN", raising=False)
            s = hip_hooks.create(P, B.default_options(deterministic=det))
            st = s.run()
            res.append((st, s.read()))
            s.close()
        (sa, ra), (sb, rb) = res
        assert [(i["trials"], i["accepted"]) for i in sa["iters"]] == [(i["trials"], i["accepted"]) for i in sb["iters"]]
        if det:
            assert [i["chi2_after"] for i in sa["iters"]] == [i["chi2_after"] for i in sb["iters"]]
            for name in ("pose_qt", "point_xyz", "plane_coef", "cuboid_qt", "cuboid_scale", "mono_chi2", "stereo_chi2", "mono_level", "stereo_level"):
                assert np.array_equal(getattr(ra, name), getattr(rb, name)), name
        else:
            np.testing.assert_allclose(ra.pose_qt, rb.pose_qt, atol=2e-6)
            assert abs(sa["iters"][-1]["chi2_after"] - sb["iters"][-1]["chi2_after"]) <= 1e-7 * sa["iters"][-1]["chi2_after"]


@pytest.mark.parametrize("det", [1, 0], ids=["deterministic", "atomics"])
@pytest.mark.parametrize("cfg,kw", [("cfg4", dict(seed=1)), ("cfg3", dict(seed=3, stereo_frac=0.25, with_corners=True, fixed_observer_frac=0.1))])
def test_overlapped_reduction_equals_the_plain_one(hip_hooks, monkeypatch, cfg, kw, det):
    """The Schur pair reduction is OVERLAPPED with the dense factorisation (csrc/ppo_chol.hip, StepOverlap): only the pair chunks of the
    first column blocks of S are reduced ahead of it, the launches of the first panels carry the rest as extra workgroups, trailing tiles
    those may be adding to take their update as atomic adds, the right-hand side is solved in place.  Sums in another order, nothing else:
    the plain reduction (test-hooks library, PPO_BA_OVERLAP=0) and three schedules -- the default, one column block per launch, and a
    late start with large slices -- must walk the same LM history and end in the same state to round-off.
    Deterministic mode (round 6): the pair workgroups add fixed-point integers to a second array, every column block is folded into the factor's own
    array by the first launch that finds it final (csrc/ppo_chol.hip: StepFix) -- each schedule is bit-reproducible, and they differ among each
    other only in where the converted sums meet the trailing updates (round-off of the same size as in the floating-point mode)."""
    P = synth.make_problem(cfg, **kw)
    res = []
    for ov in ("0", None, "1,0.0,0.0,0,2", "3,0.4,0.4,0,2"):
        if ov is None:
            monkeypatch.delenv("PPO_BA_OVERLAP", raising=False)
        else:
            monkeypatch.setenv("PPO_BA_OVERLAP", ov)
        s = hip_hooks.create(P, B.default_options(deterministic=det))
        st = s.run()
        res.append((st, s.read()))
        s.close()
    (sa, ra) = res[0]
    for (sb, rb) in res[1:]:
        assert [(i["trials"], i["accepted"]) for i in sa["iters"]] == [(i["trials"], i["accepted"]) for i in sb["iters"]]
        # (tools/overlap_diff.py: two PLAIN runs of config 4 differ by 1e-11 in the first iterations and by 3e-9 in the last ones -- floating-point
        #  atomics, amplified along the flat directions of the window; the overlapped schedules stay within a few times that: 1e-10, 3e-8)
        for k, (ia, ib) in enumerate(zip(sa["iters"], sb["iters"])):
            assert abs(ia["chi2_after"] - ib["chi2_after"]) <= (2e-9 if k < 5 else 2e-7) * ia["chi2_after"], k
        np.testing.assert_allclose(ra.pose_qt, rb.pose_qt, atol=2e-7)
        d = np.linalg.norm(ra.point_xyz - rb.point_xyz, axis=1)  # (a few landmarks left with one active observation move by 1e-5: flat directions)
        assert np.sqrt(np.mean(d ** 2)) < 1e-6 and np.quantile(d, 0.99) < 1e-5  # (a tenth of BASELINE.md's bars against the oracle)
        assert np.mean(ra.mono_level != rb.mono_level) < 1e-4


def _reduced_solution(hip_hooks, solver, with_system=False):
    lib = hip_hooks.lib
    n = C.c_int32()
    lib.ppo_ba_debug_set_lambda.argtypes = [C.c_void_p, C.c_double]
    assert lib.ppo_ba_debug_set_lambda(solver.h, 10.0) == 0  # (of the order of the first lambda of these windows: 1e-5 max diag H)
    assert lib.ppo_ba_debug_reduced(solver.h, None, None, None, C.byref(n)) == 0
    x = np.zeros(n.value)
    if not with_system:
        assert lib.ppo_ba_debug_reduced(solver.h, None, None, x.ctypes.data_as(B.c_dp), None) == 0
        return x
    S = np.zeros((n.value, n.value))
    b = np.zeros(n.value)
    assert lib.ppo_ba_debug_reduced(solver.h, S.ctypes.data_as(B.c_dp), b.ctypes.data_as(B.c_dp), x.ctypes.data_as(B.c_dp), None) == 0
    return x, S, b


@pytest.mark.parametrize("det", [1, 0], ids=["deterministic", "atomics"])
def test_overlapped_reduction_solves_the_same_system_on_randomly_structured_windows(hip_hooks, monkeypatch, det):
    """The sharp form of the test above, without fifteen LM iterations amplifying round-off: ONE reduced system, solved through the plain
    reduction and through the overlapped one (test hook PPO_BA_DEBUG_OVERLAP), on windows of random size and composition that the
    overlapped path would not normally take (two to six 64-column blocks; fixed points, fixed observers, stereo, planes with vertical /
    parallel edges, cuboids with corners, empty classes, both elimination modes -- tools/random_sweep.py) and on configurations 3 and 4:
    the pose updates agree to 1e-9 of their norm in every schedule (floating-point sums); in deterministic mode every schedule solves a system within
    the rounding of the fixed-point grid of the plain one (bound derived below) and the pose updates agree to 2e-7."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("random_sweep", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "random_sweep.py"))
    rs = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rs)
    cases = []
    for seed in range(5100, 5130):
        cfg, kw, elim = rs.random_case(seed)
        cfg["n_kf"] += 22 + (seed % 3) * 10  # (25 .. 65 keyframes)
        cfg["n_points"] *= 3
        cases.append((cfg, kw, elim))
    cases += [("cfg3", dict(seed=3, stereo_frac=0.25, with_corners=True, fixed_observer_frac=0.1), 1), ("cfg4", dict(seed=1), 1), ("cfg4", dict(seed=2), 0)]
    worst = worst_fill = 0.0
    for cfg, kw, elim in cases:
        P = synth.make_problem(cfg, **kw)
        xs = []
        for ov, dbg in (("0", None), ("1,0.0,0.0,0,2", "1"), ("1,0.14,0.10,0,2", "1"), ("2,0.5,0.3,0,2", "1")):
            monkeypatch.setenv("PPO_BA_OVERLAP", ov)
            if dbg:
                monkeypatch.setenv("PPO_BA_DEBUG_OVERLAP", dbg)
            else:
                monkeypatch.delenv("PPO_BA_DEBUG_OVERLAP", raising=False)
            s = hip_hooks.create(P, B.default_options(eliminate_cuboids=elim, deterministic=det))
            if dbg is None and det:
                x0, S0, b0 = _reduced_solution(hip_hooks, s, with_system=True)
                xs.append(x0)
            else:
                xs.append(_reduced_solution(hip_hooks, s))
            s.close()
        assert np.all(np.isfinite(xs[0])) and np.linalg.norm(xs[0]) > 0
        for x in xs[1:]:
            dev = np.linalg.norm(x - xs[0]) / np.linalg.norm(xs[0])
            worst = max(worst, dev)
            if det:
                # Deterministic mode: a trailing update that reaches a column block while that block is still OPEN (the reduction has not finished it) is added to
                # the block's 2^-24 fixed-point image like every other term there, i.e. rounded to that grid by at most half a step; a block takes at most one such
                # update per panel launch before it, and which blocks are open when is what the schedule decides.  So every overlapped schedule solves EXACTLY
                # (up to the floating-point round-off of a Cholesky) a system (S + dS) x = b + db with |dS_ij|, |db_i| <= panels * 2^-25 -- checked here as a
                # bound on the residual of its x against the plain schedule's S and b, row by row: |S x - b|_i <= eps (|x|_1 + 1) + round-off.
                n = S0.shape[0]
                panels = (n + 63) // 64
                eps = panels * 2.0 ** -25
                resid = np.abs(S0 @ x - b0).max()
                roundoff = 8 * n * np.finfo(float).eps * np.abs(S0).max() * np.abs(x).max()
                assert resid <= eps * (np.abs(x).sum() + 1.0) + roundoff, (cfg, kw, elim, resid, eps * (np.abs(x).sum() + 1.0), roundoff)
                worst_fill = max(worst_fill, resid / (eps * (np.abs(x).sum() + 1.0) + roundoff))
                # ... and what that does to the pose update (x' - x = S^-1 (S x' - b): the residual above times the conditioning of the window): with lambda = 10
                # on a diagonal whose smallest entries are ~10, one grid step is 6e-9 of them.  Worst over these 33 windows x 3 schedules: 6.0e-8 of the update's
                # norm (configuration 4 seed 2 with the cuboids kept in the dense system, 19 panels; 2.3e-8 on the random windows) -- the mode is bit-reproducible,
                # so these numbers are the same in every run.  The noise of the delta = 1e-9 Jacobians that made the system is 1e-7.
                assert dev < 2e-7, (cfg, kw, elim, dev)
            else:
                assert dev < 1e-9, (cfg, kw, elim, dev)
    if det:
        print("worst residual / its bound: %.2e" % worst_fill)
    print("worst relative deviation of the pose update: %.1e" % worst)


def test_device_builder_on_randomly_structured_windows(hip_hooks, monkeypatch):
    """The device-side builder of slots, records and plans (csrc/ppo_plan.hpp) normally takes only large windows; the test-hooks library
    lets it take everything (PPO_BA_DEV_PLAN_MIN=0).  40 windows of random size and composition -- fixed points, fixed observers,
    stereo share, planes with vertical / parallel edges, cuboids with corners, empty classes, both elimination modes
    (tools/random_sweep.py) -- through both builders in deterministic mode: BIT-identical iteration histories and estimates, i.e. the
    same slots, records and pairs."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("random_sweep", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "random_sweep.py"))
    rs = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rs)
    for seed in range(5000, 5040):
        cfg, kw, elim = rs.random_case(seed)
        P = synth.make_problem(cfg, **kw)
        out = []
        for host in (False, True):
            monkeypatch.setenv("PPO_BA_DEV_PLAN_MIN", "0")
            if host:
                monkeypatch.setenv("PPO_BA_HOST_PLAN", "1")
            else:
                monkeypatch.delenv("PPO_BA_HOST_PLAN", raising=False)
            s = hip_hooks.create(P, B.default_options(deterministic=1, eliminate_cuboids=elim))
            st = s.run()
            out.append((st, s.read()))
            s.close()
        (sa, ra), (sb, rb) = out
        assert [(i["trials"], i["accepted"], i["chi2_after"]) for i in sa["iters"]] == [(i["trials"], i["accepted"], i["chi2_after"]) for i in sb["iters"]], (seed, cfg, kw)
        for name in ("pose_qt", "point_xyz", "plane_coef", "cuboid_qt", "cuboid_scale", "mono_chi2", "stereo_chi2", "pl_chi2", "mono_level", "stereo_level", "pl_level"):
            assert np.array_equal(getattr(ra, name), getattr(rb, name)), (seed, name, cfg, kw)


@pytest.mark.parametrize("world", [2, 3])
def test_landmark_sharding_with_the_device_builder(hip_hooks, monkeypatch, world):
    """Sharded ranks whose plans come from the device-side builder (csrc/ppo_plan.hpp: pose-major copies and pair plan of the OWNED
    landmark range only, slots of all landmarks): same answer as one rank, and as the host builder, on a window with planes, cuboids,
    stereo edges and fixed observers."""
    monkeypatch.setenv("PPO_BA_DEV_PLAN_MIN", "0")
    P = synth.make_problem(dict(n_kf=14, n_points=500, n_planes=5, n_cuboids=3), seed=9, stereo_frac=0.3, fixed_observer_frac=0.2)
    s = hip_hooks.create(P, B.default_options())
    st1, r1 = s.run(), s.read()
    s.close()
    res = _run_sharded_in_process(hip_hooks, P, world, key=4000 + world)
    for st, r in res:
        assert [(i["trials"], i["accepted"], i["result"]) for i in st["iters"]] == [(i["trials"], i["accepted"], i["result"]) for i in st1["iters"]]
        compare_states(r1, r)
        assert np.array_equal(r.mono_level, r1.mono_level) and np.array_equal(r.stereo_level, r1.stereo_level) and np.array_equal(r.pl_level, r1.pl_level)
    monkeypatch.setenv("PPO_BA_HOST_PLAN", "1")
    res_h = _run_sharded_in_process(hip_hooks, P, world, key=4100 + world, deterministic=1)
    monkeypatch.delenv("PPO_BA_HOST_PLAN")
    res_d = _run_sharded_in_process(hip_hooks, P, world, key=4200 + world, deterministic=1)
    for (sa, ra), (sb, rb) in zip(res_h, res_d):  # deterministic mode: the two builders give bit-identical sharded runs
        assert [i["chi2_after"] for i in sa["iters"]] == [i["chi2_after"] for i in sb["iters"]]
        assert np.array_equal(ra.pose_qt, rb.pose_qt) and np.array_equal(ra.point_xyz, rb.point_xyz) and np.array_equal(ra.mono_chi2, rb.mono_chi2)
