"""Is a window outside the parity bar a bug of the HIP path, or is the reference's own result not defined to that accuracy there?

tests/golden/hip_vs_oracle_sweep_r01.json holds the 37 randomly structured windows (of 600) on which the HIP library and the
oracle disagreed beyond BASELINE.md's bars on an MI355X (round 1 final build, tools/random_sweep.py).  Here the ORACLE is
re-run on exactly those windows with nothing changed but floating-point rounding (tests/sensitivity.py: reversed summation
order, contraction to fused multiply-adds, both) and its own spread is compared with the recorded HIP-vs-oracle deviation.
If the oracle moved much less than the HIP path differed from it, the difference would be a kernel bug; it moves as much.

Why these windows are sensitive (asserted below where it can be): the plane / cuboid / point-cuboid edges are differentiated
numerically with delta = 1e-9 (base_binary_edge.hpp:217-320) -- tests/test_golden.py::test_reference_numeric_jacobian_noise_floor
measures 1e-4 relative noise on the REFERENCE BINARY's own numeric Jacobian -- and the culling between the passes leaves
landmarks with a single observation.  Points-only windows (analytic Jacobians) repeat to 1e-12.
"""
import importlib.util
import json
import os

import numpy as np
import pytest

import sensitivity as S
from ppo_ba import binding as B
from ppo_ba import synth

HERE = os.path.dirname(os.path.abspath(__file__))
FLAGGED = json.load(open(os.path.join(HERE, "golden", "hip_vs_oracle_sweep_r01.json")))["cases"]


def _random_sweep():
    spec = importlib.util.spec_from_file_location("random_sweep", os.path.join(os.path.dirname(HERE), "tools", "random_sweep.py"))
    rs = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rs)
    return rs


def test_oracle_self_sensitivity_on_the_windows_outside_the_bar(oracle):
    rs = _random_sweep()
    ratios, chatter, plateau = [], 0, 0
    for c in FLAGGED:
        cfg, kw, elim = rs.random_case(c["seed"])
        P = synth.make_problem(cfg, **kw)
        runs = S.oracle_runs(oracle, P, B.default_options(eliminate_cuboids=elim))
        oo, _, _ = S.spread(runs)
        ho = c["hip_vs_oracle"]
        out = [k for k in S.BARS if ho[k] >= S.BARS[k]]
        if not c["schedule_equal"]:
            # the HIP run took a different accept / reject decision somewhere.  Either the oracle's own variants flip too, or the
            # window sits on a convergence plateau where a trial's acceptance hangs on the last digits of chi2:
            flips = any(S.schedule(st) != S.schedule(runs[0][0]) for st, _ in runs[1:])
            it = runs[0][0]["iters"]
            on_plateau = any(a["trials"] > 1 and abs(a["chi2_before"] - a["chi2_after"]) < 1e-7 * a["chi2_before"] for a in it)
            assert flips or on_plateau, c["seed"]
            chatter += 1
            plateau += int(on_plateau and not flips)
            if not flips:
                continue  # the deviation comes from the extra / missing iteration, not from a summation order
        for k in out:
            ratios.append((oo[k] / ho[k], c["seed"], k))
    ratios.sort()
    r = np.array([x[0] for x in ratios])
    assert len(r) >= 35
    # every out-of-bar quantity: the oracle's own spread is at least a twentieth of the HIP-vs-oracle deviation, for nine in
    # ten at least a third, for the median it is LARGER (measured: min 0.11, median 1.4)
    assert r.min() >= 0.05, ratios[:3]
    assert np.quantile(r, 0.1) >= 0.3, ratios[:6]
    assert np.median(r) >= 1.0
    assert chatter <= 8


def test_wider_accumulation_moves_the_sensitive_windows_like_a_change_of_summation_order(oracle):
    """The other half of the question "summation order or arithmetic?": the oracle with its long sums carried in (effectively)
    twice the precision -- oracle_ba_set_variant(2), Neumaier-compensated accumulation of Hpp, b, S and the Schur right-hand side
    -- against the plain oracle, on the first flagged windows.  The exact-sum result is as far from the plain run as the reversed-
    order run is: the plain double-precision result carries that much rounding, whichever order it is summed in."""
    rs = _random_sweep()
    ratios = []
    for c in FLAGGED[:10]:
        cfg, kw, elim = rs.random_case(c["seed"])
        P = synth.make_problem(cfg, **kw)
        runs = S.oracle_runs(oracle, P, B.default_options(eliminate_cuboids=elim), variants=("reversed", "compensated"))
        if any(S.schedule(st) != S.schedule(runs[0][0]) for st, _ in runs[1:]):
            continue  # an accept / reject decision flipped: the runs are not comparable quantity by quantity
        rev, comp = S.metrics(runs[0][1], runs[1][1]), S.metrics(runs[0][1], runs[2][1])
        for k in S.BARS:
            if max(rev[k], comp[k]) > 1e-9:
                ratios.append(comp[k] / max(rev[k], 1e-300))
    r = np.array(ratios)
    assert len(r) >= 10
    assert 0.02 < np.median(r) < 50, np.sort(r)  # the same order of magnitude, not zero and not dominant


def test_points_only_windows_repeat_to_rounding(oracle):
    """Control: with analytic Jacobians only (BASELINE config 2 shape) the same re-runs agree to 1e-10 -- the summation order is
    not what moves the windows above; the delta = 1e-9 differencing of the plane / cuboid edges amplifies it."""
    P = synth.make_problem(dict(n_kf=12, n_points=400, n_planes=0, n_cuboids=0), seed=6)
    runs = S.oracle_runs(oracle, P, B.default_options(), variants=("reversed", "fma", "fma+reversed", "compensated"))
    m, _, _ = S.spread(runs)
    assert max(m.values()) < 1e-10, m
    assert all(S.schedule(st) == S.schedule(runs[0][0]) for st, _ in runs)


def test_config4_cuboids_are_not_defined_to_1e5_by_the_reference_algorithm(oracle):
    """BASELINE config 4 (200 keyframes, 50 000 points, 200 planes, 50 cuboids), seeds 1 and 2: poses, points and planes of the
    oracle's re-runs agree far inside the bars, but one to four of the fifty cuboids move by MORE than the 1e-5 bar (seed 1:
    3e-5, seed 2: 4e-4) -- flat directions of the cost seen through numerically differentiated bbox edges.  The GPU parity test
    at this size therefore compares each cuboid against the bar OR against that cuboid's own spread in the oracle
    (tests/test_parity_gpu.py::test_baseline_configurations_match_the_oracle_at_full_size)."""
    worst = 0.0
    for seed in (1, 2):
        P = synth.make_problem("cfg4", seed=seed)
        runs = S.oracle_runs(oracle, P, B.default_options(), variants=("reversed", "fma"))
        m, cub, _ = S.spread(runs)
        assert m["t"] < 1e-5 and m["r"] < 1e-6 and m["pl"] < 5e-6, (seed, m)
        assert all(S.schedule(st) == S.schedule(runs[0][0]) for st, _ in runs)
        assert 1 <= int((cub > 1e-5).sum()) <= 8, (seed, np.sort(cub)[-5:])
        worst = max(worst, float(cub.max()))
    assert worst > 1e-4
