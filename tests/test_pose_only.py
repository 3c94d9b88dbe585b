"""Pose-only optimisation (Optimizer::PoseOptimization, reference src/Optimizer.cc:247-459).

CPU tests: the oracle's unary OnlyPose edges against golden vectors made by the reference's own prebuilt libg2o.so
(EdgeSE3ProjectXYZOnlyPose / EdgeStereoSE3ProjectXYZOnlyPose computeError + linearizeOplus, oracle/pin_g2o_prebuilt.py), known
answers of the oracle's protocol, the committed trace, and the boundary (symbols, defaults, loud failure without a GPU).
GPU tests (-m gpu): the HIP library through the C ABI against the oracle.

Tolerances: pose translation 1e-7 m and rotation 1e-7 rad (the bar of BASELINE.json is 1e-5 m / 1e-6 rad); the measured
difference is ~1e-14 because the analytic Jacobians carry no 1e-9 differencing noise.  At convergence the accept / reject
decision of a trial is decided by rounding (chi2 changes by 1e-13 relative), so trial counts may differ while poses,
outlier sets and chi2 do not: the schedule is therefore not compared.
"""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

from conftest import dptr
from ppo_ba import binding as B
from ppo_ba import pose

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
GOLD = json.load(open(os.path.join(HERE, "golden", "g2o_prebuilt.json")))["cases"]


def quat_dist(qa, qb):
    """Rotation angle between unit quaternions, accurate for tiny angles (2 * |vector part of qa^-1 qb|)."""
    qa, qb = np.asarray(qa), np.asarray(qb)
    if np.dot(qa, qb) < 0:
        qb = -qb
    return 2.0 * np.linalg.norm(qa - qb)


def frames_from_json(d):
    return pose.Frames(d["pose_qt"], d["cam"], d["edge_offsets"], d["xw"], d["obs"], d["inv_sigma2"])


# ---- CPU: oracle pinned against the prebuilt g2o ------------------------------------------------------------------------------
@pytest.mark.parametrize("kind", ["mono", "stereo"])
def test_oracle_onlypose_edges_match_prebuilt_g2o(oracle_lib, kind):
    oracle_lib.oracle_onlypose_eval.argtypes = [B.c_dp] * 4 + [C.c_int] + [B.c_dp] * 2
    oracle_lib.oracle_onlypose_eval.restype = None
    cases = GOLD["onlypose_" + kind]
    assert len(cases) == 24
    for c in cases:
        e, J = np.zeros(3), np.zeros(18)
        oracle_lib.oracle_onlypose_eval(dptr(np.array(c["qt"])), dptr(np.array(c["cam"])), dptr(np.array(c["xw"])), dptr(np.array(c["obs"])),
                                        int(kind == "stereo"), dptr(e), dptr(J))
        d = len(c["error"])
        # residuals are pixels of magnitude ~1e3: 1e-11 absolute is 1e-14 relative (Eigen's q*v and ours round differently)
        np.testing.assert_allclose(e[:d], c["error"], rtol=0, atol=1e-11)
        np.testing.assert_allclose(J[:6 * d], c["J"], rtol=1e-12, atol=1e-12)


# ---- CPU: known answers of the protocol -------------------------------------------------------------------------------------
def exact_frame(n, stereo_fraction, seed, n_gross=0):
    """Noise-free observations of float32 map points: the optimum is the true pose exactly (up to float32 obs rounding)."""
    fr = pose.make_frames(n_frames=1, n_obs=n, stereo_fraction=stereo_fraction, outlier_fraction=0.0, seed=seed)
    rng = np.random.default_rng(seed + 100)
    # rebuild the observations without noise from a known pose
    q = rng.normal(size=4); q /= np.linalg.norm(q); q = q if q[3] >= 0 else -q
    x, y, z, w = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    t = rng.normal(size=3)
    pc = np.stack([rng.uniform(-1, 1, n), rng.uniform(-0.8, 0.8, n), np.ones(n)], 1) * rng.uniform(0.5, 8.0, (n, 1))
    Xw = ((pc - t) @ R).astype(np.float32).astype(np.float64)
    pcf = Xw @ R.T + t
    fx, fy, cx, cy, bf = fr.cam[0]
    u = fx * pcf[:, 0] / pcf[:, 2] + cx
    v = fy * pcf[:, 1] / pcf[:, 2] + cy
    ur = np.where(fr.obs[:, 2] >= 0, u - bf / pcf[:, 2], -1.0)
    obs = np.stack([u, v, ur], 1)
    if n_gross:
        obs[:n_gross, 0] += 50.0  # 50 px off: far outside any chi2 gate
    truth = np.concatenate([q, t])
    start = truth.copy()
    start[4:] += [0.03, -0.02, 0.04]
    return pose.Frames([start], fr.cam, fr.edge_offsets, Xw, obs, np.ones(n)), truth


def test_oracle_recovers_the_true_pose_from_exact_observations(oracle_lib):
    fr, truth = exact_frame(120, 0.5, seed=3)
    out = pose.optimize(oracle_lib, fr, prefix="oracle_")
    assert np.abs(out["pose_qt"][0, 4:] - truth[4:]).max() < 1e-8 and quat_dist(out["pose_qt"][0, :4], truth[:4]) < 1e-8
    assert out["n_inliers"][0] == 120 and out["outlier"].sum() == 0 and out["stats"][0]["rounds_run"] == 4
    assert out["chi2"].max() < 1e-7


def test_oracle_flags_gross_mismatches_and_still_recovers_the_pose(oracle_lib):
    fr, truth = exact_frame(200, 0.3, seed=4, n_gross=20)
    out = pose.optimize(oracle_lib, fr, prefix="oracle_")
    assert out["outlier"][:20].all() and not out["outlier"][20:].any() and out["n_inliers"][0] == 180
    assert np.abs(out["pose_qt"][0, 4:] - truth[4:]).max() < 1e-7  # the last round runs without the outliers and without the kernel


def test_oracle_small_frame_rules(oracle_lib):
    # Optimizer.cc:372-373: fewer than 3 correspondences -> return 0, pose untouched; :448-449: fewer than 10 edges -> one round only
    fr = pose.make_frames(n_frames=3, n_obs=[2, 9, 10], seed=7)
    out = pose.optimize(oracle_lib, fr, prefix="oracle_")
    assert out["n_inliers"][0] == 0 and np.array_equal(out["pose_qt"][0], fr.pose_qt[0]) and out["stats"][0]["rounds_run"] == 0
    assert out["stats"][1]["rounds_run"] == 1 and out["stats"][2]["rounds_run"] == 4
    assert not np.array_equal(out["pose_qt"][1], fr.pose_qt[1])


def test_oracle_reproduces_its_committed_pose_only_trace(oracle_lib):
    doc = json.load(open(os.path.join(HERE, "golden", "pose_only_small.json")))
    out = pose.optimize(oracle_lib, frames_from_json(doc["inputs"]), prefix="oracle_")
    exp = doc["expected"]
    np.testing.assert_allclose(out["pose_qt"], exp["pose_qt"], rtol=0, atol=1e-10)
    assert out["outlier"].tolist() == exp["outlier"] and out["n_inliers"].tolist() == exp["n_inliers"]
    np.testing.assert_allclose(out["chi2"], exp["chi2"], rtol=1e-8, atol=1e-9)
    assert [s["rounds_run"] for s in out["stats"]] == exp["rounds_run"]


# ---- CPU: boundary ------------------------------------------------------------------------------------------------------------
def test_pose_header_symbols_exported_and_defaults_match():
    lib = B.load_library()
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(REPO, "include", "ppo_pose.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(ppo_pose_[a-z_0-9]+)\s*\(", src)))
    assert names == ["ppo_pose_default_options", "ppo_pose_optimize"]
    for n in names:
        assert hasattr(lib, n), "libppo_ba.so does not export " + n
    o = pose.CPoseOptions()
    lib.ppo_pose_default_options.argtypes = [C.POINTER(pose.CPoseOptions)]
    lib.ppo_pose_default_options.restype = None
    lib.ppo_pose_default_options(C.byref(o))
    p = pose.default_options()
    for k, _ in pose.CPoseOptions._fields_:
        assert getattr(o, k) == getattr(p, k), k
    assert (o.rounds, o.iters_per_round, o.robust_rounds, o.min_correspondences, o.min_edges) == (4, 10, 3, 3, 10)
    assert o.chi2_mono == np.float32(5.991) and o.huber_stereo == np.float32(np.sqrt(7.815))
    assert C.sizeof(pose.CPoseStats) == 4 * 4 + 3 * 8 and C.sizeof(pose.CPoseResult) == 5 * 8 and C.sizeof(pose.CPoseProblem) == 7 * 8


def test_pose_optimize_rejects_bad_arguments_and_has_no_cpu_path():
    import torch
    lib = B.load_library()
    fr = pose.make_frames(n_frames=1, n_obs=20, seed=1)
    with pytest.raises(RuntimeError, match="-1"):
        pose.optimize(lib, fr, pose.default_options(abi_version=99))
    bad = pose.make_frames(n_frames=1, n_obs=20, seed=1)
    bad.xw[3, 1] = np.nan
    with pytest.raises(RuntimeError, match="-2"):
        pose.optimize(lib, bad)
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="-4"):  # PPO_BA_ERR_HIP: fails loudly, there is no fallback
            pose.optimize(lib, fr)


# ---- GPU: parity through the C ABI --------------------------------------------------------------------------------------------
def compare(g, o, fr, opts):
    for f in range(fr.n_frames):
        assert np.abs(g["pose_qt"][f, 4:] - o["pose_qt"][f, 4:]).max() < 1e-7, f
        assert quat_dist(g["pose_qt"][f, :4], o["pose_qt"][f, :4]) < 1e-7, f
        assert g["stats"][f]["rounds_run"] == o["stats"][f]["rounds_run"]
    thr = np.where(fr.obs[:, 2] < 0, opts.chi2_mono, opts.chi2_stereo)
    near = np.abs(o["chi2"] - thr) < 1e-6 * thr  # float(chi2) > float threshold can flip only there
    assert np.array_equal(g["outlier"][~near], o["outlier"][~near])
    np.testing.assert_allclose(g["chi2"], o["chi2"], rtol=1e-7, atol=1e-9)
    if not near.any():
        assert np.array_equal(g["n_inliers"], o["n_inliers"])
    for f in range(fr.n_frames):
        e0, e1 = fr.edge_offsets[f], fr.edge_offsets[f + 1]
        expected = (e1 - e0) - int(g["outlier"][e0:e1].sum()) if g["stats"][f]["rounds_run"] else 0
        assert g["n_inliers"][f] == expected


CASES = {
    "mixed": dict(n_frames=1, n_obs=400, seed=1),
    "mono_only": dict(n_frames=1, n_obs=1000, stereo_fraction=0.0, seed=2),
    "stereo_only": dict(n_frames=1, n_obs=200, stereo_fraction=1.0, seed=3),
    "batch": dict(n_frames=64, n_obs=[300 + 10 * i for i in range(64)], seed=4),
    "ragged": dict(n_frames=7, n_obs=[0, 2, 3, 9, 10, 257, 2500], seed=5),
    "heavy_outliers": dict(n_frames=4, n_obs=500, outlier_fraction=0.5, seed=6, pose_noise=(0.05, 0.2)),
    "far_start": dict(n_frames=4, n_obs=300, seed=7, pose_noise=(0.15, 0.5)),  # rejected trials, lambda growth
}


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_pose_only_matches_the_oracle(oracle_lib, name):
    lib = B.load_library()
    fr = pose.make_frames(**CASES[name])
    opts = pose.default_options()
    compare(pose.optimize(lib, fr, opts), pose.optimize(oracle_lib, fr, opts, prefix="oracle_"), fr, opts)


@pytest.mark.gpu
def test_pose_only_non_default_protocol_matches_the_oracle(oracle_lib):
    lib = B.load_library()
    fr = pose.make_frames(n_frames=3, n_obs=350, seed=8)
    for over in (dict(rounds=2, iters_per_round=5, robust_rounds=1), dict(rounds=6, robust_rounds=0), dict(lm_max_trials=2), dict(chi2_mono=3.0, chi2_stereo=4.0)):
        opts = pose.default_options(**over)
        compare(pose.optimize(lib, fr, opts), pose.optimize(oracle_lib, fr, opts, prefix="oracle_"), fr, opts)


@pytest.mark.gpu
def test_pose_only_matches_the_committed_trace():
    doc = json.load(open(os.path.join(HERE, "golden", "pose_only_small.json")))
    out = pose.optimize(B.load_library(), frames_from_json(doc["inputs"]))
    np.testing.assert_allclose(out["pose_qt"], doc["expected"]["pose_qt"], rtol=0, atol=1e-8)
    assert out["outlier"].tolist() == doc["expected"]["outlier"] and out["n_inliers"].tolist() == doc["expected"]["n_inliers"]


@pytest.mark.gpu
def test_pose_only_is_bit_reproducible_and_batching_does_not_change_a_frame():
    # (same workgroup size in both launches: it is chosen from the largest frame of a call, 256 threads up to 700 observations)
    lib = B.load_library()
    fr = pose.make_frames(n_frames=16, n_obs=450, seed=9)
    a, b = pose.optimize(lib, fr), pose.optimize(lib, fr)
    assert np.array_equal(a["pose_qt"], b["pose_qt"]) and np.array_equal(a["chi2"], b["chi2"]) and np.array_equal(a["outlier"], b["outlier"])
    e0, e1 = fr.edge_offsets[5], fr.edge_offsets[6]
    one = pose.Frames(fr.pose_qt[5:6], fr.cam[5:6], [0, e1 - e0], fr.xw[e0:e1], fr.obs[e0:e1], fr.inv_sigma2[e0:e1])
    c = pose.optimize(lib, one)
    assert np.array_equal(c["pose_qt"][0], a["pose_qt"][5]) and np.array_equal(c["outlier"], a["outlier"][e0:e1])


@pytest.mark.gpu
def test_pose_only_recovers_the_true_pose_at_scale(oracle_lib):
    # size-independent property on a batch the oracle would need seconds for: exact observations -> the true pose, no outliers
    lib = B.load_library()
    frames, truths = zip(*[exact_frame(600, 0.5, seed=100 + i) for i in range(256)])
    off = np.concatenate([[0], np.cumsum([f.n_edges for f in frames])])
    big = pose.Frames(np.concatenate([f.pose_qt for f in frames]), np.concatenate([f.cam for f in frames]), off, np.concatenate([f.xw for f in frames]),
                      np.concatenate([f.obs for f in frames]), np.concatenate([f.inv_sigma2 for f in frames]))
    out = pose.optimize(lib, big)
    T = np.array(truths)
    assert np.abs(out["pose_qt"][:, 4:] - T[:, 4:]).max() < 1e-7
    assert max(quat_dist(out["pose_qt"][i, :4], T[i, :4]) for i in range(256)) < 1e-7
    assert out["outlier"].sum() == 0 and (out["n_inliers"] == 600).all()
    sub = [3, 77, 200]
    for i in sub:  # and a few of them against the oracle
        o = pose.optimize(oracle_lib, frames[i], prefix="oracle_")
        assert np.abs(o["pose_qt"][0] - out["pose_qt"][i]).max() < 1e-9


@pytest.mark.gpu
def test_pose_only_empty_batch_is_ok():
    fr = pose.Frames(np.zeros((0, 7)), np.zeros((0, 5)), [0], np.zeros((0, 3)), np.zeros((0, 3)), np.zeros(0))
    out = pose.optimize(B.load_library(), fr)
    assert out["pose_qt"].shape == (0, 7)


@pytest.mark.gpu
def test_pose_only_cpp_caller_through_the_c_abi():
    """examples/pose_only.cpp (plain g++ against libppo_ba.so): one frame per call like the reference, then the batch; it exits
    non-zero unless the batched poses are bit-identical to the per-frame ones."""
    import subprocess
    exe = os.path.join(REPO, "examples", "pose_only")
    if not os.path.exists(exe):
        pytest.skip("examples/pose_only not built (run __graft_entry__.build())")
    out = subprocess.run([exe, "300", "12"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "bit-identical" in out.stdout
    err = float(out.stdout.split("max |t - truth|")[1].split()[0])
    assert err < 0.02  # noisy observations: the optimised translation is within 2 cm of the truth
