"""Golden vectors (CPU tests).

tests/golden/g2o_prebuilt.json was produced by oracle/pin_g2o_prebuilt.py by CALLING THE REFERENCE'S OWN prebuilt
Thirdparty/g2o/lib/libg2o.so (SE3Quat::exp, VertexSE3Expmap::oplusImpl, RobustKernelHuber::robustify,
EdgeSE3ProjectXYZ::cam_project, EdgeStereoSE3ProjectXYZ::cam_project).  Two things are checked against it:
  * the oracle (oracle/libppo_oracle.so) -- this is what pins the restatement;
  * the product's device math (csrc/ppo_math.hpp, compiled for the host as libppo_hostcheck.so).
The vectors already caught two restatement errors that no analytic test would have found: `float dsqr` in
RobustKernelHuber (robust_kernel_impl.h:84) and `1.0f/z` being a double division rounded to float
(types_six_dof_expmap.cpp:183).

tests/golden/oracle_trace_small.json is a regression trace of the oracle itself (made by tests/golden/make_oracle_trace.py):
it freezes the LM schedule (lambda, chi2, trials per iteration) and the final estimates of one small graph.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

from conftest import dptr
from ppo_ba import binding as B
from ppo_ba import synth

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "g2o_prebuilt.json")))["cases"]


@pytest.fixture(scope="module")
def hc():
    path = os.path.join(B.PKG_ROOT, "csrc", "libppo_hostcheck.so")
    if not os.path.exists(path):
        pytest.skip("libppo_hostcheck.so not built (run __graft_entry__.build())")
    lib = C.CDLL(path)
    lib.hc_huber.argtypes = [C.c_double, C.c_double, B.c_dp]
    lib.hc_factor_W.argtypes = [B.c_dp, B.c_dp, B.c_dp, C.c_double, C.c_int, B.c_dp]
    return lib


def test_vector_counts():
    assert {k: len(v) for k, v in GOLD.items()} == {"se3_exp": 39, "pose_oplus": 24, "huber": 28, "mono_project": 32, "stereo_project": 32,
                                                    "onlypose_mono": 24, "onlypose_stereo": 24, "binary_mono": 32, "binary_stereo": 32,
                                                    "unary_mono": 32, "unary_stereo": 32}


def test_oracle_se3_exp_matches_prebuilt_g2o(oracle_lib):
    for c in GOLD["se3_exp"]:
        o = np.zeros(7)
        oracle_lib.oracle_se3_exp(dptr(np.array(c["update"])), dptr(o))
        np.testing.assert_allclose(o, c["qt"], rtol=0, atol=4e-16)


def test_oracle_pose_oplus_matches_prebuilt_g2o(oracle_lib):
    for c in GOLD["pose_oplus"]:
        o = np.zeros(7)
        oracle_lib.oracle_pose_oplus(dptr(np.array(c["qt"])), dptr(np.array(c["update"])), dptr(o))
        np.testing.assert_allclose(o, c["out"], rtol=0, atol=1e-15)


def test_oracle_huber_matches_prebuilt_g2o_bit_for_bit(oracle_lib):
    for c in GOLD["huber"]:
        r = np.zeros(3)
        oracle_lib.oracle_huber(c["e"], c["delta"], dptr(r))
        assert r.tolist() == c["rho"], c


def _one_edge_problem(kind, cam, X):
    P = B.Problem()
    P.set("pose_qt", [[0, 0, 0, 1, 0, 0, 0]]); P.set("pose_fixed", [0]); P.set("pose_cam", [cam + ([0.0] if len(cam) == 4 else [])])
    P.set("point_xyz", [X])
    for n_ in ("mono_point", "mono_pose", "mono_obs", "mono_info", "mono_huber", "stereo_point", "stereo_pose", "stereo_obs", "stereo_info", "stereo_huber",
               "plane_coef", "cuboid_qt", "cuboid_scale", "pl_plane", "pl_pose", "pl_meas", "pl_info", "pl_huber", "bbox_pose", "bbox_cuboid", "bbox_meas",
               "bbox_info", "bbox_huber", "bbox_K", "corner_pose", "corner_cuboid", "corner_meas", "corner_info", "corner_huber", "corner_K", "ptcub_cuboid",
               "ptcub_margin_ratio", "ptcub_info", "cubpl_cuboid", "cubpl_plane", "cubpl_meas", "cubpl_info", "cubpl_huber"):
        P.set(n_, np.zeros(0))
    P.set(kind + "_point", [0]); P.set(kind + "_pose", [0]); P.set(kind + "_obs", [[0.0] * (2 if kind == "mono" else 3)])
    P.set(kind + "_info", [1.0]); P.set(kind + "_huber", [0.0])
    return P


@pytest.mark.parametrize("kind,key,etype", [("mono", "mono_project", 1), ("stereo", "stereo_project", 2)])
def test_oracle_reprojection_matches_prebuilt_g2o_bit_for_bit(oracle, oracle_lib, kind, key, etype):
    """e = obs - cam_project(T.map(X)) with obs = 0 and T = identity gives -cam_project(X)."""
    for c in GOLD[key]:
        s = oracle.create(_one_edge_problem(kind, c["cam"], c["X"]), B.default_options())
        e = np.zeros(3)
        oracle_lib.oracle_edge_eval(s.h, etype, 0, dptr(e), None, None)
        ref = c["uv"] if kind == "mono" else c["uvr"]
        assert (-e[:len(ref)]).tolist() == ref
        s.close()


# ---- the hot binary edges: computeError, linearizeOplus, numeric linearizeOplusXi/Xj, constructQuadraticForm -------------------
_BIN_KEYS = ("error", "J_point", "J_pose", "Jnum_point", "Jnum_pose", "H_point", "H_pose", "H_point_pose", "b_point", "b_pose")


def _oracle_binary(oracle_lib, c, stereo):
    c_dp = C.POINTER(C.c_double)
    oracle_lib.oracle_binary_edge_pin.argtypes = [c_dp] * 4 + [C.c_int, C.c_double, C.c_double, C.c_int, C.c_int] + [c_dp] * 10
    oracle_lib.oracle_binary_edge_pin.restype = C.c_int
    o = {"error": np.zeros(3), "J_point": np.zeros(9), "J_pose": np.zeros(18), "Jnum_point": np.zeros(9), "Jnum_pose": np.zeros(18),
         "H_point": np.zeros(9), "H_pose": np.zeros(36), "H_point_pose": np.zeros(18), "b_point": np.zeros(3), "b_pose": np.zeros(6)}
    oracle_lib.oracle_binary_edge_pin(dptr(np.array(c["qt"])), dptr(np.array(c["cam"])), dptr(np.array(c["xw"])), dptr(np.array(c["obs"])),
                                      int(stereo), c["info"], c["huber_delta"], c["fix_point"], c["fix_pose"], *[dptr(o[k]) for k in _BIN_KEYS])
    return o


@pytest.mark.parametrize("kind", ["mono", "stereo"])
def test_oracle_binary_edges_match_prebuilt_g2o(oracle_lib, kind):
    """EdgeSE3ProjectXYZ / EdgeStereoSE3ProjectXYZ -- the dominant edge population -- taken through computeError, the analytic
    linearizeOplus (types_six_dof_expmap.cpp:135-179, 220-266), the NUMERIC BaseBinaryEdge::linearizeOplusXi/Xj (delta = 1e-9
    push / oplus / pop through the reference's own vertices, base_binary_edge.hpp:217-320) and constructQuadraticForm with a
    RobustKernelHuber attached (base_binary_edge.hpp:55-120) IN THE REFERENCE'S PREBUILT libg2o.so: 32 cases each, with fixed
    point / fixed pose cases, residuals below, at and far above the Huber knee, both mapHessianMemory orientations.
    The oracle reproduces residual, both analytic Jacobians, the point-side numeric Jacobian, H_ii, H_jj, b_i and b_j BIT FOR
    BIT and H_ij to 1e-14 relative.  The pose-side numeric Jacobian agrees to 2e-4 relative only: the oracle's pose oplus
    differs from g2o's by one unit in the last place (test_oracle_pose_oplus_matches_prebuilt_g2o, 2.2e-16) and the central
    difference divides that by 2e-9 -- the delta = 1e-9 noise floor every numerically differentiated edge of this path has."""
    n_fixed = n_knee = 0
    for c in GOLD["binary_" + kind]:
        D = len(c["error"])
        o = _oracle_binary(oracle_lib, c, kind == "stereo")
        n_fixed += c["fix_point"] + c["fix_pose"]
        chi2 = c["info"] * float(np.dot(c["error"], c["error"]))
        n_knee += int(c["huber_delta"] > 0 and chi2 > c["huber_delta"] ** 2)
        for k, n, tol in (("error", D, 0.0), ("J_point", 3 * D, 0.0), ("J_pose", 6 * D, 0.0), ("Jnum_point", 3 * D, 0.0), ("Jnum_pose", 6 * D, 5e-4),
                          ("H_point", 9, 1e-15), ("H_pose", 36, 0.0), ("H_point_pose", 18, 5e-14), ("b_point", 3, 0.0), ("b_pose", 6, 0.0)):
            ref = np.array(c[k])
            if tol == 0.0:
                assert o[k][:n].tolist() == ref.tolist(), (kind, k, c)
            else:
                np.testing.assert_array_less(np.abs(o[k][:n] - ref), tol * np.maximum(1.0, np.abs(ref)) + 1e-300, err_msg="%s %s" % (kind, k))
        if c["fix_point"]:
            assert not np.any(c["H_point"]) and not np.any(c["b_point"]) and not np.any(c["H_point_pose"]) and not np.any(c["Jnum_point"])
        if c["fix_pose"]:
            assert not np.any(c["H_pose"]) and not np.any(c["b_pose"]) and not np.any(c["H_point_pose"]) and not np.any(c["Jnum_pose"])
    assert n_fixed >= 6 and n_knee >= 8


_UN_KEYS = ("error", "chi2", "J", "H_pose", "b_pose")


@pytest.mark.parametrize("kind", ["mono", "stereo"])
def test_oracle_unary_edges_match_prebuilt_g2o_bit_for_bit(oracle_lib, kind):
    """Edge(Stereo)SE3ProjectXYZOnlyPose -- the edges of Optimizer::PoseOptimization (OPT:247-459), and the driver every unary edge
    of the path (EdgePointCuboidOnlyObject, A13) runs through -- taken IN THE REFERENCE'S PREBUILT libg2o.so through computeError,
    BaseEdge::chi2, BaseUnaryEdge<2|3,...,VertexSE3Expmap>::linearizeOplus(JacobianWorkspace&) and ::constructQuadraticForm()
    (base_unary_edge.hpp:43-80) with a RobustKernelHuber attached: 32 cases each, residuals below, AT (chi2 within an ulp of
    delta^2) and far above the knee, no-kernel cases, fixed-pose cases.  The oracle reproduces every output bit for bit --
    including b, whose grouping differs from the binary edges' ((rho1 A' Omega) e, not A' ((-Omega e) rho1)).
    The exported linearizeOplus(JacobianWorkspace&) maps the workspace and tail-calls the virtual linearizeOplus(), which both
    classes override analytically; the numeric body (base_unary_edge.hpp:82-123) does not exist in the binary (see the pinning
    script's docstring) -- it is the Xi half of BaseBinaryEdge::linearizeOplusXi, pinned in the test above."""
    c_dp = C.POINTER(C.c_double)
    oracle_lib.oracle_unary_edge_pin.argtypes = [c_dp] * 4 + [C.c_int, C.c_double, C.c_double, C.c_int] + [c_dp] * 5
    oracle_lib.oracle_unary_edge_pin.restype = C.c_int
    n_fixed = n_above = n_at = n_plain = 0
    for c in GOLD["unary_" + kind]:
        D = len(c["error"])
        o = {"error": np.zeros(3), "chi2": np.zeros(1), "J": np.zeros(18), "H_pose": np.zeros(36), "b_pose": np.zeros(6)}
        oracle_lib.oracle_unary_edge_pin(dptr(np.array(c["qt"])), dptr(np.array(c["cam"])), dptr(np.array(c["xw"])), dptr(np.array(c["obs"])),
                                         int(kind == "stereo"), c["info"], c["huber_delta"], c["fix_pose"], *[dptr(o[k]) for k in _UN_KEYS])
        for k, n in (("error", D), ("chi2", 1), ("J", 6 * D), ("H_pose", 36), ("b_pose", 6)):
            assert o[k][:n].tolist() == np.atleast_1d(np.array(c[k], dtype=float)).tolist(), (kind, k, c)
        d2 = c["huber_delta"] ** 2
        n_fixed += c["fix_pose"]
        n_plain += int(c["huber_delta"] == 0)
        n_above += int(c["huber_delta"] > 0 and c["chi2"] > 1.5 * d2)
        n_at += int(c["huber_delta"] > 0 and abs(c["chi2"] - d2) <= 1e-12 * d2)
        if c["fix_pose"]:
            assert not np.any(c["H_pose"]) and not np.any(c["b_pose"])
    assert n_fixed >= 2 and n_above >= 6 and n_at >= 3 and n_plain >= 6


def test_reference_numeric_jacobian_noise_floor():
    """What delta = 1e-9 central differences are worth, measured on the reference's own binary: its numeric Jacobian of the mono
    edge differs from its analytic one by up to ~1e-4 relative (pixel-scale residuals carry ~1e-13 of rounding, divided by
    2e-9).  The parity bars of tests/test_parity_gpu.py for plane / cuboid states sit on this floor, not on the solver."""
    worst = 0.0
    for c in GOLD["binary_mono"]:
        if c["fix_point"] or c["fix_pose"]:
            continue
        for a, n in (("J_point", "Jnum_point"), ("J_pose", "Jnum_pose")):
            ja, jn = np.array(c[a]), np.array(c[n])
            worst = max(worst, float((np.abs(ja - jn) / np.maximum(1.0, np.abs(ja))).max()))
    assert 1e-7 < worst < 1e-2, worst


# ---- the product's device math, compiled for the host -----------------------------------------------------------------------
def test_device_math_se3_and_pose_oplus_match_prebuilt_g2o(hc):
    for c in GOLD["se3_exp"]:
        o = np.zeros(7)
        hc.hc_se3_exp(dptr(np.array(c["update"])), dptr(o))
        np.testing.assert_allclose(o, c["qt"], rtol=0, atol=4e-16)
    for c in GOLD["pose_oplus"]:
        o = np.zeros(7)
        hc.hc_pose_oplus(dptr(np.array(c["qt"])), dptr(np.array(c["update"])), dptr(o))
        np.testing.assert_allclose(o, c["out"], rtol=0, atol=1e-15)


def test_device_math_huber_and_projection_match_prebuilt_g2o(hc):
    for c in GOLD["huber"]:
        r = np.zeros(2)
        hc.hc_huber(c["e"], c["delta"], dptr(r))
        assert r.tolist() == c["rho"][:2], c
    ident = np.array([0, 0, 0, 1, 0, 0, 0.0])
    for c in GOLD["mono_project"]:
        e, Jl, Jp = np.zeros(2), np.zeros(6), np.zeros(12)
        hc.hc_mono(dptr(ident), dptr(np.array(c["X"])), dptr(np.array(c["cam"])), dptr(np.zeros(2)), dptr(e), dptr(Jl), dptr(Jp))
        assert (-e).tolist() == c["uv"]
    for c in GOLD["stereo_project"]:
        e, Jl, Jp = np.zeros(3), np.zeros(9), np.zeros(18)
        hc.hc_stereo(dptr(ident), dptr(np.array(c["X"])), dptr(np.array(c["cam"])), dptr(np.zeros(3)), dptr(e), dptr(Jl), dptr(Jp))
        assert (-e).tolist() == c["uvr"]


@pytest.mark.parametrize("kind", ["mono", "stereo"])
def test_device_math_binary_edges_match_prebuilt_g2o(hc, kind):
    """The per-edge arithmetic of the linearisation kernels (ppo_math.hpp: mono_eval / stereo_eval, huber, acc_landmark, acc_pose,
    factor_record + factor_M) against computeError / linearizeOplus / constructQuadraticForm of the reference's prebuilt
    libg2o.so on the same 32 + 32 cases.  The device header regroups the arithmetic (three divisions instead of eight, W from a
    32-byte record): agreement is to rounding, 1e-12 relative, not bit for bit."""
    hc.hc_quadform.argtypes = [B.c_dp] * 4 + [C.c_int, C.c_double, C.c_double] + [B.c_dp] * 5
    stereo = kind == "stereo"
    for c in GOLD["binary_" + kind]:
        D = len(c["error"])
        e, Jl, Jp = np.zeros(3), np.zeros(9), np.zeros(18)
        (hc.hc_stereo if stereo else hc.hc_mono)(dptr(np.array(c["qt"])), dptr(np.array(c["xw"])), dptr(np.array(c["cam"])), dptr(np.array(c["obs"])),
                                                dptr(e), dptr(Jl), dptr(Jp))
        np.testing.assert_allclose(e[:D], c["error"], rtol=0, atol=2e-12)  # residuals of up to ~1e3 px
        for got, key in ((Jl[:3 * D], "J_point"), (Jp[:6 * D], "J_pose")):
            ref = np.array(c[key])
            assert (np.abs(got - ref) <= 1e-12 * np.maximum(1.0, np.abs(ref))).all(), (kind, key)
        ee, Dm, bl, A, W = np.zeros(3), np.zeros(6), np.zeros(3), np.zeros(27), np.zeros(18)
        hc.hc_quadform(dptr(np.array(c["qt"])), dptr(np.array(c["xw"])), dptr(np.array(c["cam"])), dptr(np.array(c["obs"])), int(stereo),
                       c["info"], c["huber_delta"], dptr(ee), dptr(Dm), dptr(bl), dptr(A), dptr(W))
        if c["fix_point"] or c["fix_pose"]:
            continue  # the kernels skip a fixed vertex's blocks by index (lm_free / pose_col < 0), not inside this arithmetic
        Hll = np.array(c["H_point"]).reshape(3, 3)
        Hpp = np.array(c["H_pose"]).reshape(6, 6)
        Hlp = np.array(c["H_point_pose"]).reshape(3, 6)
        scale = max(1.0, np.abs(Hpp).max())
        got_D = np.array([[Dm[0], Dm[1], Dm[2]], [Dm[1], Dm[3], Dm[4]], [Dm[2], Dm[4], Dm[5]]])
        np.testing.assert_allclose(got_D, Hll, rtol=0, atol=1e-12 * max(1.0, np.abs(Hll).max()))
        got_A = np.zeros((6, 6))
        for r in range(6):
            for cc in range(r + 1):
                got_A[r, cc] = got_A[cc, r] = A[r * (r + 1) // 2 + cc]
        np.testing.assert_allclose(got_A, Hpp, rtol=0, atol=1e-12 * scale)
        np.testing.assert_allclose(W.reshape(6, 3), Hlp.T, rtol=0, atol=1e-12 * scale)
        # b = J' (rho1 w e) sums terms of both signs: 1e-11 of the largest entry
        np.testing.assert_allclose(bl, c["b_point"], rtol=0, atol=1e-11 * max(1.0, np.abs(c["b_point"]).max()))
        np.testing.assert_allclose(A[21:27], c["b_pose"], rtol=0, atol=1e-11 * max(1.0, np.abs(c["b_pose"]).max()))


@pytest.mark.parametrize("kind", ["mono", "stereo"])
def test_device_math_unary_edges_match_prebuilt_g2o(hc, kind):
    """The per-observation arithmetic of the pose-only kernel (csrc/ppo_pose_math.hpp: residual with the OnlyPose stereo flavour --
    `bf` is a DOUBLE member there, a float argument in the binary edge --, chi2, Huber, 6x6 block and right-hand side) against
    computeError / chi2 / BaseUnaryEdge::constructQuadraticForm of the reference's prebuilt libg2o.so on the same 32 + 32 cases:
    to rounding (1e-12 relative), the kernel regroups."""
    hc.hc_pose_obs.argtypes = [B.c_dp] * 4 + [C.c_double, C.c_double] + [B.c_dp] * 2
    hc.hc_pose_obs.restype = C.c_double
    for c in GOLD["unary_" + kind]:
        D = len(c["error"])
        obs = np.array(c["obs"], dtype=float)
        if kind == "mono":
            obs[2] = -1.0
        e, acc = np.zeros(3), np.zeros(28)
        chi2 = hc.hc_pose_obs(dptr(np.array(c["qt"])), dptr(np.array(c["cam"])), dptr(np.array(c["xw"])), dptr(obs), c["info"], c["huber_delta"], dptr(e), dptr(acc))
        np.testing.assert_allclose(e[:D], c["error"], rtol=0, atol=2e-12)  # residuals of up to ~1e3 px
        assert abs(chi2 - c["chi2"]) <= 1e-12 * max(1.0, c["chi2"])
        if c["fix_pose"]:
            continue  # a fixed pose has no system at all in the kernel
        Hpp = np.array(c["H_pose"]).reshape(6, 6)
        got = np.zeros((6, 6))
        for r in range(6):
            for cc in range(r + 1):
                got[r, cc] = got[cc, r] = acc[r * (r + 1) // 2 + cc]
        np.testing.assert_allclose(got, Hpp, rtol=0, atol=1e-12 * max(1.0, np.abs(Hpp).max()))
        np.testing.assert_allclose(acc[21:27], c["b_pose"], rtol=0, atol=1e-11 * max(1.0, np.abs(c["b_pose"]).max()))


def test_device_math_matches_oracle_on_plane_and_cuboid_functions(hc, oracle_lib):
    """The custom plane / cuboid types cannot be pinned by a reference binary (lib/libORB_SLAM2.so does not load here);
    the two independent implementations (oracle_math.hpp, ppo_math.hpp) are at least held against each other."""
    rng = np.random.default_rng(11)
    K = np.array([480, 0, 320, 0, 480, 240, 0, 0, 1.0])
    for _ in range(50):
        n = rng.normal(size=3)
        p = np.concatenate([n / np.linalg.norm(n), [rng.uniform(0.5, 3.0)]])
        v = rng.normal(size=3) * 0.1
        a, b = np.zeros(4), np.zeros(4)
        oracle_lib.oracle_plane_oplus(dptr(p), dptr(v), dptr(a))
        hc.hc_plane_oplus(dptr(p), dptr(v), dptr(b))
        # (the product writes the plane frame algebraically, the oracle through atan2 and half-angle quaternions like the
        # reference: a few ulp apart)
        np.testing.assert_allclose(a, b, rtol=0, atol=4e-15)
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        if q[3] < 0:
            q = -q
        qt = np.concatenate([q, rng.normal(size=3)])
        m = np.concatenate([n / np.linalg.norm(n) + rng.normal(size=3) * 0.05, [rng.uniform(0.5, 3.0)]])
        for kind in (0, 2):
            loc, eo, eh = np.zeros(4), np.zeros(3), np.zeros(3)
            oracle_lib.oracle_plane_transform(dptr(qt), dptr(p), dptr(loc))
            mm = m / np.linalg.norm(m[:3])
            oracle_lib.oracle_plane_ominus(dptr(loc), dptr(mm), kind, dptr(eo))
            hc.hc_plane_residual(dptr(qt), dptr(p), dptr(mm), kind, dptr(eh))
            np.testing.assert_allclose(eo, eh, rtol=0, atol=1e-13)
        # cuboid: oplus in the mode the reference uses, then bbox + corners
        yaw = rng.uniform(-3, 3)
        cq = np.array([0, 0, np.sin(yaw / 2), np.cos(yaw / 2)])
        if cq[3] < 0:
            cq = -cq
        c10 = np.concatenate([cq, [rng.normal(), rng.normal(), 6.0 + rng.normal()], rng.uniform(0.1, 0.5, size=3)])
        upd = rng.normal(size=9) * 0.05
        oq, os_, oh = np.zeros(7), np.zeros(3), np.zeros(10)
        oracle_lib.oracle_cuboid_oplus(dptr(c10[:7].copy()), dptr(c10[7:].copy()), dptr(upd), 3, None, dptr(oq), dptr(os_))
        hc.hc_cuboid_oplus(dptr(c10), dptr(upd), 3, None, dptr(oh))
        np.testing.assert_allclose(np.concatenate([oq, os_]), oh, rtol=0, atol=1e-15)
        uv, bb, eb, ec = np.zeros(16), np.zeros(4), np.zeros(4), np.zeros(16)
        cam = np.array([0, 0, 0, 1, 0, 0, 0.0])
        oracle_lib.oracle_cuboid_project(dptr(c10[:7].copy()), dptr(c10[7:].copy()), dptr(cam), dptr(K), dptr(uv), dptr(bb))
        hc.hc_bbox(dptr(cam), dptr(c10), dptr(K), dptr(np.zeros(4)), dptr(eb))
        hc.hc_corner(dptr(cam), dptr(c10), dptr(K), dptr(np.zeros(16)), dptr(ec))
        np.testing.assert_allclose(eb, bb, rtol=0, atol=1e-11)
        np.testing.assert_allclose(ec, uv, rtol=0, atol=1e-11)


def test_plane_frame_without_trigonometry_matches_oracle_at_steps_and_poles(hc, oracle_lib):
    """ppo_math.hpp writes Rz(az) Ry(-el) from the direction cosines and skips sin/cos for numeric-differentiation steps;
    the oracle follows include/G2O_Plane3D.h:66-87 through atan2 and angle-axis quaternions.  Held against each other at the
    steps the numeric Jacobians take (+-1e-9 per column) and for normals on / near the z axis (azimuth undefined)."""
    rng = np.random.default_rng(12)
    normals = [np.array([0, 0, 1.0]), np.array([0, 0, -1.0]), np.array([1e-12, 0, 1.0]), np.array([0, -1e-9, -1.0]),
               np.array([1.0, 0, 0]), np.array([-1.0, 0, 0]), np.array([0, 1.0, 0])] + [rng.normal(size=3) for _ in range(40)]
    for n in normals:
        p = np.concatenate([n / np.linalg.norm(n), [rng.uniform(0.5, 3.0)]])
        for c in range(3):
            for step in (1e-9, -1e-9, 9.9e-9, 1.1e-8, 1e-3):
                v = np.zeros(3); v[c] = step
                a, b = np.zeros(4), np.zeros(4)
                oracle_lib.oracle_plane_oplus(dptr(p), dptr(v), dptr(a))
                hc.hc_plane_oplus(dptr(p), dptr(v), dptr(b))
                np.testing.assert_allclose(a, b, rtol=0, atol=4e-15)
        qt = np.array([0, 0, 0, 1, 0.1, -0.2, 0.3])
        m = np.concatenate([p[:3] + rng.normal(size=3) * 0.02, [p[3] + 0.01]])
        m[:3] /= np.linalg.norm(m[:3])
        loc, eo, eh = np.zeros(4), np.zeros(3), np.zeros(3)
        oracle_lib.oracle_plane_transform(dptr(qt), dptr(p), dptr(loc))
        oracle_lib.oracle_plane_ominus(dptr(loc), dptr(m), 0, dptr(eo))
        hc.hc_plane_residual(dptr(qt), dptr(p), dptr(m), 0, dptr(eh))
        np.testing.assert_allclose(eo, eh, rtol=0, atol=1e-13)


def test_factor_record_rebuilds_the_coupling_block(hc):
    """The linearisation keeps 32 bytes per reprojection edge -- (x/z, y/z, 1/z, +-w) -- instead of the 6x3 block
    W = Jp' w Jl, and the Schur pair kernel rebuilds W = M(record, intrinsics) R.  The rebuilt block must
    equal the one formed from the Jacobians of EdgeSE3ProjectXYZ / EdgeStereoSE3ProjectXYZ (which test_golden pins on the
    reference's libg2o) to rounding."""
    rng = np.random.default_rng(13)
    for case in range(200):
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        if q[3] < 0:
            q = -q
        qt = np.concatenate([q, rng.normal(size=3)])
        Rm = np.array([[1 - 2 * (q[1] ** 2 + q[2] ** 2), 2 * (q[0] * q[1] - q[2] * q[3]), 2 * (q[0] * q[2] + q[1] * q[3])],
                       [2 * (q[0] * q[1] + q[2] * q[3]), 1 - 2 * (q[0] ** 2 + q[2] ** 2), 2 * (q[1] * q[2] - q[0] * q[3])],
                       [2 * (q[0] * q[2] - q[1] * q[3]), 2 * (q[1] * q[2] + q[0] * q[3]), 1 - 2 * (q[0] ** 2 + q[1] ** 2)]])
        pc = np.array([rng.uniform(-2, 2), rng.uniform(-1.5, 1.5), rng.uniform(0.3, 8.0)])
        X = Rm.T @ (pc - qt[4:])
        cam = np.array([rng.uniform(300, 600), rng.uniform(300, 600), 320.0, 240.0, rng.uniform(20, 60)])
        w = float(rng.uniform(0.05, 2.0))
        for stereo, dim, fn in ((0, 2, hc.hc_mono), (1, 3, hc.hc_stereo)):
            e, Jl, Jp = np.zeros(dim), np.zeros(3 * dim), np.zeros(6 * dim)
            fn(dptr(qt), dptr(X), dptr(cam), dptr(np.zeros(dim)), dptr(e), dptr(Jl), dptr(Jp))
            W_ref = Jp.reshape(dim, 6).T @ (w * Jl.reshape(dim, 3))
            W = np.zeros(18)
            hc.hc_factor_W(dptr(qt), dptr(X), dptr(cam), w, stereo, dptr(W))
            np.testing.assert_allclose(W.reshape(6, 3), W_ref, rtol=0, atol=1e-12 * max(1.0, np.abs(W_ref).max()))
        # an inactive edge (w = 0) contributes nothing
        hc.hc_factor_W(dptr(qt), dptr(X), dptr(cam), 0.0, 0, dptr(W))
        assert not W.any()


# ---- regression trace of the oracle -----------------------------------------------------------------------------------------
def test_oracle_reproduces_its_committed_lm_trace(oracle):
    gold = json.load(open(os.path.join(HERE, "golden", "oracle_trace_small.json")))
    P = synth.make_problem(gold["config"], seed=gold["seed"])
    s = oracle.create(P, B.default_options())
    st = s.run()
    r = s.read()
    assert len(st["iters"]) == len(gold["iters"])
    for a, b in zip(st["iters"], gold["iters"]):
        assert (a["pass_"], a["iteration"], a["trials"], a["accepted"], a["result"]) == (b["pass_"], b["iteration"], b["trials"], b["accepted"], b["result"])
        # the numeric (delta = 1e-9) Jacobians make the trace sensitive to libm at the 1e-7 level
        assert abs(a["chi2_after"] - b["chi2_after"]) <= 1e-6 * abs(b["chi2_after"])
        assert abs(a["lambda_"] - b["lambda_"]) <= 1e-3 * abs(b["lambda_"])
    np.testing.assert_allclose(r.pose_qt, np.array(gold["pose_qt"]), atol=1e-6)
    np.testing.assert_allclose(r.cuboid_scale, np.array(gold["cuboid_scale"]), atol=1e-5)
    assert int(r.mono_level.sum()) == gold["n_mono_level1"]
