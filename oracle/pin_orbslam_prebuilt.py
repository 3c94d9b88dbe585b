#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: golden vectors from the reference's OWN prebuilt lib/libORB_SLAM2.so -- the custom plane / cuboid half.

Stage A of the reference-binary pin.  Loaded as oracle/orbslam_loader.py describes (empty sonames + a trap
sink that aborts on any call that leaves reference code; `armed hits: 0` is recorded in the fixture), objects built as
oracle/orbslam_objects.py describes.  What runs inside the reference binary, per case:

  plane_normalize   g2o::Plane3D::fromVector through VertexPlane::setEstimateDataImpl          include/G2O_Plane3D.h:45-48,120-125
  plane_oplus       g2o::VertexPlane::oplusImpl -> Plane3D::oplus                               include/G2O_Plane3D.h:74-87,152-155
  cuboid_oplus      g2o::VertexCuboid::oplusImpl (every flag combination, fixedscale)           src/g2o_cuboid.cc:39-67
  exptwist          g2o::exptwist_norollpitch                                                   src/g2o_cuboid.cc:6-36
  edge_plane / edge_plane_ver / edge_plane_par / edge_bbox / edge_corner / edge_ptcub / edge_cubpl / edge_cuboid3d
                    <Edge>::computeError, BaseEdge::chi2, the NUMERIC BaseBinaryEdge / BaseUnaryEdge::linearizeOplus (delta 1e-9
                    through the vertices' own push / oplus / pop), mapHessianMemory + constructQuadraticForm with a
                    RobustKernelHuber attached                      include/G2O_Plane3D.h:176-310,470-473, src/g2o_cuboid.cc:70-160,
                                                                    include/g2o_cuboid.h:83-109,322-352,
                                                                    G2O/core/base_binary_edge.hpp:55-120,217-320, base_unary_edge.hpp:43-123

  python oracle/pin_orbslam_prebuilt.py               regenerate tests/golden/orbslam_prebuilt.json (needs /root/reference)
  python oracle/pin_orbslam_prebuilt.py --check-only  compare the oracle with the committed file (runs anywhere)
"""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, HERE)
GOLD = os.path.join(REPO, "tests", "golden", "orbslam_prebuilt.json")

KINDS = ["plane", "plane_ver", "plane_par", "bbox", "corner", "ptcub", "cubpl", "cuboid3d"]
K_DEFAULT = [480.0, 0, 320.0, 0, 480.0, 240.0, 0, 0, 1.0]


def unit_quat(rng, yaw_only=False):
    if yaw_only:
        a = rng.uniform(-np.pi, np.pi)
        q = np.array([0.0, 0.0, np.sin(a / 2), np.cos(a / 2)])
    else:
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
    if q[3] < 0:
        q = -q
    return q


def rot(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def flat(m):
    return None if m is None else np.asarray(m, dtype=np.float64).reshape(-1).tolist()


# ---------------------------------------------------------------------------------------------------------------------------------
# case inputs (plain numbers; the same dictionaries are what the checker feeds the oracle)
# ---------------------------------------------------------------------------------------------------------------------------------
def edge_inputs(kind, k, rng):
    """Inputs of one custom-edge case: vertex estimates as written into the objects, measurement, information, kernel."""
    c = {"kind": kind, "fix0": 0, "fix1": 0}
    if kind in ("plane", "plane_ver", "plane_par"):
        q, t = unit_quat(rng), rng.normal(size=3) * 1.5
        n = rng.normal(size=3)
        n /= np.linalg.norm(n)
        d = rng.uniform(0.3, 5.0)
        world = np.concatenate([n, [d]]) * rng.uniform(0.5, 2.0)  # not normalised: VertexPlane::setEstimateDataImpl does it
        if k % 8 == 3:
            world = -world  # d < 0 on input
        R = rot(q)
        nc = R @ n
        dc = d - t @ nc
        local = np.concatenate([nc, [dc]])  # may have d < 0: operator*(Isometry, Plane3D) flips (G2O_Plane3D.h:137-138)
        if kind == "plane":
            m = local.copy()
            m[:3] += rng.normal(size=3) * [0.005, 0.02, 0.2, 0.6][k % 4]
            m[3] += rng.normal() * [0.01, 0.05, 0.5, 1.0][k % 4]
            info, delta = [3282.8, 3282.8, 10000.0], float(np.float32(np.sqrt(500.0)))  # Optimizer.cc:2194-2203 with Parameters.cc:68-70
        else:
            # a measured plane roughly vertical / parallel to the local one
            a = rng.normal(size=3)
            if kind == "plane_ver":
                v = np.cross(nc, a)
                v /= np.linalg.norm(v)
                mn = v + rng.normal(size=3) * [0.005, 0.05, 0.3][k % 3]
            else:
                mn = nc * (1 if k % 2 else -1) + rng.normal(size=3) * [0.005, 0.05, 0.3][k % 3]
            m = np.concatenate([mn, [rng.uniform(-3, 3)]])
            if k == 31 and kind == "plane_ver":
                # the degenerate direction of ominus_ver: measured normal EXACTLY the local one -> cross product 0, 0 / 0 = NaN
                # (G2O_Plane3D.h:99-100); identity pose so that the reference's own R n + normalisation reproduces the input bits
                q, t = np.array([0.0, 0.0, 0.0, 1.0]), np.zeros(3)
                world = np.concatenate([n, [d]])
                m = world.copy()
            info, delta = [3282.8 / 0.25, 3282.8 / 0.25], float(np.float32(np.sqrt(200.0)))  # Optimizer.cc:2198-2205
        if k % 4 == 2:
            delta = 0.0
        c.update(v0=world.tolist(), v1=q.tolist() + t.tolist(), meas=m.tolist(), info=info, huber=delta)
        c["fix1"] = 1 if k % 8 == 5 else 0
    elif kind in ("bbox", "corner", "cuboid3d"):
        flags = [3, 3, 0, 1, 2, 3, 4, 7][k % 8]
        cq = unit_quat(rng, yaw_only=(flags & 1) or k % 2 == 0)
        ct = np.array([rng.normal() * 0.8, rng.normal() * 0.4, rng.uniform(2.5, 7.0)])
        scale = rng.uniform(0.1, 0.6, size=3)
        # a camera near the origin looking down +z with a small tilt
        pq = unit_quat(rng)
        pq = np.array([pq[0] * 0.08, pq[1] * 0.08, pq[2] * 0.08, 1.0])
        pq /= np.linalg.norm(pq)
        pt = rng.normal(size=3) * 0.2
        c.update(v0=pq.tolist() + pt.tolist(), v1=cq.tolist() + ct.tolist() + scale.tolist(), flags=flags,
                 fixedscale=[0.0, 0.0, 0.0] if k % 16 != 9 else rng.uniform(0.2, 0.5, size=3).tolist(), K=K_DEFAULT)
        if kind == "cuboid3d":
            # measured cuboid in the CAMERA frame: the true one seen from the camera, perturbed; every 4th rotated by a multiple of 90 deg
            Rcw, Rwo = rot(pq), rot(cq)
            Rco = Rcw @ Rwo
            tco = Rcw @ ct + pt
            turn = [0, 1, 2, 3][k % 4] * np.pi / 2 + rng.normal() * [0.01, 0.1][k % 2]
            Rz = np.array([[np.cos(turn), -np.sin(turn), 0], [np.sin(turn), np.cos(turn), 0], [0, 0, 1]])
            Rm = Rco @ Rz
            # quaternion of Rm (numpy, input only)
            w = np.sqrt(max(0.0, 1 + Rm[0, 0] + Rm[1, 1] + Rm[2, 2])) / 2
            if w > 1e-6:
                mq = np.array([(Rm[2, 1] - Rm[1, 2]) / (4 * w), (Rm[0, 2] - Rm[2, 0]) / (4 * w), (Rm[1, 0] - Rm[0, 1]) / (4 * w), w])
            else:
                mq = unit_quat(rng)
            mq /= np.linalg.norm(mq)
            ms = scale.copy()
            if k % 2 == 1 and k % 4 in (1, 3):
                ms[[0, 1]] = ms[[1, 0]]
            ms = ms + rng.normal(size=3) * 0.02
            mt = tco + rng.normal(size=3) * 0.05
            c.update(meas=mq.tolist() + mt.tolist() + ms.tolist(), info=(rng.uniform(0.5, 4.0, size=9)).tolist(), huber=0.0 if k % 3 else 1.5)
        else:
            # measurement = projection of the true box + pixel noise; computed by the reference below (needs the projection) -> filled later
            c.update(meas=None, info=[0.49] * (4 if kind == "bbox" else 16), huber=float(np.sqrt(80.0)) if k % 4 != 2 else 0.0,
                     noise=(rng.normal(size=16) * [1.0, 4.0, 40.0][k % 3]).tolist())
        c["fix0"] = 1 if k % 8 == 6 else 0
    elif kind == "ptcub":
        flags = [3, 3, 0, 1, 7][k % 5]
        cq = unit_quat(rng, yaw_only=True)
        ct = np.array([rng.normal(), rng.normal() * 0.4, rng.uniform(2.5, 7.0)])
        scale = rng.uniform(0.15, 0.6, size=3)
        npts = int(rng.integers(0, 40)) if k != 7 else 0
        # points around the box: inside, within the margin, far outside
        local = rng.uniform(-1, 1, size=(npts, 3)) * scale * rng.choice([0.8, 1.5, 4.0], size=(npts, 1))
        pts = (rot(cq) @ local.T).T + ct
        c.update(v0=cq.tolist() + ct.tolist() + scale.tolist(), v1=None, flags=flags, fixedscale=[0.0, 0.0, 0.0], meas=[0.0, 0.0, 0.0], info=[1.0, 1.0, 1.0],
                 huber=0.0, points=pts.tolist(), margin_ratio=[2.0, 1.0, 0.5][k % 3],
                 prior_half=[0.0, 0.0, 0.0] if k % 4 else (scale * rng.uniform(0.8, 1.2, size=3)).tolist())
    elif kind == "cubpl":
        cq = unit_quat(rng, yaw_only=True)
        n = rng.normal(size=3)
        n /= np.linalg.norm(n)
        c.update(v0=cq.tolist() + rng.normal(size=3).tolist() + rng.uniform(0.1, 0.5, size=3).tolist(), v1=n.tolist() + [rng.uniform(0.5, 3.0)], flags=3,
                 fixedscale=[0.0, 0.0, 0.0], meas=(rng.normal(size=3) * [0.01, 1.0][k % 2]).tolist(), info=[3282.8 / 4.0, 3282.8 / 4.0, 2500.0],
                 huber=float(np.sqrt(10.0)) if k % 2 else 0.0)
    return c


# ---------------------------------------------------------------------------------------------------------------------------------
# the reference side
# ---------------------------------------------------------------------------------------------------------------------------------
def drive_edge(W, c):
    """One case through the reference binary; returns the output dictionary."""
    import orbslam_objects as O
    kind = c["kind"]
    if kind in ("plane", "plane_ver", "plane_par"):
        v0 = W.plane(c["v0"], vid=1, marginalized=1)
        v1 = W.pose(c["v1"], fixed=c["fix1"], vid=0)
        meas = W.plane3d(c["meas"])  # Plane3D(const Vector4D&) of setMeasurement(Converter::toPlane3D(..))
        e = W.edge(kind, v0, v1, meas, c["info"], c["huber"])
        extra = {"v0_normalized": W.plane_state(v0), "meas_normalized": meas.tolist()}
    elif kind in ("bbox", "corner", "cuboid3d"):
        v0 = W.pose(c["v0"], fixed=c["fix0"], vid=0)
        v1 = W.cuboid(c["v1"][:7], c["v1"][7:], c["flags"], c["fixedscale"], vid=1)
        meas = c["meas"]
        extra = {}
        if kind == "cuboid3d":
            meas = c["meas"][:7] + [0.0] + c["meas"][7:]  # g2o::cuboid = SE3Quat (q, t, pad to 64 bytes) + scale
        elif meas is None:
            D = 4 if kind == "bbox" else 16
            e0 = W.edge(kind, v0, v1, [0.0] * D, c["info"], 0.0, K=c["K"])
            proj = W.compute_error(kind, e0)  # error against a zero measurement = the projection itself
            meas = (proj + np.array(c["noise"][:D])).tolist()
            c["meas"] = meas
        e = W.edge(kind, v0, v1, meas, c["info"], c["huber"], K=c.get("K"))
    elif kind == "ptcub":
        v0 = W.cuboid(c["v0"][:7], c["v0"][7:], c["flags"], c["fixedscale"], vid=1)
        v1 = None
        e = W.edge(kind, v0, None, c["meas"], c["info"], c["huber"], points=c["points"], margin_ratio=c["margin_ratio"], prior_half=c["prior_half"])
        extra = {}
    else:  # cubpl
        v0 = W.cuboid(c["v0"][:7], c["v0"][7:], c["flags"], c["fixedscale"], vid=1)
        v1 = W.plane(c["v1"], vid=2, marginalized=1)
        e = W.edge(kind, v0, v1, c["meas"], c["info"], c["huber"])
        extra = {"v1_normalized": W.plane_state(v1)}
    err = W.compute_error(kind, e)
    chi2 = W.chi2(kind, e)
    J0, J1 = W.linearize(kind, e)
    L = O.EDGE_CLASSES[kind][2]
    assert np.array_equal(err, e[L["err"] // 8:L["err"] // 8 + L["D"]], equal_nan=True)  # _error restored after the differences
    q = W.quadratic_form(kind, e, v0, v1)
    out = {"error": err.tolist(), "chi2": chi2, "J0": flat(J0), "J1": flat(J1), "H0": flat(q["H0"]), "b0": flat(q["b0"]),
           "H1": flat(q.get("H1")), "b1": flat(q.get("b1")), "H01": flat(q.get("H01"))}
    out.update(extra)
    return out


def generate():
    import orbslam_loader as loader
    import orbslam_objects as O
    W = O.World()
    out = {"source": "lib/libORB_SLAM2.so + Thirdparty/g2o/lib/libg2o.so (prebuilt, gcc 7.5 / Ubuntu 18.04, Eigen as compiled in)", "cases": {}}
    rng = np.random.default_rng(20261002)
    # ---- plane normalisation + oplus ----
    norm_cases, op_cases = [], []
    for k in range(40):
        n = rng.normal(size=3)
        if k == 36: n = np.array([0.0, 0.0, 1.0])       # elevation +90 deg: azimuth of (0, 0)
        if k == 37: n = np.array([0.0, 0.0, -1.0])
        if k == 38: n = np.array([-1.0, 0.0, 0.0])      # azimuth pi
        if k == 39: n = np.array([-1.0, -1e-17, 0.0])   # azimuth -pi
        d = rng.uniform(-3, 5)
        raw = np.concatenate([n, [d]]) * rng.uniform(0.3, 3.0)
        v = W.plane(raw)
        st0 = W.plane_state(v)
        norm_cases.append({"raw": raw.tolist(), "out": st0})
        upd = rng.normal(size=3) * [1.0, 0.1, 1e-3, 1e-9][k % 4]
        if k % 8 == 5:
            upd[2] = st0[3] + 0.2  # drives the distance through zero: the sign flip of normalize()
        W.oplus(v, upd)
        op_cases.append({"coef": st0, "update": upd.tolist(), "out": W.plane_state(v)})
    out["cases"]["plane_normalize"] = norm_cases
    out["cases"]["plane_oplus"] = op_cases
    # ---- cuboid oplus ----
    cub = []
    for k in range(48):
        flags = [0, 1, 2, 3, 4, 5, 6, 7][k % 8]
        q = unit_quat(rng, yaw_only=(k % 3 == 0))
        t = rng.normal(size=3) * 2
        scale = rng.uniform(0.1, 0.8, size=3)
        fixedscale = [0.0, 0.0, 0.0] if k % 6 != 4 else rng.uniform(0.2, 0.5, size=3).tolist()
        upd = rng.normal(size=9) * [0.5, 0.05, 1e-3, 1e-9][k % 4]
        if k % 12 == 7:
            upd[:3] = [0.0, 0.0, 3e-6]  # theta below the 1e-5 threshold of exptwist_norollpitch
        v = W.cuboid(q.tolist() + t.tolist(), scale, flags, fixedscale)
        W.oplus(v, upd)
        oqt, osc = W.cuboid_state(v)
        cub.append({"qt": q.tolist() + t.tolist(), "scale": scale.tolist(), "flags": flags, "fixedscale": list(fixedscale), "update": upd.tolist(),
                    "out_qt": oqt, "out_scale": osc})
    out["cases"]["cuboid_oplus"] = cub
    # ---- exptwist_norollpitch (returns SE3Quat {Quaterniond, Vector3d} through the hidden pointer) ----
    f = W.R.fn("g2o::exptwist_norollpitch(Eigen::Matrix<double, 6, 1, 0, 6, 1> const&)", C.c_void_p, C.c_void_p, restype=C.c_void_p)
    ex = []
    for k in range(24):
        u = rng.normal(size=6) * [1.0, 0.1, 1e-3, 1e-6][k % 4]
        if k % 2:
            u[0] = u[1] = 0.0  # what VertexCuboid::oplusImpl passes
        buf = O.aligned(128)
        f(buf.ctypes.data, np.ascontiguousarray(u).ctypes.data)
        ex.append({"update": u.tolist(), "qt": buf[:7].tolist()})
    out["cases"]["exptwist_norollpitch"] = ex
    # ---- the custom edges ----
    for kind in KINDS:
        cases = []
        for k in range(32 if kind != "cubpl" else 8):
            c = edge_inputs(kind, k, rng)
            res = drive_edge(W, c)
            c.pop("noise", None)
            c.update(res)
            cases.append(c)
        out["cases"]["edge_" + kind] = cases
    info = loader.info()
    assert info["armed_hits"] == 0
    out["trap"] = {"symbols_trapped": info["trap_symbols"], "sink_hits_during_pinned_calls": info["armed_hits"]}
    os.makedirs(os.path.dirname(GOLD), exist_ok=True)
    with open(GOLD, "w") as fh:
        json.dump(out, fh, indent=0)
    print("wrote", GOLD, {k: len(v) for k, v in out["cases"].items()}, out["trap"])


# ---------------------------------------------------------------------------------------------------------------------------------
# the oracle side (also imported by tests/test_orbslam_pin.py)
# ---------------------------------------------------------------------------------------------------------------------------------
_EMPTY = ["mono_point", "mono_pose", "mono_obs", "mono_info", "mono_huber", "stereo_point", "stereo_pose", "stereo_obs", "stereo_info", "stereo_huber",
          "pl_plane", "pl_pose", "pl_kind", "pl_meas", "pl_info", "pl_huber", "bbox_pose", "bbox_cuboid", "bbox_meas", "bbox_info", "bbox_huber", "bbox_K",
          "corner_pose", "corner_cuboid", "corner_meas", "corner_info", "corner_huber", "corner_K", "ptcub_cuboid", "ptcub_margin_ratio", "ptcub_info",
          "ptcub_prior_half", "ptcub_huber", "cubpl_cuboid", "cubpl_plane", "cubpl_meas", "cubpl_info", "cubpl_huber"]
_ETYPE = {"plane": 0, "plane_ver": 0, "plane_par": 0, "bbox": 3, "corner": 4, "ptcub": 5, "cubpl": 6, "cuboid3d": 7}
_VT = {"pose": 0, "point": 1, "plane": 2, "cuboid": 3}


def dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def oracle_edge_case(oracle_py, c):
    """The oracle on the inputs of one edge case; returns the same output dictionary as drive_edge."""
    B = oracle_py.B
    lib = oracle_py.load()
    kind = c["kind"]
    P = B.Problem()
    for n_ in _EMPTY:
        P.set(n_, np.zeros(0))
    pose = c["v1"] if kind.startswith("plane") else (c["v0"] if kind in ("bbox", "corner", "cuboid3d") else [0, 0, 0, 1, 0, 0, 0.0])
    P.set("pose_qt", [pose]); P.set("pose_fixed", [c["fix1"] if kind.startswith("plane") else c["fix0"]]); P.set("pose_cam", [[480.0, 480.0, 320.0, 240.0, 40.0]])
    P.set("point_xyz", np.zeros((0, 3)))
    plane = c["v0"] if kind.startswith("plane") else (c["v1"] if kind == "cubpl" else None)
    P.set("plane_coef", [plane] if plane is not None else np.zeros((0, 4)))
    cub = c["v1"] if kind in ("bbox", "corner", "cuboid3d") else (c["v0"] if kind in ("ptcub", "cubpl") else None)
    if cub is not None:
        P.set("cuboid_qt", [cub[:7]]); P.set("cuboid_scale", [cub[7:]]); P.set("cuboid_flags", [c["flags"]]); P.set("cuboid_fixedscale", [c["fixedscale"]])
    else:
        P.set("cuboid_qt", np.zeros((0, 7))); P.set("cuboid_scale", np.zeros((0, 3)))
    hub = [c["huber"]]
    if kind.startswith("plane"):
        P.set("pl_plane", [0]); P.set("pl_pose", [0]); P.set("pl_kind", [["plane", "plane_ver", "plane_par"].index(kind)])
        P.set("pl_meas", [c["meas"]]); P.set("pl_info", [c["info"] + [0.0] * (3 - len(c["info"]))]); P.set("pl_huber", hub)
    elif kind in ("bbox", "corner"):
        P.set(kind + "_pose", [0]); P.set(kind + "_cuboid", [0]); P.set(kind + "_meas", [c["meas"]]); P.set(kind + "_info", [c["info"]])
        P.set(kind + "_huber", hub); P.set(kind + "_K", [c["K"]])
    elif kind == "ptcub":
        P.set("ptcub_cuboid", [0]); P.set("ptcub_offsets", [0, len(c["points"])]); P.set("ptcub_points", np.asarray(c["points"], dtype=float).reshape(-1, 3))
        P.set("ptcub_margin_ratio", [c["margin_ratio"]]); P.set("ptcub_prior_half", [c["prior_half"]]); P.set("ptcub_info", [c["info"]]); P.set("ptcub_huber", hub)
    elif kind == "cubpl":
        P.set("cubpl_cuboid", [0]); P.set("cubpl_plane", [0]); P.set("cubpl_meas", [c["meas"]]); P.set("cubpl_info", [c["info"]]); P.set("cubpl_huber", hub)
    elif kind == "cuboid3d":
        P.set("cub3d_pose", [0]); P.set("cub3d_cuboid", [0]); P.set("cub3d_meas", [c["meas"]]); P.set("cub3d_info", [c["info"]]); P.set("cub3d_huber", hub)
    opt = B.default_options()
    s = oracle_py.backend().create(P, opt)
    try:
        # the exact numbers the pinning script wrote into the reference's vertices (q is not re-normalised there)
        lib.oracle_ba_set_state(s.h, dptr(np.array(pose, dtype=float)), None,
                                dptr(np.array(c.get("v0_normalized") or c.get("v1_normalized"), dtype=float)) if plane is not None else None,
                                dptr(np.array(cub[:7], dtype=float)) if cub is not None else None, dptr(np.array(cub[7:], dtype=float)) if cub is not None else None)
        chi = C.c_double()
        lib.oracle_ba_linearize(s.h, C.byref(chi))
        t = _ETYPE[kind]
        D = len(c["error"])
        slot = {"plane": ("plane", "pose"), "bbox": ("pose", "cuboid"), "corner": ("pose", "cuboid"), "cuboid3d": ("pose", "cuboid"), "ptcub": ("cuboid", None),
                "cubpl": ("cuboid", "plane")}[kind if not kind.startswith("plane") else "plane"]
        dims = {"pose": 6, "plane": 3, "cuboid": 9}
        d0, d1 = dims[slot[0]], dims[slot[1]] if slot[1] else 0
        e, J0, J1 = np.zeros(16), np.zeros(16 * 9), np.zeros(16 * 9)
        lib.oracle_edge_eval(s.h, t, 0, dptr(e), dptr(J0), dptr(J1))
        out = {"error": e[:D].tolist(), "J0": J0[:D * d0].tolist(), "J1": J1[:D * d1].tolist() if d1 else None}
        info = np.array(c["info"], dtype=float) * np.ones(D)
        out["chi2"] = float(sum(e[r] * (info[r] * e[r]) for r in range(D)))  # BaseEdge::chi2 with a diagonal Omega (checked against the oracle's robust chi2 below)
        H0, b0, H1, b1, H01 = np.zeros(81), np.zeros(9), np.zeros(81), np.zeros(9), np.zeros(81)
        r0 = lib.oracle_ba_block(s.h, _VT[slot[0]], 0, -1, 0, dptr(H0), dptr(b0))
        out["H0"], out["b0"] = (H0[:d0 * d0] * (r0 > 0)).tolist(), (b0[:d0] * (r0 > 0)).tolist()
        if d1:
            r1 = lib.oracle_ba_block(s.h, _VT[slot[1]], 0, -1, 0, dptr(H1), dptr(b1))
            r01 = lib.oracle_ba_block(s.h, _VT[slot[0]], 0, _VT[slot[1]], 0, dptr(H01), None)
            out["H1"], out["b1"] = (H1[:d1 * d1] * (r1 > 0)).tolist(), (b1[:d1] * (r1 > 0)).tolist()
            out["H01"] = (H01[:d0 * d1] * (r01 > 0)).tolist()
        else:
            out["H1"] = out["b1"] = out["H01"] = None
        if kind.startswith("plane"):
            m = np.zeros(4)
            lib.oracle_ba_plane_meas(s.h, 0, dptr(m))
            out["meas_normalized"] = m.tolist()
        out["robust_chi2"] = chi.value
    finally:
        s.close()
    return out


def rel_dev(a, b, block=False):
    """max |a - b| / max(1, |b|) with NaN == NaN; block=True scales by the largest entry of the whole array (Jacobians, H and b
    blocks: an entry that is small next to its neighbours carries their rounding)."""
    a, b = np.atleast_1d(np.asarray(a, dtype=float)), np.atleast_1d(np.asarray(b, dtype=float))
    both_nan = np.isnan(a) & np.isnan(b)
    scale = np.maximum(1.0, np.nanmax(np.abs(b)) if (block and b.size and not np.isnan(b).all()) else np.abs(b))
    d = np.abs(a - b) / scale
    d[both_nan] = 0.0
    if np.isnan(d).any():
        return float("inf")
    return float(d.max()) if d.size else 0.0


def check(verbose=True):
    import oracle_py
    lib = oracle_py.load()
    gold = json.load(open(GOLD))["cases"]
    worst = {}

    def upd(key, dev):
        worst[key] = max(worst.get(key, 0.0), dev)
    dp = C.POINTER(C.c_double)
    lib.oracle_plane_normalize.argtypes = [dp, C.c_int, dp]
    lib.oracle_exptwist_norollpitch.argtypes = [dp, dp]
    for c in gold["plane_normalize"]:
        o = np.zeros(4)
        lib.oracle_plane_normalize(dptr(np.array(c["raw"])), 0, dptr(o))
        upd("plane_normalize", rel_dev(o, c["out"]))
        lib.oracle_plane_normalize(dptr(np.array(c["raw"])), 1, dptr(o))  # through Converter::toPlane3D's flip first: same plane
        upd("plane_normalize_via_converter", rel_dev(o, c["out"]))
    for c in gold["plane_oplus"]:
        o = np.zeros(4)
        lib.oracle_plane_oplus(dptr(np.array(c["coef"])), dptr(np.array(c["update"])), dptr(o))
        upd("plane_oplus", rel_dev(o, c["out"]))
    for c in gold["cuboid_oplus"]:
        oq, os_ = np.zeros(7), np.zeros(3)
        lib.oracle_cuboid_oplus(dptr(np.array(c["qt"])), dptr(np.array(c["scale"])), dptr(np.array(c["update"])), c["flags"], dptr(np.array(c["fixedscale"])),
                                dptr(oq), dptr(os_))
        upd("cuboid_oplus_qt", rel_dev(oq, c["out_qt"]))
        upd("cuboid_oplus_scale", rel_dev(os_, c["out_scale"]))
    for c in gold["exptwist_norollpitch"]:
        o = np.zeros(7)
        lib.oracle_exptwist_norollpitch(dptr(np.array(c["update"])), dptr(o))
        upd("exptwist_norollpitch", rel_dev(o, c["qt"]))
    for kind in KINDS:
        for c in gold.get("edge_" + kind, []):
            if kind == "cuboid3d" and not hasattr(lib, "oracle_has_cuboid3d"):
                continue
            o = oracle_edge_case(oracle_py, c)
            for key in ("error", "chi2", "J0", "J1", "H0", "b0", "H1", "b1", "H01", "meas_normalized"):
                if c.get(key) is None or o.get(key) is None:
                    continue
                upd("edge_%s_%s" % (kind, key), rel_dev(o[key], c[key], block=key not in ("error", "chi2", "meas_normalized")))
    if verbose:
        print("max relative deviation oracle vs prebuilt libORB_SLAM2.so:")
        for k, v in worst.items():
            print("  %-40s %.3e" % (k, v))
    return worst


if __name__ == "__main__":
    if "--check-only" in sys.argv:
        if os.path.exists(GOLD):
            check()
    else:
        generate()
        check()
    sys.stdout.flush()
    os._exit(0)  # the reference's objects live in numpy buffers: skip every destructor
