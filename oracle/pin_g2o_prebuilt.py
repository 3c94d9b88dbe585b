#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: golden vectors from the reference's OWN prebuilt Thirdparty/g2o/lib/libg2o.so.

The reference cannot be compiled here (Eigen3 / OpenCV / PCL / Pangolin are absent) but its prebuilt g2o core library
loads, and the following entry points can be driven with plain pointers (layouts found as described, nothing is
assumed from headers that could not be compiled):
  g2o::SE3Quat::exp(Vector6d)                        returns the 7-double (qx qy qz qw tx ty tz) state by hidden pointer
  g2o::VertexSE3Expmap::oplusImpl(const double*)     object built by its exported constructor; the estimate is located by
                                                     scanning the object for the identity pose the constructor stores
  g2o::VertexSBAPointXYZ::oplusImpl(const double*)   estimate located by a probe update
  g2o::RobustKernelHuber::setDelta / robustify       on a zeroed object (both fields are written by setDelta)
  g2o::EdgeSE3ProjectXYZ::cam_project                fx fy cx cy at this+0x128..0x140 (read off the disassembly)
  g2o::EdgeStereoSE3ProjectXYZ::cam_project          fx fy cx cy at this+0x160..0x178, bf passed as const float&
  g2o::EdgeSE3ProjectXYZOnlyPose::computeError / linearizeOplus        (the unary edges of Optimizer::PoseOptimization)
  g2o::EdgeStereoSE3ProjectXYZOnlyPose::computeError / linearizeOplus  driven on hand-laid-out objects: _vertices.data() at
                                                     this+0x8, _measurement at +0xa0, _error at +0xd0 / +0x100, the Jacobian
                                                     map's data pointer at +0xe0 / +0x118, Xw at +0xf0 / +0x128, fx fy cx cy
                                                     (bf) at +0x108 / +0x140; the vertex keeps q at +0xb0 and t at +0xd0
                                                     (all read off the disassembly of the four functions)
  g2o::BaseUnaryEdge<2|3, ..., VertexSE3Expmap>::linearizeOplus(JacobianWorkspace&) / ::constructQuadraticForm()
                                                     (base_unary_edge.hpp:43-80) on the two OnlyPose edge classes with a
                                                     RobustKernelHuber attached.  NOTE what the binary holds: the exported
                                                     linearizeOplus(JacobianWorkspace&) is five instructions -- it stores
                                                     workspaceForVertex(0) into the Jacobian map (this+0xe0 / +0x118) and tail-calls
                                                     the VIRTUAL linearizeOplus() (vtable slot 0xe0), which both instantiations
                                                     override with their analytic form.  The NUMERIC body of
                                                     BaseUnaryEdge::linearizeOplus() (base_unary_edge.hpp:82-123) was never emitted
                                                     into libg2o.so (no symbol, no caller: `nm -C` lists every BaseUnaryEdge member
                                                     there is), so it cannot be pinned; it is line by line the Xi half of
                                                     BaseBinaryEdge::linearizeOplusXi, which IS pinned bit for bit below.
The custom plane / cuboid types live in lib/libORB_SLAM2.so, which does not load in this container (OpenCV 3.3, PCL 1.8,
Pangolin, VTK missing): they stay unpinned.

  python oracle/pin_g2o_prebuilt.py            regenerate tests/golden/g2o_prebuilt.json (needs /root/reference)
  python oracle/pin_g2o_prebuilt.py --check-only   compare the oracle with the committed file
"""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
GOLD = os.path.join(REPO, "tests", "golden", "g2o_prebuilt.json")
G2O = "/root/reference/Thirdparty/g2o/lib/libg2o.so"


def dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def generate():
    lib = C.CDLL(G2O)
    rng = np.random.default_rng(20260929)
    out = {"source": "Thirdparty/g2o/lib/libg2o.so (prebuilt, gcc 7.5 / Ubuntu 18.04, Eigen as compiled in)", "cases": {}}

    # ---- SE3Quat::exp ---------------------------------------------------------------------------------------------
    f = lib._ZN3g2o7SE3Quat3expERKN5Eigen6MatrixIdLi6ELi1ELi0ELi6ELi1EEE
    f.argtypes = [C.c_void_p, C.c_void_p]
    f.restype = C.c_void_p
    cases = []
    ups = [rng.normal(size=6) * s for s in (1.0, 0.3, 0.05, 1e-3, 1e-7, 1e-9) for _ in range(6)]
    ups += [np.array([1e-9, 0, 0, 0, 0, 0.0]), np.array([0, 0, 3.0, 1.0, 2.0, 3.0]), np.array([2.0, -2.0, 1.0, 0, 0, 0.0])]
    for u in ups:
        buf = np.zeros(16)
        f(buf.ctypes.data, np.ascontiguousarray(u).ctypes.data)
        cases.append({"update": u.tolist(), "qt": buf[:7].tolist()})  # Quaterniond coeffs x y z w, then translation
    out["cases"]["se3_exp"] = cases

    # ---- VertexSE3Expmap::oplusImpl -----------------------------------------------------------------------------------
    ctor = lib._ZN3g2o15VertexSE3ExpmapC1Ev
    ctor.argtypes = [C.c_void_p]
    opl = lib._ZN3g2o15VertexSE3Expmap9oplusImplEPKd
    opl.argtypes = [C.c_void_p, C.c_void_p]
    obj = np.zeros(1024)  # 8 KB, far larger than the object
    ctor(obj.ctypes.data)
    ident = np.array([0, 0, 0, 1, 0, 0, 0.0])
    off = [i for i in range(0, 400) if np.array_equal(obj[i:i + 7], ident)]
    assert len(off) >= 1, "estimate not found"
    eo = off[0]
    cases = []
    for k in range(24):
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        if q[3] < 0:
            q = -q
        t = rng.normal(size=3) * 3
        u = rng.normal(size=6) * [1.0, 0.1, 1e-3, 1e-9][k % 4]
        obj[eo:eo + 4] = q
        obj[eo + 4:eo + 7] = t
        opl(obj.ctypes.data, np.ascontiguousarray(u).ctypes.data)
        cases.append({"qt": q.tolist() + t.tolist(), "update": u.tolist(), "out": obj[eo:eo + 7].tolist()})
    out["cases"]["pose_oplus"] = cases

    # ---- RobustKernelHuber ---------------------------------------------------------------------------------------------
    sd = lib._ZN3g2o17RobustKernelHuber8setDeltaEd
    sd.argtypes = [C.c_void_p, C.c_double]
    rb = lib._ZNK3g2o17RobustKernelHuber9robustifyEdRN5Eigen6MatrixIdLi3ELi1ELi0ELi3ELi1EEE
    rb.argtypes = [C.c_void_p, C.c_double, C.c_void_p]
    cases = []
    for delta in (float(np.float32(np.sqrt(5.991))), float(np.float32(np.sqrt(7.815))), float(np.float32(np.sqrt(500.0))), 1.0):
        k = np.zeros(8)
        sd(k.ctypes.data, delta)
        for e in (0.0, 0.5 * delta * delta, delta * delta, delta * delta * (1 + 1e-12), 2 * delta * delta, 1e4, 1e7):
            rho = np.zeros(4)
            rb(k.ctypes.data, e, rho.ctypes.data)
            cases.append({"delta": delta, "e": e, "rho": rho[:3].tolist()})
    out["cases"]["huber"] = cases

    # ---- cam_project (mono / stereo) ----------------------------------------------------------------------------------------
    cm = lib._ZNK3g2o17EdgeSE3ProjectXYZ11cam_projectERKN5Eigen6MatrixIdLi3ELi1ELi0ELi3ELi1EEE
    cm.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    cs = lib._ZNK3g2o23EdgeStereoSE3ProjectXYZ11cam_projectERKN5Eigen6MatrixIdLi3ELi1ELi0ELi3ELi1EEERKf
    cs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    mono, stereo = [], []
    for _ in range(32):
        fx, fy, cx, cy = rng.uniform(300, 700), rng.uniform(300, 700), rng.uniform(300, 340), rng.uniform(220, 260)
        X = np.array([rng.normal() * 2, rng.normal() * 2, rng.uniform(0.3, 9.0)])
        this = np.zeros(64)
        this[0x128 // 8:0x128 // 8 + 4] = [fx, fy, cx, cy]
        res = np.zeros(4)
        cm(res.ctypes.data, this.ctypes.data, X.ctypes.data)
        mono.append({"cam": [fx, fy, cx, cy], "X": X.tolist(), "uv": res[:2].tolist()})
        this = np.zeros(64)
        this[0x160 // 8:0x160 // 8 + 4] = [fx, fy, cx, cy]
        bf = np.array([rng.uniform(20, 60)], dtype=np.float32)
        res = np.zeros(4)
        cs(res.ctypes.data, this.ctypes.data, X.ctypes.data, bf.ctypes.data)
        stereo.append({"cam": [fx, fy, cx, cy, float(bf[0])], "X": X.tolist(), "uvr": res[:3].tolist()})
    out["cases"]["mono_project"] = mono
    out["cases"]["stereo_project"] = stereo


    # ---- unary OnlyPose edges: computeError + linearizeOplus --------------------------------------------------------------
    def aligned(n):
        a = np.zeros(n + 8)
        o = (-a.ctypes.data // 8) % 2  # 16-byte alignment: computeError uses movapd / movaps on the object
        return a[o:o + n]
    only = {"mono": [], "stereo": []}
    fn = {
        "mono": (lib._ZN3g2o25EdgeSE3ProjectXYZOnlyPose12computeErrorEv, lib._ZN3g2o25EdgeSE3ProjectXYZOnlyPose14linearizeOplusEv,
                 dict(meas=0xa0, err=0xd0, jac=0xe0, xw=0xf0, cam=0x108, dim=2)),
        "stereo": (lib._ZN3g2o31EdgeStereoSE3ProjectXYZOnlyPose12computeErrorEv, lib._ZN3g2o31EdgeStereoSE3ProjectXYZOnlyPose14linearizeOplusEv,
                   dict(meas=0xa0, err=0x100, jac=0x118, xw=0x128, cam=0x140, dim=3)),
    }
    for kind, (f_err, f_lin, L) in fn.items():
        f_err.argtypes = [C.c_void_p]; f_lin.argtypes = [C.c_void_p]
        f_err.restype = None; f_lin.restype = None
        for k in range(24):
            q = rng.normal(size=4); q /= np.linalg.norm(q)
            if q[3] < 0:
                q = -q
            t = rng.normal(size=3)
            # a point in front of the camera: choose it in the camera frame and move it to the world
            pc = np.array([rng.normal() * 1.5, rng.normal() * 1.5, rng.uniform(0.3, 9.0)])
            x, y, z, w = q
            R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                          [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                          [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
            Xw = R.T @ (pc - t)
            cam = [rng.uniform(300, 700), rng.uniform(300, 700), rng.uniform(300, 340), rng.uniform(220, 260), rng.uniform(20, 60)]
            obs = [rng.uniform(0, 640), rng.uniform(0, 480), rng.uniform(0, 640)]
            vert = aligned(64)
            vert[0xb0 // 8:0xb0 // 8 + 4] = q
            vert[0xd0 // 8:0xd0 // 8 + 3] = t
            vptrs = np.array([vert.ctypes.data], dtype=np.uint64)
            this = aligned(64)
            assert this.ctypes.data % 16 == 0
            J = np.zeros(18)
            this.view(np.uint64)[0x8 // 8] = vptrs.ctypes.data
            this.view(np.uint64)[L["jac"] // 8] = J.ctypes.data
            this[L["meas"] // 8:L["meas"] // 8 + L["dim"]] = obs[:L["dim"]]
            this[L["xw"] // 8:L["xw"] // 8 + 3] = Xw
            ncam = 4 if kind == "mono" else 5
            this[L["cam"] // 8:L["cam"] // 8 + ncam] = cam[:ncam]
            f_err(this.ctypes.data)
            f_lin(this.ctypes.data)
            err = this[L["err"] // 8:L["err"] // 8 + L["dim"]].tolist()
            Jrow = J[:6 * L["dim"]].reshape(6, L["dim"]).T  # Eigen stores the dim x 6 matrix column-major
            only[kind].append({"qt": q.tolist() + t.tolist(), "cam": cam, "xw": Xw.tolist(), "obs": obs, "error": err, "J": Jrow.reshape(-1).tolist()})
    out["cases"]["onlypose_mono"] = only["mono"]
    out["cases"]["onlypose_stereo"] = only["stereo"]

    out["cases"].update(generate_binary_edges(lib))
    out["cases"].update(generate_unary_edges(lib))

    os.makedirs(os.path.dirname(GOLD), exist_ok=True)
    with open(GOLD, "w") as fh:
        json.dump(out, fh, indent=1)
    print("wrote", GOLD, {k: len(v) for k, v in out["cases"].items()})


# ---- binary reprojection edges: computeError, analytic linearizeOplus, NUMERIC linearizeOplusXi/Xj, constructQuadraticForm ----
# Real objects this time: the edge and both vertices are built by their exported constructors (so every virtual call the
# numeric differentiation makes -- computeError, push, oplus, pop -- goes through the reference's own vtables), then the
# fields are written at the offsets read off the disassembly:
#   vertex   _fixed +0x54 (constructQuadraticForm tests it), Hessian map pointer +0x70 (BaseVertex::mapHessianMemory),
#            _b +0x80 (BaseVertex::b), point estimate +0x98, pose q +0xb0 / t +0xd0
#   edge     _vertices.data() at *(this+0x8), robust kernel pointer +0x30, _measurement +0xa0,
#            mono (D = 2): _information +0xb0, _error +0xd0, Jacobian map pointers +0x108 / +0x118, fx fy cx cy +0x128
#            stereo (D = 3): _information +0xb8, _error +0x100, Jacobian map pointers +0x140 / +0x150, fx fy cx cy bf +0x160
#            the off-diagonal Hessian map is set through the exported mapHessianMemory(double*, int, int, bool rowMajor)
#   kernel   RobustKernelHuber = exported vtable + 16, _delta / dsqr written by the exported setDelta
_BB2 = "_ZN3g2o14BaseBinaryEdgeILi2EN5Eigen6MatrixIdLi2ELi1ELi0ELi2ELi1EEENS_17VertexSBAPointXYZENS_15VertexSE3ExpmapEE"
_BB3 = "_ZN3g2o14BaseBinaryEdgeILi3EN5Eigen6MatrixIdLi3ELi1ELi0ELi3ELi1EEENS_17VertexSBAPointXYZENS_15VertexSE3ExpmapEE"
_KIND = {
    "mono": dict(D=2, ctor="_ZN3g2o17EdgeSE3ProjectXYZC1Ev", err_fn="_ZN3g2o17EdgeSE3ProjectXYZ12computeErrorEv",
                 lin="_ZN3g2o17EdgeSE3ProjectXYZ14linearizeOplusEv", xi=_BB2 + "16linearizeOplusXiEv", xj=_BB2 + "16linearizeOplusXjEv",
                 cqf=_BB2 + "22constructQuadraticFormEv", mhm=_BB2 + "16mapHessianMemoryEPdiib",
                 meas=0xa0, info=0xb0, err=0xd0, jxi=0x108, jxj=0x118, cam=0x128, ncam=4),
    "stereo": dict(D=3, ctor="_ZN3g2o23EdgeStereoSE3ProjectXYZC1Ev", err_fn="_ZN3g2o23EdgeStereoSE3ProjectXYZ12computeErrorEv",
                   lin="_ZN3g2o23EdgeStereoSE3ProjectXYZ14linearizeOplusEv", xi=_BB3 + "16linearizeOplusXiEv", xj=_BB3 + "16linearizeOplusXjEv",
                   cqf=_BB3 + "22constructQuadraticFormEv", mhm=_BB3 + "16mapHessianMemoryEPdiib",
                   meas=0xa0, info=0xb8, err=0x100, jxi=0x140, jxj=0x150, cam=0x160, ncam=5),
}


def _aligned(nbytes):
    a = np.zeros(nbytes // 8 + 4)
    o = (-a.ctypes.data // 8) % 2
    return a[o:o + nbytes // 8]


def _rot(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def drive_binary_edge(lib, kind, q, t, Xw, cam, obs, w, delta, fix_pt, fix_ps, row_major):
    """One edge through the prebuilt library; returns every array row-major."""
    vp = C.c_void_p
    u64 = lambda a: a.view(np.uint64)  # noqa: E731

    def fn(name, *argtypes):
        f = getattr(lib, name)
        f.argtypes = list(argtypes)
        f.restype = None
        return f
    K = _KIND[kind]
    D = K["D"]
    e, pt, ps = _aligned(0x200), _aligned(0x200), _aligned(0x200)
    fn(K["ctor"], vp)(e.ctypes.data)
    fn("_ZN3g2o17VertexSBAPointXYZC1Ev", vp)(pt.ctypes.data)
    fn("_ZN3g2o15VertexSE3ExpmapC1Ev", vp)(ps.ctypes.data)
    pt[0x98 // 8:0x98 // 8 + 3] = Xw
    pt.view(np.uint8)[0x54] = fix_pt
    ps[0xb0 // 8:0xb0 // 8 + 4] = q
    ps[0xd0 // 8:0xd0 // 8 + 3] = t
    ps.view(np.uint8)[0x54] = fix_ps
    vptr = C.cast(int(u64(e)[1]), C.POINTER(C.c_uint64))  # _vertices was resized to 2 by the constructor
    vptr[0] = pt.ctypes.data
    vptr[1] = ps.ctypes.data
    e[K["meas"] // 8:K["meas"] // 8 + D] = obs[:D]
    e[K["info"] // 8:K["info"] // 8 + D * D] = (np.eye(D) * w).reshape(-1)
    e[K["cam"] // 8:K["cam"] // 8 + K["ncam"]] = cam[:K["ncam"]]
    kern = None
    if delta > 0:
        kern = _aligned(64)
        u64(kern)[0] = C.addressof(C.c_void_p.in_dll(lib, "_ZTVN3g2o17RobustKernelHuberE")) + 16
        f = lib._ZN3g2o17RobustKernelHuber8setDeltaEd
        f.argtypes = [vp, C.c_double]
        f(kern.ctypes.data, delta)
        u64(e)[0x30 // 8] = kern.ctypes.data
    Jxi, Jxj = _aligned(8 * 3 * D + 64), _aligned(8 * 6 * D + 64)
    u64(e)[K["jxi"] // 8] = Jxi.ctypes.data
    u64(e)[K["jxj"] // 8] = Jxj.ctypes.data
    fn(K["err_fn"], vp)(e.ctypes.data)
    err = e[K["err"] // 8:K["err"] // 8 + D].copy()
    fn(K["lin"], vp)(e.ctypes.data)  # Eigen stores the D x Di Jacobians column-major
    Ja_i, Ja_j = Jxi[:3 * D].reshape(3, D).T.copy(), Jxj[:6 * D].reshape(6, D).T.copy()
    Hii, Hjj, Hij = _aligned(8 * 9 + 64), _aligned(8 * 36 + 64), _aligned(8 * 18 + 64)
    u64(pt)[0x70 // 8] = Hii.ctypes.data
    u64(ps)[0x70 // 8] = Hjj.ctypes.data
    f = getattr(lib, K["mhm"])
    f.argtypes = [vp, vp, C.c_int, C.c_int, C.c_bool]
    f.restype = None
    f(e.ctypes.data, Hij.ctypes.data, 0, 1, bool(row_major))
    fn(K["cqf"], vp)(e.ctypes.data)
    bi, bj = pt[0x80 // 8:0x80 // 8 + 3].copy(), ps[0x80 // 8:0x80 // 8 + 6].copy()
    Hii_m, Hjj_m = Hii[:9].reshape(3, 3).T.copy(), Hjj[:36].reshape(6, 6).T.copy()
    # _hessian is Di x Dj = 3 x 6 column-major; with rowMajor the 6 x 3 transposed map is written instead
    Hij_m = Hij[:18].reshape(3, 6).copy() if row_major else Hij[:18].reshape(6, 3).T.copy()
    Jxi[:] = 0
    Jxj[:] = 0
    fn(K["xi"], vp)(e.ctypes.data)
    fn(K["xj"], vp)(e.ctypes.data)
    Jn_i, Jn_j = Jxi[:3 * D].reshape(3, D).T.copy(), Jxj[:6 * D].reshape(6, D).T.copy()
    err_after = e[K["err"] // 8:K["err"] // 8 + D].copy()
    assert np.array_equal(err, err_after)  # base_binary_edge.hpp:258 / :318 restore _error
    flat = lambda m: np.asarray(m).reshape(-1).tolist()  # noqa: E731
    return {"error": err.tolist(), "J_point": flat(Ja_i), "J_pose": flat(Ja_j), "Jnum_point": flat(Jn_i), "Jnum_pose": flat(Jn_j),
            "H_point": flat(Hii_m), "H_pose": flat(Hjj_m), "H_point_pose": flat(Hij_m), "b_point": bi.tolist(), "b_pose": bj.tolist()}


def generate_binary_edges(lib):
    rng = np.random.default_rng(20260930)  # its own stream: the vectors above stay bit-identical
    out = {"binary_mono": [], "binary_stereo": []}
    for kind in ("mono", "stereo"):
        for k in range(32):
            q = rng.normal(size=4)
            q /= np.linalg.norm(q)
            if q[3] < 0:
                q = -q
            t = rng.normal(size=3)
            pc = np.array([rng.normal() * 1.2, rng.normal() * 0.9, rng.uniform(0.4, 9.0)])
            Xw = _rot(q).T @ (pc - t)
            cam = [rng.uniform(300, 700), rng.uniform(300, 700), rng.uniform(300, 340), rng.uniform(220, 260), float(np.float32(rng.uniform(20, 60)))]
            proj = np.array([cam[0] * pc[0] / pc[2] + cam[2], cam[1] * pc[1] / pc[2] + cam[3], 0.0])
            proj[2] = proj[0] - cam[4] / pc[2]
            # residuals below the Huber knee, around it and far above it (gross outliers)
            mag = [0.5, 1.0, 2.0, 4.0, 30.0, 300.0][k % 6]
            obs = proj + rng.normal(size=3) * mag
            w = float(1.2 ** (-2 * int(rng.integers(0, 8))))  # ORB pyramid inverse variances, Optimizer.cc:2376-2377
            delta = 0.0 if k % 4 == 3 else float(np.float32(np.sqrt(5.991 if kind == "mono" else 7.815)))  # Optimizer.cc:2326-2327 (const float)
            fix_pt, fix_ps = (1, 0) if k % 8 == 5 else ((0, 1) if k % 8 == 6 else (0, 0))
            row_major = bool(k % 2)
            res = drive_binary_edge(lib, kind, q, t, Xw, cam, obs, w, delta, fix_pt, fix_ps, row_major)
            res.update({"qt": q.tolist() + t.tolist(), "cam": cam, "xw": Xw.tolist(), "obs": obs.tolist(), "info": w, "huber_delta": delta,
                        "fix_point": fix_pt, "fix_pose": fix_ps, "row_major": int(row_major)})
            out["binary_" + kind].append(res)
    return out


# ---- unary OnlyPose edges through BaseUnaryEdge::linearizeOplus(JacobianWorkspace&) and ::constructQuadraticForm() ------------
# The OnlyPose edge classes export no constructor (inline in the header), so the object is laid out by hand on top of the exported
# OptimizableGraph::Edge::Edge(): vtable pointer (exported vtable + 16), _vertices = one slot, _dimension (+0x24) = D, the robust
# kernel at +0x30, measurement / information / error / Jacobian map / Xw / intrinsics at the offsets listed in the module docstring
# (information: +0xb0 for D = 2, +0xb8 for D = 3, the BaseEdge<D> layout of the binary edges above).  The pose vertex is built by
# its exported constructor; Hessian map pointer +0x70, _b +0x80, _fixed +0x54.
_BU = {2: "_ZN3g2o13BaseUnaryEdgeILi2EN5Eigen6MatrixIdLi2ELi1ELi0ELi2ELi1EEENS_15VertexSE3ExpmapEE",
       3: "_ZN3g2o13BaseUnaryEdgeILi3EN5Eigen6MatrixIdLi3ELi1ELi0ELi3ELi1EEENS_15VertexSE3ExpmapEE"}
_UNARY = {
    "mono": dict(D=2, vt="_ZTVN3g2o25EdgeSE3ProjectXYZOnlyPoseE", err_fn="_ZN3g2o25EdgeSE3ProjectXYZOnlyPose12computeErrorEv",
                 chi2="_ZNK3g2o8BaseEdgeILi2EN5Eigen6MatrixIdLi2ELi1ELi0ELi2ELi1EEEE4chi2Ev", meas=0xa0, info=0xb0, err=0xd0, jac=0xe0, xw=0xf0, cam=0x108, ncam=4),
    "stereo": dict(D=3, vt="_ZTVN3g2o31EdgeStereoSE3ProjectXYZOnlyPoseE", err_fn="_ZN3g2o31EdgeStereoSE3ProjectXYZOnlyPose12computeErrorEv",
                   chi2="_ZNK3g2o8BaseEdgeILi3EN5Eigen6MatrixIdLi3ELi1ELi0ELi3ELi1EEEE4chi2Ev", meas=0xa0, info=0xb8, err=0x100, jac=0x118, xw=0x128, cam=0x140, ncam=5),
}


def make_unary_edge(lib, kind, vertex, xw, cam, obs, w, delta, keep):
    """A hand-laid-out OnlyPose edge attached to `vertex`; returns (edge buffer, Jacobian buffer)."""
    vp = C.c_void_p
    K = _UNARY[kind]
    D = K["D"]
    e = _aligned(0x200)
    f = lib._ZN3g2o16OptimizableGraph4EdgeC1Ev
    f.argtypes = [vp]
    f.restype = None
    f(e.ctypes.data)
    u64 = e.view(np.uint64)
    u64[0] = C.addressof(C.c_void_p.in_dll(lib, K["vt"])) + 16
    slots = np.array([vertex.ctypes.data], dtype=np.uint64)  # _vertices: begin / end / end of storage
    u64[1], u64[2], u64[3] = slots.ctypes.data, slots.ctypes.data + 8, slots.ctypes.data + 8
    e.view(np.int32)[0x24 // 4] = D
    e[K["meas"] // 8:K["meas"] // 8 + D] = obs[:D]
    e[K["info"] // 8:K["info"] // 8 + D * D] = (np.eye(D) * w).reshape(-1)
    e[K["xw"] // 8:K["xw"] // 8 + 3] = xw
    e[K["cam"] // 8:K["cam"] // 8 + K["ncam"]] = cam[:K["ncam"]]
    kern = None
    if delta > 0:
        kern = _aligned(64)
        kern.view(np.uint64)[0] = C.addressof(C.c_void_p.in_dll(lib, "_ZTVN3g2o17RobustKernelHuberE")) + 16
        sd = lib._ZN3g2o17RobustKernelHuber8setDeltaEd
        sd.argtypes = [vp, C.c_double]
        sd(kern.ctypes.data, delta)
        u64[0x30 // 8] = kern.ctypes.data
    J = _aligned(8 * 6 * D + 64)
    keep.extend([e, slots, kern, J])
    return e, J


def fake_jacobian_workspace(J, keep):
    """What linearizeOplus(JacobianWorkspace&) dereferences: _workspace.begin() -> VectorXd{data, size} -> data."""
    vec0 = np.array([J.ctypes.data, J.size], dtype=np.uint64)
    ws = np.array([vec0.ctypes.data, vec0.ctypes.data + 16, vec0.ctypes.data + 16, 0], dtype=np.uint64)
    keep.extend([vec0, ws])
    return ws


def drive_unary_edge(lib, kind, q, t, xw, cam, obs, w, delta, fix_pose):
    vp = C.c_void_p
    K = _UNARY[kind]
    D = K["D"]
    keep = []

    def fn(name, *argtypes, restype=None):
        f = getattr(lib, name)
        f.argtypes = list(argtypes)
        f.restype = restype
        return f
    ps = _aligned(0x200)
    fn("_ZN3g2o15VertexSE3ExpmapC1Ev", vp)(ps.ctypes.data)
    ps[0xb0 // 8:0xb0 // 8 + 4] = q
    ps[0xd0 // 8:0xd0 // 8 + 3] = t
    ps.view(np.uint8)[0x54] = fix_pose
    e, J = make_unary_edge(lib, kind, ps, xw, cam, obs, w, delta, keep)
    fn(K["err_fn"], vp)(e.ctypes.data)
    err = e[K["err"] // 8:K["err"] // 8 + D].copy()
    chi2 = float(fn(K["chi2"], vp, restype=C.c_double)(e.ctypes.data))
    ws = fake_jacobian_workspace(J, keep)
    fn(_BU[D] + "14linearizeOplusERNS_17JacobianWorkspaceE", vp, vp)(e.ctypes.data, ws.ctypes.data)
    assert int(e.view(np.uint64)[K["jac"] // 8]) == J.ctypes.data  # the wrapper mapped the workspace
    Jm = J[:6 * D].reshape(6, D).T.copy()
    H = _aligned(8 * 36 + 64)
    ps.view(np.uint64)[0x70 // 8] = H.ctypes.data
    fn(_BU[D] + "22constructQuadraticFormEv", vp)(e.ctypes.data)
    b = ps[0x80 // 8:0x80 // 8 + 6].copy()
    Hm = H[:36].reshape(6, 6).T.copy()
    flat = lambda m: np.asarray(m).reshape(-1).tolist()  # noqa: E731
    return {"error": err.tolist(), "chi2": chi2, "J": flat(Jm), "H_pose": flat(Hm), "b_pose": b.tolist()}


def generate_unary_edges(lib):
    rng = np.random.default_rng(20261001)  # its own stream: every vector above stays bit-identical
    out = {"unary_mono": [], "unary_stereo": []}
    for kind in ("mono", "stereo"):
        for k in range(32):
            q = rng.normal(size=4)
            q /= np.linalg.norm(q)
            if q[3] < 0:
                q = -q
            t = rng.normal(size=3)
            pc = np.array([rng.normal() * 1.2, rng.normal() * 0.9, rng.uniform(0.4, 9.0)])
            Xw = (_rot(q).T @ (pc - t)).astype(np.float32).astype(np.float64)  # Optimizer.cc:318-320: Xw.at<float>
            pc = _rot(q) @ Xw + t
            cam = [rng.uniform(300, 700), rng.uniform(300, 700), rng.uniform(300, 340), rng.uniform(220, 260), float(np.float32(rng.uniform(20, 60)))]
            proj = np.array([cam[0] * pc[0] / pc[2] + cam[2], cam[1] * pc[1] / pc[2] + cam[3], 0.0])
            proj[2] = proj[0] - cam[4] / pc[2]
            delta = 0.0 if k % 4 == 3 else float(np.float32(np.sqrt(5.991 if kind == "mono" else 7.815)))  # Optimizer.cc:282-283 (const float)
            w = float(np.float32(1.2 ** (-2 * int(rng.integers(0, 8)))))  # Optimizer.cc:308: const float invSigma2
            at_knee = k % 8 == 4 and delta > 0
            mag = [0.3, 1.0, 2.0, 4.0, 30.0, 300.0][k % 6]  # below the Huber knee, around it, far above it
            obs = proj + rng.normal(size=3) * (1.0 if at_knee else mag)
            if kind == "stereo" and obs[2] < 0:
                obs[2] = abs(obs[2])  # a negative right coordinate MEANS "monocular" to the caller (Optimizer.cc:294)
            if at_knee:
                # AT the knee: take the residual the library itself computes for this observation and rescale it so that
                # chi2 = e' (w I) e lands within a few ulps of delta^2 (the projection does not depend on the observation)
                e0 = np.array(drive_unary_edge(lib, kind, q, t, Xw, cam, obs, w, delta, 0)["error"])
                scale = delta / np.sqrt(w * float(e0 @ e0))
                obs[:len(e0)] = obs[:len(e0)] - e0 + e0 * scale
            fix_pose = 1 if k % 16 == 9 else 0
            res = drive_unary_edge(lib, kind, q, t, Xw, cam, obs, w, delta, fix_pose)
            res.update({"qt": q.tolist() + t.tolist(), "cam": cam, "xw": Xw.tolist(), "obs": obs.tolist(), "info": w, "huber_delta": delta, "fix_pose": fix_pose})
            out["unary_" + kind].append(res)
    return out


def check_unary_edges(lib, gold, worst):
    """Oracle (oracle_unary_edge_pin) against the unary-edge vectors; the same comparison tests/test_golden.py makes."""
    c_dp = C.POINTER(C.c_double)
    lib.oracle_unary_edge_pin.argtypes = [c_dp] * 4 + [C.c_int, C.c_double, C.c_double, C.c_int] + [c_dp] * 5
    lib.oracle_unary_edge_pin.restype = C.c_int
    for kind in ("mono", "stereo"):
        for c in gold.get("unary_" + kind, []):
            D = len(c["error"])
            o = {"error": np.zeros(3), "chi2": np.zeros(1), "J": np.zeros(18), "H_pose": np.zeros(36), "b_pose": np.zeros(6)}
            lib.oracle_unary_edge_pin(dptr(np.array(c["qt"])), dptr(np.array(c["cam"])), dptr(np.array(c["xw"])), dptr(np.array(c["obs"])),
                                      int(kind == "stereo"), c["info"], c["huber_delta"], c["fix_pose"], *[dptr(o[k]) for k in ("error", "chi2", "J", "H_pose", "b_pose")])
            for k, n in (("error", D), ("chi2", 1), ("J", 6 * D), ("H_pose", 36), ("b_pose", 6)):
                ref = np.atleast_1d(np.array(c[k], dtype=float))
                dev = float((np.abs(o[k][:n] - ref) / np.maximum(1.0, np.abs(ref))).max())
                key = "unary_%s_%s" % (kind, k)
                worst[key] = max(worst.get(key, 0.0), dev)
    return worst


def check_binary_edges(lib, gold, worst):
    """Oracle (oracle_binary_edge_pin) against the binary-edge vectors; the same comparison tests/test_golden.py makes."""
    c_dp = C.POINTER(C.c_double)
    lib.oracle_binary_edge_pin.argtypes = [c_dp] * 4 + [C.c_int, C.c_double, C.c_double, C.c_int, C.c_int] + [c_dp] * 10
    lib.oracle_binary_edge_pin.restype = C.c_int
    for kind in ("mono", "stereo"):
        for c in gold.get("binary_" + kind, []):
            D = len(c["error"])
            o = {"error": np.zeros(3), "J_point": np.zeros(9), "J_pose": np.zeros(18), "Jnum_point": np.zeros(9), "Jnum_pose": np.zeros(18),
                 "H_point": np.zeros(9), "H_pose": np.zeros(36), "H_point_pose": np.zeros(18), "b_point": np.zeros(3), "b_pose": np.zeros(6)}
            lib.oracle_binary_edge_pin(dptr(np.array(c["qt"])), dptr(np.array(c["cam"])), dptr(np.array(c["xw"])), dptr(np.array(c["obs"])),
                                       int(kind == "stereo"), c["info"], c["huber_delta"], c["fix_point"], c["fix_pose"],
                                       *[dptr(o[k]) for k in ("error", "J_point", "J_pose", "Jnum_point", "Jnum_pose", "H_point", "H_pose", "H_point_pose", "b_point", "b_pose")])
            for k, n in (("error", D), ("J_point", 3 * D), ("J_pose", 6 * D), ("Jnum_point", 3 * D), ("Jnum_pose", 6 * D), ("H_point", 9), ("H_pose", 36),
                         ("H_point_pose", 18), ("b_point", 3), ("b_pose", 6)):
                ref = np.array(c[k])
                dev = float((np.abs(o[k][:n] - ref) / np.maximum(1.0, np.abs(ref))).max())
                key = "binary_%s_%s" % (kind, k)
                worst[key] = max(worst.get(key, 0.0), dev)
    return worst


def check():
    """Oracle vs the committed vectors (the same comparison tests/test_golden.py runs under pytest)."""
    sys.path.insert(0, HERE)
    import oracle_py
    lib = oracle_py.load()
    gold = json.load(open(GOLD))["cases"]
    worst = {}
    for c in gold["se3_exp"]:
        o = np.zeros(7)
        lib.oracle_se3_exp(dptr(np.array(c["update"])), dptr(o))
        worst["se3_exp"] = max(worst.get("se3_exp", 0), float(np.abs(o - c["qt"]).max()))
    for c in gold["pose_oplus"]:
        o = np.zeros(7)
        lib.oracle_pose_oplus(dptr(np.array(c["qt"])), dptr(np.array(c["update"])), dptr(o))
        worst["pose_oplus"] = max(worst.get("pose_oplus", 0), float(np.abs(o - c["out"]).max()))
    for c in gold["huber"]:
        r = np.zeros(3)
        lib.oracle_huber(c["e"], c["delta"], dptr(r))
        worst["huber"] = max(worst.get("huber", 0), float(np.abs(r - c["rho"]).max() / max(1.0, np.abs(c["rho"]).max())))
    # reprojection through the oracle's edge evaluation: a 1-keyframe / 1-point graph with identity pose, e = obs - proj
    B = oracle_py.B
    be = oracle_py.backend()
    for kind, key in (("mono", "mono_project"), ("stereo", "stereo_project")):
        for c in gold[key]:
            P = B.Problem()
            cam = c["cam"] + ([0.0] if kind == "mono" else [])
            P.set("pose_qt", [[0, 0, 0, 1, 0, 0, 0]]); P.set("pose_fixed", [0]); P.set("pose_cam", [cam])
            P.set("point_xyz", [c["X"]])
            names = ["mono_point", "mono_pose", "mono_obs", "mono_info", "mono_huber", "stereo_point", "stereo_pose", "stereo_obs", "stereo_info", "stereo_huber"]
            for n_ in names:
                P.set(n_, np.zeros(0))
            P.set(kind + "_point", [0]); P.set(kind + "_pose", [0]); P.set(kind + "_obs", [[0.0] * (2 if kind == "mono" else 3)])
            P.set(kind + "_info", [1.0]); P.set(kind + "_huber", [0.0])
            for n_ in ("plane_coef", "cuboid_qt", "cuboid_scale", "pl_plane", "pl_pose", "pl_meas", "pl_info", "pl_huber", "bbox_pose", "bbox_cuboid",
                       "bbox_meas", "bbox_info", "bbox_huber", "bbox_K", "corner_pose", "corner_cuboid", "corner_meas", "corner_info", "corner_huber",
                       "corner_K", "ptcub_cuboid", "ptcub_margin_ratio", "ptcub_info", "cubpl_cuboid", "cubpl_plane", "cubpl_meas", "cubpl_info", "cubpl_huber"):
                P.set(n_, np.zeros(0))
            s_ = be.create(P, B.default_options())
            e = np.zeros(3)
            lib.oracle_edge_eval(s_.h, 1 if kind == "mono" else 2, 0, dptr(e), None, None)
            ref = np.array(c["uv"] if kind == "mono" else c["uvr"])
            worst[key] = max(worst.get(key, 0), float(np.abs(-e[:len(ref)] - ref).max()))
            s_.close()
    lib.oracle_onlypose_eval.argtypes = [C.POINTER(C.c_double)] * 4 + [C.c_int] + [C.POINTER(C.c_double)] * 2
    lib.oracle_onlypose_eval.restype = None
    for kind in ("mono", "stereo"):
        for c in gold.get("onlypose_" + kind, []):
            e, J = np.zeros(3), np.zeros(18)
            lib.oracle_onlypose_eval(dptr(np.array(c["qt"])), dptr(np.array(c["cam"])), dptr(np.array(c["xw"])), dptr(np.array(c["obs"])),
                                     int(kind == "stereo"), dptr(e), dptr(J))
            d = len(c["error"])
            worst["onlypose_" + kind + "_error"] = max(worst.get("onlypose_" + kind + "_error", 0), float(np.abs(e[:d] - c["error"]).max()))
            ref = np.array(c["J"])
            worst["onlypose_" + kind + "_J"] = max(worst.get("onlypose_" + kind + "_J", 0), float((np.abs(J[:6 * d] - ref) / np.maximum(1.0, np.abs(ref))).max()))
    check_binary_edges(lib, gold, worst)
    check_unary_edges(lib, gold, worst)
    print("max abs deviation oracle vs prebuilt g2o:", worst)
    return worst


if __name__ == "__main__":
    if "--check-only" in sys.argv:
        if os.path.exists(GOLD):
            check()
    else:
        generate()
        check()
