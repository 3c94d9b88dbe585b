#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (build container only): g2o objects of the reference's prebuilt libraries, driven through ctypes.

Every object below lives in a zeroed, 16-byte aligned numpy buffer and is initialised by the reference's OWN exported
constructor where one exists (VertexSE3Expmap, VertexSBAPointXYZ, VertexPlane, EdgePlane, EdgeVerticalPlane, EdgeParallelPlane,
BaseVertex<9, cuboid>, OptimizableGraph::Edge, BlockSolver<BlockSolverTraits<-1,-1>>, OptimizationAlgorithmLevenberg,
SparseOptimizer).  Classes whose constructors are inline in the headers (VertexCuboid, the cuboid edges, EdgeCuboidPlane,
LinearSolverDense) are completed exactly as the compiled call sites in lib/libORB_SLAM2.so do it (disassembly of
Optimizer::LocalBACameraPointCuboids2D at +0x6f9: operator new(0x78), _reset = 1, zeros, m_sign = 2, vtable + 16): base
constructor, `_dimension`, `_vertices.resize(2)`, the class's exported vtable.  No function body is supplied from here: every
virtual call lands in the reference's vtables.

Field offsets (all read off the disassembly, and cross-checked by the pattern they follow):
  OptimizableGraph::Vertex   _id +0x08, _hessianIndex +0x50, _fixed +0x54, _marginalized +0x55, _dimension +0x58
  BaseVertex<D, T>           Hessian map pointer +0x70, _b +0x80, _estimate after _b (16-byte aligned when T is):
                             VertexSBAPointXYZ +0x98; VertexSE3Expmap q +0xb0 / t +0xd0; VertexPlane +0xa0 (getEstimateData);
                             VertexCuboid q +0xd0 / t +0xf0 / scale +0x110, fixedscale +0x148, whether_fixrollpitch +0x160,
                             whether_fixrotation +0x161, whether_fixheight +0x162 (VertexCuboid::oplusImpl)
  OptimizableGraph::Edge     _vertices +0x08, _dimension +0x24, _level +0x28, _robustKernel +0x30
  BaseEdge<D, E>             _measurement +0xa0, _information next (16-byte aligned when 8 D D is a multiple of 16), _error next
  BaseBinaryEdge             R = end of _error: _hessianRowMajor +R, _hessian +R+8, _hessianTransposed +R+0x18,
                             _jacobianOplusXi +R+0x28, _jacobianOplusXj +R+0x38 (mapHessianMemory / linearizeOplus(JacobianWorkspace&))
                             derived members from R+0x48: Kalib of the two cuboid projection edges
  BaseUnaryEdge              _jacobianOplusXi +R; EdgePointCuboidOnlyObject: object_points +0x130, max_outside_margin_ratio +0x148,
                             prior_object_half_size +0x150 (EdgePointCuboidOnlyObject::computeError)
"""
import ctypes as C
import subprocess

import numpy as np

import orbslam_loader as loader

vp = C.c_void_p
_keep = []


def aligned(nbytes):
    a = np.zeros(nbytes // 8 + 4)
    o = (-a.ctypes.data // 8) % 2
    b = a[o:o + nbytes // 8]
    _keep.append(a)
    return b


def u64(a):
    return a.view(np.uint64)


def _symtab(path):
    raw = subprocess.check_output(["nm", "-D", "--defined-only", path], text=True).splitlines()
    names = [ln.split()[-1].split("@")[0] for ln in raw if ln.strip()]
    dem = subprocess.check_output(["c++filt"], input="\n".join(names), text=True).splitlines()
    tab = {}
    for m, d in zip(names, dem):
        tab.setdefault(d, m)  # C1 before C2: nm lists them in that order
    return tab


class Ref:
    """Symbol access by DEMANGLED name into libORB_SLAM2.so (first) and libg2o.so."""

    def __init__(self):
        self.g2o, self.orb, self.sink = loader.load()
        self.tabs = [(self.orb, _symtab(loader.ORB)), (self.g2o, _symtab(loader.G2O))]

    def _find(self, demangled):
        for lib, tab in self.tabs:
            if demangled in tab:
                return lib, tab[demangled]
        raise KeyError(demangled)

    def fn(self, demangled, *argtypes, restype=None):
        lib, m = self._find(demangled)
        f = getattr(lib, m)
        f.argtypes = list(argtypes)
        f.restype = restype
        return f

    def vtable(self, cls):
        lib, m = self._find("vtable for " + cls)
        return C.addressof(C.c_void_p.in_dll(lib, m)) + 16

    def virtual(self, obj, slot, *argtypes, restype=None):
        """The function in `slot` (byte offset) of the object's own vtable."""
        vptr = int(u64(obj)[0])
        addr = C.cast(vptr + slot, C.POINTER(C.c_uint64))[0]
        return C.CFUNCTYPE(restype, *argtypes)(addr)


V3 = "Eigen::Matrix<double, 3, 1, 0, 3, 1>"
V4 = "Eigen::Matrix<double, 4, 1, 0, 4, 1>"
V16 = "Eigen::Matrix<double, 16, 1, 0, 16, 1>"
V_ID, V_HIDX, V_FIXED, V_MARG, V_DIM, V_HESS, V_B = 0x08, 0x50, 0x54, 0x55, 0x58, 0x70, 0x80
E_DIM, E_LEVEL, E_KERNEL, E_MEAS = 0x24, 0x28, 0x30, 0xa0
SLOT_LINEARIZE = 0xe0  # virtual linearizeOplus(): BaseBinaryEdge / BaseUnaryEdge::linearizeOplus(JacobianWorkspace&) jump through it


def edge_layout(D, meas_bytes, meas_aligned, unary=False):
    """Offsets of a BaseEdge<D, E> / BaseBinaryEdge / BaseUnaryEdge instantiation (module docstring)."""
    off = E_MEAS + meas_bytes
    if (8 * D * D) % 16 == 0:
        off = (off + 15) & ~15
    info = off
    off += 8 * D * D
    if (8 * D) % 16 == 0:
        off = (off + 15) & ~15
    err = off
    R = err + 8 * D
    L = dict(D=D, meas=E_MEAS, info=info, err=err, R=R)
    if unary:
        L.update(jxi=R, derived=R + 0x10)
    else:
        L.update(rowmajor=R, hess=R + 8, hessT=R + 0x18, jxi=R + 0x28, jxj=R + 0x38, derived=R + 0x48)
    return L


EDGE_CLASSES = {
    # name: (class, base template, layout, (dim of vertex 0, dim of vertex 1))
    "plane": ("g2o::EdgePlane", "g2o::BaseBinaryEdge<3, g2o::Plane3D, g2o::VertexPlane, g2o::VertexSE3Expmap>", edge_layout(3, 32, True), (3, 6)),
    "plane_ver": ("g2o::EdgeVerticalPlane", "g2o::BaseBinaryEdge<2, g2o::Plane3D, g2o::VertexPlane, g2o::VertexSE3Expmap>", edge_layout(2, 32, True), (3, 6)),
    "plane_par": ("g2o::EdgeParallelPlane", "g2o::BaseBinaryEdge<2, g2o::Plane3D, g2o::VertexPlane, g2o::VertexSE3Expmap>", edge_layout(2, 32, True), (3, 6)),
    "bbox": ("g2o::EdgeSE3CuboidProj", "g2o::BaseBinaryEdge<4, %s, g2o::VertexSE3Expmap, g2o::VertexCuboid>" % V4, edge_layout(4, 32, True), (6, 9)),
    "corner": ("g2o::EdgeSE3CuboidCornerProj", "g2o::BaseBinaryEdge<16, %s, g2o::VertexSE3Expmap, g2o::VertexCuboid>" % V16, edge_layout(16, 128, True), (6, 9)),
    "cuboid3d": ("g2o::EdgeSE3Cuboid", "g2o::BaseBinaryEdge<9, g2o::cuboid, g2o::VertexSE3Expmap, g2o::VertexCuboid>", edge_layout(9, 96, True), (6, 9)),
    "cubpl": ("g2o::EdgeCuboidPlane", "g2o::BaseBinaryEdge<3, %s, g2o::VertexCuboid, g2o::VertexPlane>" % V3, edge_layout(3, 24, False), (9, 3)),
    "ptcub": ("g2o::EdgePointCuboidOnlyObject", "g2o::BaseUnaryEdge<3, %s, g2o::VertexCuboid>" % V3, edge_layout(3, 24, False, unary=True), (9, 0)),
    "mono": ("g2o::EdgeSE3ProjectXYZ", "g2o::BaseBinaryEdge<2, Eigen::Matrix<double, 2, 1, 0, 2, 1>, g2o::VertexSBAPointXYZ, g2o::VertexSE3Expmap>", edge_layout(2, 16, True), (3, 6)),
    "stereo": ("g2o::EdgeStereoSE3ProjectXYZ", "g2o::BaseBinaryEdge<3, %s, g2o::VertexSBAPointXYZ, g2o::VertexSE3Expmap>" % V3, edge_layout(3, 24, False), (3, 6)),
}
assert EDGE_CLASSES["plane"][2]["hess"] == 0x128 and EDGE_CLASSES["plane_ver"][2]["hess"] == 0xf8 and EDGE_CLASSES["bbox"][2]["hess"] == 0x168
assert EDGE_CLASSES["corner"][2]["hess"] == 0x9a8 and EDGE_CLASSES["cuboid3d"][2]["hess"] == 0x3d8 and EDGE_CLASSES["cubpl"][2]["hess"] == 0x120
assert EDGE_CLASSES["bbox"][2]["derived"] == 0x1a8 and EDGE_CLASSES["ptcub"][2]["err"] == 0x100 and EDGE_CLASSES["ptcub"][2]["jxi"] == 0x118
assert EDGE_CLASSES["mono"][2]["jxi"] == 0x108 and EDGE_CLASSES["stereo"][2]["jxj"] == 0x150 and EDGE_CLASSES["stereo"][2]["info"] == 0xb8


class World:
    """Factory for vertices, edges and robust kernels inside the reference binaries."""

    def __init__(self):
        self.R = Ref()

    # ---- vertices --------------------------------------------------------------------------------------------------------
    def pose(self, qt, fixed=0, vid=0):
        v = aligned(0x200)
        self.R.fn("g2o::VertexSE3Expmap::VertexSE3Expmap()", vp)(v.ctypes.data)
        v[0xb0 // 8:0xb0 // 8 + 4] = qt[:4]
        v[0xd0 // 8:0xd0 // 8 + 3] = qt[4:7]
        v.view(np.int32)[V_ID // 4] = vid
        v.view(np.uint8)[V_FIXED] = fixed
        return v

    @staticmethod
    def pose_state(v):
        return v[0xb0 // 8:0xb0 // 8 + 4].tolist() + v[0xd0 // 8:0xd0 // 8 + 3].tolist()

    def point(self, xyz, fixed=0, vid=0, marginalized=0):
        v = aligned(0x200)
        self.R.fn("g2o::VertexSBAPointXYZ::VertexSBAPointXYZ()", vp)(v.ctypes.data)
        v[0x98 // 8:0x98 // 8 + 3] = xyz
        v.view(np.int32)[V_ID // 4] = vid
        v.view(np.uint8)[V_FIXED] = fixed
        v.view(np.uint8)[V_MARG] = marginalized
        return v

    @staticmethod
    def point_state(v):
        return v[0x98 // 8:0x98 // 8 + 3].tolist()

    def plane(self, coef, vid=0, marginalized=0):
        """VertexPlane; the estimate goes through the reference's own Plane3D::fromVector (normalisation, d >= 0)."""
        v = aligned(0x200)
        self.R.fn("g2o::VertexPlane::VertexPlane()", vp)(v.ctypes.data)
        c = np.ascontiguousarray(coef, dtype=np.float64)
        self.R.fn("g2o::VertexPlane::setEstimateDataImpl(double const*)", vp, vp, restype=C.c_bool)(v.ctypes.data, c.ctypes.data)
        v.view(np.int32)[V_ID // 4] = vid
        v.view(np.uint8)[V_MARG] = marginalized
        return v

    @staticmethod
    def plane_state(v):
        return v[0xa0 // 8:0xa0 // 8 + 4].tolist()

    def plane3d(self, coef):
        """A normalised Plane3D value (4 doubles) as the reference's Plane3D(const Vector4D&) makes it."""
        return np.array(self.plane_state(self.plane(coef)))

    def cuboid(self, qt, scale, flags=0, fixedscale=(0.0, 0.0, 0.0), vid=0):
        """VertexCuboid: BaseVertex<9, cuboid>() + the inline VertexCuboid() body + its vtable (g2o_cuboid.h:262-268)."""
        v = aligned(0x200)
        self.R.fn("g2o::BaseVertex<9, g2o::cuboid>::BaseVertex()", vp)(v.ctypes.data)
        u64(v)[0] = self.R.vtable("g2o::VertexCuboid")
        assert v.view(np.int32)[V_DIM // 4] == 9
        v[0xd0 // 8:0xd0 // 8 + 4] = qt[:4]
        v[0xf0 // 8:0xf0 // 8 + 3] = qt[4:7]
        v[0x110 // 8:0x110 // 8 + 3] = scale
        v[0x148 // 8:0x148 // 8 + 3] = fixedscale
        b = v.view(np.uint8)
        b[0x160], b[0x161], b[0x162] = flags & 1, (flags >> 2) & 1, (flags >> 1) & 1  # ppo_ba.h: bit0 rollpitch, bit1 height, bit2 rotation
        v.view(np.int32)[V_ID // 4] = vid
        return v

    @staticmethod
    def cuboid_state(v):
        return v[0xd0 // 8:0xd0 // 8 + 4].tolist() + v[0xf0 // 8:0xf0 // 8 + 3].tolist(), v[0x110 // 8:0x110 // 8 + 3].tolist()

    def oplus(self, v, upd):
        """The virtual oplusImpl of the vertex (slot found through the exported symbol of its class is not needed: call by vtable
        is avoided here -- each vertex class exports oplusImpl)."""
        u = np.ascontiguousarray(upd, dtype=np.float64)
        vt = int(u64(v)[0])
        for cls in ("g2o::VertexPlane", "g2o::VertexCuboid", "g2o::VertexSE3Expmap", "g2o::VertexSBAPointXYZ"):
            if vt == self.R.vtable(cls):
                self.R.fn(cls + "::oplusImpl(double const*)", vp, vp)(v.ctypes.data, u.ctypes.data)
                return
        raise RuntimeError("unknown vertex class")

    # ---- robust kernel ---------------------------------------------------------------------------------------------------
    def huber(self, delta):
        k = aligned(64)
        u64(k)[0] = self.R.vtable("g2o::RobustKernelHuber")
        self.R.fn("g2o::RobustKernelHuber::setDelta(double)", vp, C.c_double)(k.ctypes.data, delta)
        return k

    # ---- edges -----------------------------------------------------------------------------------------------------------
    def edge(self, kind, v0, v1, meas, info_diag, huber_delta=0.0, K=None, points=None, margin_ratio=0.0, prior_half=None):
        cls, base, L, _ = EDGE_CLASSES[kind]
        D = L["D"]
        e = aligned(0x1000)
        ctor = cls + "::" + cls.split("::")[-1] + "()"
        try:
            self.R.fn(ctor, vp)(e.ctypes.data)
        except KeyError:  # inline constructor: OptimizableGraph::Edge() + what the BaseEdge / BaseBinaryEdge constructors do
            self.R.fn("g2o::OptimizableGraph::Edge::Edge()", vp)(e.ctypes.data)
            e.view(np.int32)[E_DIM // 4] = D
            u64(e)[0] = self.R.vtable(cls)
            self.R.fn(base + "::resize(unsigned long)", vp, C.c_size_t)(e.ctypes.data, 1 if v1 is None else 2)
        assert e.view(np.int32)[E_DIM // 4] == D
        slots = C.cast(int(u64(e)[1]), C.POINTER(C.c_uint64))
        assert (int(u64(e)[2]) - int(u64(e)[1])) // 8 == (1 if v1 is None else 2)
        slots[0] = v0.ctypes.data
        if v1 is not None:
            slots[1] = v1.ctypes.data
        m = np.ascontiguousarray(meas, dtype=np.float64)
        e[L["meas"] // 8:L["meas"] // 8 + m.size] = m
        e[L["info"] // 8:L["info"] // 8 + D * D] = np.diag(np.asarray(info_diag, dtype=np.float64) * np.ones(D)).reshape(-1)
        if huber_delta > 0:
            u64(e)[E_KERNEL // 8] = self.huber(huber_delta).ctypes.data
        if K is not None:
            e[L["derived"] // 8:L["derived"] // 8 + 9] = np.asarray(K, dtype=np.float64).reshape(3, 3).T.reshape(-1)  # Matrix3d is column-major
        if kind == "ptcub":
            pts = aligned(8 * 3 * max(1, len(points)) + 64)
            pts[:3 * len(points)] = np.asarray(points, dtype=np.float64).reshape(-1)
            u64(e)[0x130 // 8] = pts.ctypes.data
            u64(e)[0x138 // 8] = pts.ctypes.data + 24 * len(points)
            u64(e)[0x140 // 8] = pts.ctypes.data + 24 * len(points)
            e[0x148 // 8] = margin_ratio
            e[0x150 // 8:0x150 // 8 + 3] = prior_half if prior_half is not None else (0.0, 0.0, 0.0)
        return e

    def compute_error(self, kind, e):
        cls, _, L, _ = EDGE_CLASSES[kind]
        self.R.fn(cls + "::computeError()", vp)(e.ctypes.data)
        return e[L["err"] // 8:L["err"] // 8 + L["D"]].copy()

    def chi2(self, kind, e):
        """BaseEdge<D, E>::chi2() of the reference (the stored _error against the information matrix)."""
        cls, base, L, _ = EDGE_CLASSES[kind]
        btype = base[base.index("<") + 1:]
        D = L["D"]
        etype = btype[btype.index(",") + 2:]
        depth, k = 0, 0
        for k, ch in enumerate(etype):  # the measurement type ends at the first comma outside <>
            depth += ch == "<"
            depth -= ch == ">"
            if ch == "," and depth == 0:
                break
        name = "g2o::BaseEdge<%d, %s%s>::chi2() const" % (D, etype[:k], " " if etype[:k].endswith(">") else "")
        return float(self.R.fn(name, vp, restype=C.c_double)(e.ctypes.data))

    def linearize(self, kind, e):
        """The virtual linearizeOplus() of the edge (numeric central differences for every custom edge): returns (J0, J1)
        row-major, D x d0 and D x d1."""
        _, _, L, (d0, d1) = EDGE_CLASSES[kind]
        D = L["D"]
        J0, J1 = aligned(8 * D * d0 + 64), aligned(8 * D * max(d1, 1) + 64)
        u64(e)[L["jxi"] // 8] = J0.ctypes.data
        if d1:
            u64(e)[L["jxj"] // 8] = J1.ctypes.data
        self.R.virtual(e, SLOT_LINEARIZE, vp)(e.ctypes.data)
        out0 = J0[:D * d0].reshape(d0, D).T.copy()
        out1 = J1[:D * d1].reshape(d1, D).T.copy() if d1 else None
        return out0, out1

    def quadratic_form(self, kind, e, v0, v1):
        """mapHessianMemory + constructQuadraticForm of the reference; returns H00, H11, H01 (row-major) and b0, b1."""
        _, base, L, (d0, d1) = EDGE_CLASSES[kind]
        H0 = aligned(8 * d0 * d0 + 64)
        u64(v0)[V_HESS // 8] = H0.ctypes.data
        v0[V_B // 8:V_B // 8 + d0] = 0.0
        H1 = H01 = None
        if d1:
            H1, H01 = aligned(8 * d1 * d1 + 64), aligned(8 * d0 * d1 + 64)
            u64(v1)[V_HESS // 8] = H1.ctypes.data
            v1[V_B // 8:V_B // 8 + d1] = 0.0
            self.R.fn(base + "::mapHessianMemory(double*, int, int, bool)", vp, vp, C.c_int, C.c_int, C.c_bool)(e.ctypes.data, H01.ctypes.data, 0, 1, False)
        self.R.fn(base + "::constructQuadraticForm()", vp)(e.ctypes.data)
        out = {"H0": H0[:d0 * d0].reshape(d0, d0).T.copy(), "b0": v0[V_B // 8:V_B // 8 + d0].copy()}
        if d1:
            out.update(H1=H1[:d1 * d1].reshape(d1, d1).T.copy(), b1=v1[V_B // 8:V_B // 8 + d1].copy(), H01=H01[:d0 * d1].reshape(d1, d0).T.copy())
        return out
