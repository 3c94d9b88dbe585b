"""Helpers (tests only): how far does the REFERENCE ALGORITHM's own result move when nothing but floating-point rounding changes?

Three re-runs of the oracle on the same graph:
  reversed      edges (chi2, H, b) and landmarks (Schur complement) summed in reverse order -- oracle_ba_set_variant(1);
                g2o itself has that freedom (edge order follows std::map<KeyFrame*, ...> pointer order, Optimizer.cc:2350)
  fma           the same source compiled with floating-point contraction allowed (oracle/libppo_oracle_fma.so)
  fma+reversed  both
  compensated   the long sums (pose blocks of Hpp, b, the Schur complement and its right-hand side) carried with a compensation
                term each -- oracle_ba_set_variant(2): what accumulation in a wider type (long double) would give
The spread between those runs is the accuracy to which the reference's result is DEFINED on that graph; a parity bar below
it cannot be met by any second implementation, the reference compiled with another compiler included.
"""
import ctypes as C
import os

import numpy as np

from ppo_ba import binding as B

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_fma = None


def fma_backend():
    global _fma
    if _fma is None:
        path = os.path.join(REPO, "oracle", "libppo_oracle_fma.so")
        _fma = B._Backend(C.CDLL(path), "oracle_ba_")
    return _fma


def run(backend, P, opt, variant=0):
    s = backend.create(P, opt)
    if variant:
        backend.lib.oracle_ba_set_variant.argtypes = [C.c_void_p, C.c_int]
        assert backend.lib.oracle_ba_set_variant(s.h, variant) == 0
    st = s.run()
    r = s.read()
    s.close()
    return st, r


def oracle_runs(oracle, P, opt, variants=("reversed", "fma", "fma+reversed")):
    """[(stats, result)] of the base oracle followed by the requested variants."""
    out = [run(oracle, P, opt)]
    for v in variants:
        be = fma_backend() if v.startswith("fma") else oracle
        out.append(run(be, P, opt, 2 if v == "compensated" else (1 if v.endswith("reversed") else 0)))
    return out


def quat_angle(qa, qb):
    qa, qb = np.asarray(qa), np.asarray(qb)
    return 2.0 * min(np.linalg.norm(qa - qb), np.linalg.norm(qa + qb))


def determined_points(P, r):
    """Points the second pass determines: at least two observations still active (level 0) in result `r` -- >= 4 residual rows for 3
    unknowns.  A landmark the culling left with ONE observation sits on a flat direction of the cost (its depth along the ray is held only
    by the damping): DESIGN.md section 6 keeps those out of the point RMSE, and so does the reference's own practice of erasing them later."""
    n = r.point_xyz.shape[0]
    cnt = np.zeros(n)
    if P["mono_point"].size:
        np.add.at(cnt, P["mono_point"].reshape(-1)[r.mono_level.reshape(-1) == 0], 1)
    if P["stereo_point"].size:
        np.add.at(cnt, P["stereo_point"].reshape(-1)[r.stereo_level.reshape(-1) == 0], 1)
    return cnt >= 2


def metrics(ra, rb, points=None):
    """The quantities of the parity bar between two results (points: boolean mask of the points that enter the point RMSE)."""
    n = ra.pose_qt.shape[0]
    d_pt = np.sum((ra.point_xyz - rb.point_xyz) ** 2, axis=1) if ra.point_xyz.size else np.zeros(0)
    if points is not None and d_pt.size:
        d_pt = d_pt[points]
    m = {"t": float(np.sqrt(np.mean(np.sum((ra.pose_qt[:, 4:] - rb.pose_qt[:, 4:]) ** 2, axis=1)))),
         "r": max(quat_angle(ra.pose_qt[i, :4], rb.pose_qt[i, :4]) for i in range(n)),
         "p": float(np.sqrt(np.mean(d_pt))) if d_pt.size else 0.0,
         "pl": float(np.abs(ra.plane_coef - rb.plane_coef).max()) if ra.plane_coef.size else 0.0,
         "cb": float(per_cuboid(ra, rb).max()) if ra.cuboid_qt.size else 0.0}
    return m


def per_cuboid(ra, rb):
    if not ra.cuboid_qt.size:
        return np.zeros(0)
    return np.maximum(np.abs(ra.cuboid_qt - rb.cuboid_qt).max(axis=1), np.abs(ra.cuboid_scale - rb.cuboid_scale).max(axis=1))


def per_point(ra, rb):
    return np.linalg.norm(ra.point_xyz - rb.point_xyz, axis=1) if ra.point_xyz.size else np.zeros(0)


def spread(runs, points=None):
    """Largest metric between the base run and any variant, and the per-cuboid / per-point maxima."""
    base = runs[0][1]
    m = {k: 0.0 for k in ("t", "r", "p", "pl", "cb")}
    cub = np.zeros(base.cuboid_qt.shape[0])
    pts = np.zeros(base.point_xyz.shape[0])
    for _, r in runs[1:]:
        mm = metrics(base, r, points)
        for k in m:
            m[k] = max(m[k], mm[k])
        if cub.size:
            cub = np.maximum(cub, per_cuboid(base, r))
        if pts.size:
            pts = np.maximum(pts, per_point(base, r))
    return m, cub, pts


def schedule(st):
    return [(a["pass_"], a["iteration"], a["trials"], a["accepted"], a["result"]) for a in st["iters"]]


VARIANTS = ("reversed", "fma", "fma+reversed", "compensated")  # the four rounding-only re-runs of the oracle the parity tests use
BARS = {"t": 1e-5, "r": 1e-6, "p": 1e-5, "pl": 1e-6, "cb": 1e-5}


def oracle_cost_at(oracle, so, r):
    """The oracle's CURRENT objective (after so.run(): the pass-2 cost -- its own active set, levels and kernels) evaluated at the
    state of result `r` (oracle_ba_chi2_at).  With r = so.read() it returns the oracle's own final robust chi2."""
    f = oracle.lib.oracle_ba_chi2_at
    f.argtypes = [C.c_void_p] + [B.c_dp] * 5 + [B.c_dp]
    f.restype = C.c_int
    chi = C.c_double()

    def p(a):
        a = np.ascontiguousarray(a, np.float64)
        return a, (a.ctypes.data_as(B.c_dp) if a.size else None)
    keep = [p(r.pose_qt), p(r.point_xyz), p(r.plane_coef), p(r.cuboid_qt), p(r.cuboid_scale)]
    assert f(so.h, *[k[1] for k in keep], C.cast(C.byref(chi), B.c_dp)) == 0
    return chi.value


def optimality_gap(oracle, so, ro, r_other):
    """|cost_ref(x_other) - cost_ref(x_oracle)| / cost_ref(x_oracle): how much worse (or better) a minimiser of the REFERENCE cost the
    other implementation's final state is."""
    c0 = oracle_cost_at(oracle, so, ro)
    c1 = oracle_cost_at(oracle, so, r_other)
    return abs(c1 - c0) / max(1.0, abs(c0)), c0, c1
