"""Seeded synthetic local-BA graphs for the BASELINE.json configurations.

Scene: a 8 x 6 x 3 m room (world z up, world origin outside the room so that plane offsets keep one sign),
cameras on a closed loop looking roughly inward, points on the room surfaces, random planes, upright cuboids.
Everything the reference would read from its data model is produced here in the flattened form of
include/ppo_ba.h, with the reference's constants:
  Huber deltas are float32 as in Optimizer.cc:2203,2205,2326-2327,2474,2538,2669; invSigma2 = 1.2^(-2*octave) in
  float32 (mvInvLevelSigma2); information and thresholds from src/Parameters.cc:52-74 and Optimizer.cc:2194-2205.
Initial estimates are truth + noise, rounded through float32 like Converter.cc:37-47,110-116,172-181.
"""
import numpy as np

from .binding import Problem

FX = FY = 480.0
CX, CY = 320.0, 240.0
W_IMG, H_IMG = 640, 480
ROOM_LO = np.array([1.0, 1.0, 0.5])
ROOM_HI = ROOM_LO + np.array([8.0, 6.0, 3.0])

CONFIGS = {
    # name: n_kf, n_points, n_planes, n_cuboids
    "tiny": dict(n_kf=6, n_points=120, n_planes=4, n_cuboids=2),
    "small": dict(n_kf=10, n_points=200, n_planes=4, n_cuboids=2),
    "cfg2": dict(n_kf=50, n_points=10000, n_planes=0, n_cuboids=0),
    "cfg3": dict(n_kf=100, n_points=30000, n_planes=100, n_cuboids=0),
    "cfg4": dict(n_kf=200, n_points=50000, n_planes=200, n_cuboids=50),
    # BASELINE config 5 (the config-4 graph sharded over N GPUs) is "cfg4" with world_size = N.  "cfg5w" is the weak-scaling
    # companion SURVEY 8(e) asks for: the same 200 keyframes / planes / cuboids with 50 000 points PER GPU (400 000 on 8), so that
    # the sharded part (linearisation, Schur complement, back-substitution) stays constant per rank while the all-reduce does not grow
    "cfg5w": dict(n_kf=200, n_points=50000, n_planes=200, n_cuboids=50, points_per_rank=True),
}


def f32(x):
    return np.asarray(x, dtype=np.float32).astype(np.float64)


def quat_from_matrix(R):
    """Eigen's rotation-matrix -> quaternion (x,y,z,w), then g2o's normalizeRotation (w >= 0, unit norm)."""
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0)
        w = 0.5 * s
        s = 0.5 / s
        q = np.array([(R[2, 1] - R[1, 2]) * s, (R[0, 2] - R[2, 0]) * s, (R[1, 0] - R[0, 1]) * s, w])
    else:
        i = 0
        if R[1, 1] > R[0, 0]:
            i = 1
        if R[2, 2] > R[i, i]:
            i = 2
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
        q = np.zeros(4)
        q[i] = 0.5 * s
        s = 0.5 / s
        q[3] = (R[k, j] - R[j, k]) * s
        q[j] = (R[j, i] + R[i, j]) * s
        q[k] = (R[k, i] + R[i, k]) * s
    if q[3] < 0:
        q = -q
    return q / np.linalg.norm(q)


def rodrigues(w):
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-12:
        return np.eye(3) + K
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th**2 * K @ K


def look_at(pos, fwd):
    fwd = fwd / np.linalg.norm(fwd)
    up = np.array([0.0, 0.0, 1.0])
    right = np.cross(fwd, up)
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    Rwc = np.stack([right, down, fwd], axis=1)  # columns = camera axes in world
    Rcw = Rwc.T
    return Rcw, -Rcw @ pos


def make_cameras(rng, n_kf):
    c = 0.5 * (ROOM_LO + ROOM_HI)
    ang = 2 * np.pi * np.arange(n_kf) / n_kf
    pos = np.stack([c[0] + 2.6 * np.cos(ang), c[1] + 1.7 * np.sin(ang), ROOM_LO[2] + rng.uniform(1.2, 1.6, n_kf)], axis=1)
    Rs, ts = [], []
    for i in range(n_kf):
        # look across the room centre with yaw / pitch jitter
        tgt = c + np.array([0.0, 0.0, -0.2])
        f = tgt - pos[i]
        yaw = rng.uniform(-np.pi / 6, np.pi / 6)
        cz, sz = np.cos(yaw), np.sin(yaw)
        f = np.array([cz * f[0] - sz * f[1], sz * f[0] + cz * f[1], f[2] + rng.uniform(-0.3, 0.3)])
        R, t = look_at(pos[i], f)
        Rs.append(R)
        ts.append(t)
    return np.array(Rs), np.array(ts), pos


def sample_surface_points(rng, n):
    """Uniform by area on the six room faces."""
    ext = ROOM_HI - ROOM_LO
    areas = np.array([ext[1] * ext[2], ext[1] * ext[2], ext[0] * ext[2], ext[0] * ext[2], ext[0] * ext[1], ext[0] * ext[1]])
    face = rng.choice(6, size=n, p=areas / areas.sum())
    u = rng.uniform(size=(n, 3)) * ext + ROOM_LO
    for f in range(6):
        ax = f // 2
        u[face == f, ax] = ROOM_LO[ax] if f % 2 == 0 else ROOM_HI[ax]
    return u


def project(Rcw, tcw, X):
    Xc = X @ Rcw.T + tcw
    z = Xc[:, 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        u = FX * Xc[:, 0] / z + CX
        v = FY * Xc[:, 1] / z + CY
    return u, v, z


def plane_rotation(n):
    az = np.arctan2(n[1], n[0])
    el = np.arctan2(n[2], np.hypot(n[0], n[1]))
    ca, sa, ce, se = np.cos(az), np.sin(az), np.cos(-el), np.sin(-el)
    Rz = np.array([[ca, -sa, 0], [sa, ca, 0], [0, 0, 1]])
    Ry = np.array([[ce, 0, se], [0, 1, 0], [-se, 0, ce]])
    return Rz @ Ry


def plane_normalize(c):
    c = c / np.linalg.norm(c[:3])
    return -c if c[3] < 0 else c


def plane_oplus(c, v):
    s, co = np.sin(v[1]), np.cos(v[1])
    n = np.array([co * np.cos(v[0]), co * np.sin(v[0]), s])
    R = plane_rotation(c[:3])
    d = -c[3] + v[2]
    return plane_normalize(np.concatenate([R @ n, [-d]]))


def plane_to_camera(Rcw, tcw, c):
    n2 = Rcw @ c[:3]
    d2 = c[3] - tcw @ n2
    out = np.concatenate([n2, [d2]])
    return plane_normalize(out if d2 >= 0 else -out)


def nearest_in_sequence(rng, visible, k, n_kf):
    """For each row of the boolean matrix pick k visible columns nearest (circularly) to a random visible anchor."""
    n = visible.shape[0]
    nvis = visible.sum(axis=1)
    r = rng.uniform(size=n)
    order = np.argsort(~visible, axis=1, kind="stable")  # visible columns first, ascending
    anchor = order[np.arange(n), np.minimum((r * np.maximum(nvis, 1)).astype(int), np.maximum(nvis - 1, 0))]
    idx = np.arange(n_kf)[None, :]
    d = np.abs(idx - anchor[:, None])
    d = np.minimum(d, n_kf - d).astype(np.float64)
    d = d + 1e-3 * idx / n_kf  # deterministic tie-break
    d[~visible] = np.inf
    sel = np.argsort(d, axis=1, kind="stable")
    kk = np.minimum(k, nvis)
    return sel, kk


class _ScaledRng:
    """numpy Generator whose ``normal`` draws are scaled: noise_scale = 0 yields an exact (noise-free) graph."""

    def __init__(self, rng, scale):
        self._rng, self._scale = rng, scale

    def normal(self, *a, **k):
        return self._rng.normal(*a, **k) * self._scale

    def geometry_normal(self, *a, **k):  # scene geometry (directions), never scaled
        return self._rng.normal(*a, **k)

    def __getattr__(self, name):
        return getattr(self._rng, name)


def make_problem(config="cfg4", seed=1, outlier_frac=0.05, stereo_frac=0.0, with_corners=False, with_vp_planes=0,
                 fixed_observer_frac=0.0, flags=None, noise_scale=1.0, world_size=1, with_cuboid3d=False):
    """Build one seeded graph.  ``config`` is a key of CONFIGS or a dict with n_kf/n_points/n_planes/n_cuboids."""
    cfg = dict(CONFIGS[config]) if isinstance(config, str) else dict(config)
    if cfg.pop("points_per_rank", False):
        cfg["n_points"] *= max(1, int(world_size))
    rng = _ScaledRng(np.random.default_rng(seed), noise_scale)
    n_kf, n_pt, n_pl, n_cb = cfg["n_kf"], cfg["n_points"], cfg["n_planes"], cfg["n_cuboids"]
    on = dict(plane_3d=n_pl > 0, cuboid_2d=n_cb > 0, pt_obj_3d=n_cb > 0, cuboid_plane=n_cb > 0 and n_pl > 0, corners_2d=with_corners)
    if flags:
        on.update(flags)

    Rcw, tcw, cam_pos = make_cameras(rng, n_kf)
    n_fixed_extra = int(round(fixed_observer_frac * n_kf))

    # ---- points and reprojection edges ---------------------------------------------------------
    need = n_pt
    Xs, obs_lists = [], []
    X_true = np.zeros((0, 3))
    sel_pose, sel_pt = np.zeros(0, np.int64), np.zeros(0, np.int64)
    while X_true.shape[0] < need:
        m = int((need - X_true.shape[0]) * 1.4) + 16
        cand = sample_surface_points(rng, m)
        vis = np.zeros((m, n_kf), dtype=bool)
        for j in range(n_kf):
            u, v, z = project(Rcw[j], tcw[j], cand)
            vis[:, j] = (z > 0.3) & (z < 8.0) & (u > 1) & (u < W_IMG - 1) & (v > 1) & (v < H_IMG - 1)
        k = np.clip(rng.poisson(5.0, size=m), 2, 12)
        sel, kk = nearest_in_sequence(rng, vis, k, n_kf)
        good = np.nonzero(kk >= 2)[0]
        good = good[: need - X_true.shape[0]]
        base = X_true.shape[0]
        X_true = np.vstack([X_true, cand[good]])
        reps = kk[good]
        pt_ids = np.repeat(np.arange(len(good)) + base, reps)
        # columns 0..kk-1 of sel for every kept row
        col = np.concatenate([np.arange(r) for r in reps]) if len(reps) else np.zeros(0, int)
        rows = np.repeat(good, reps)
        pose_ids = sel[rows, col]
        sel_pt = np.concatenate([sel_pt, pt_ids])
        sel_pose = np.concatenate([sel_pose, pose_ids])
    n_obs = sel_pt.size
    Xc = np.einsum("eij,ej->ei", Rcw[sel_pose], X_true[sel_pt]) + tcw[sel_pose]
    u = FX * Xc[:, 0] / Xc[:, 2] + CX
    v = FY * Xc[:, 1] / Xc[:, 2] + CY
    octave = rng.integers(0, 8, size=n_obs)
    sigma = 1.2 ** octave
    u_n = u + rng.normal(size=n_obs) * sigma
    v_n = v + rng.normal(size=n_obs) * sigma
    is_out = rng.uniform(size=n_obs) < outlier_frac
    u_n[is_out] = rng.uniform(0, W_IMG, size=is_out.sum())
    v_n[is_out] = rng.uniform(0, H_IMG, size=is_out.sum())
    inv_sigma2 = f32(1.0 / (np.float32(1.2) ** octave.astype(np.float32)) ** 2)  # mvInvLevelSigma2 is float
    is_stereo = rng.uniform(size=n_obs) < stereo_frac
    bf = 40.0
    ur = u_n - bf / Xc[:, 2] + rng.normal(size=n_obs) * sigma

    P = Problem()
    # ---- initial estimates (truth + noise, through float32) -------------------------------------
    pose_qt = np.zeros((n_kf, 7))
    for j in range(n_kf):
        if j == 0:
            R0, t0 = Rcw[j], tcw[j]
        else:
            dR = rodrigues(rng.normal(size=3) * 0.01)
            R0 = dR @ Rcw[j]
            t0 = dR @ tcw[j] + rng.normal(size=3) * 0.02
        Rf, tf = f32(R0), f32(t0)
        pose_qt[j, :4] = quat_from_matrix(Rf)
        pose_qt[j, 4:] = tf
    pose_fixed = np.zeros(n_kf, np.uint8)
    pose_fixed[0] = 1  # Optimizer.cc:2126
    if n_fixed_extra:
        pose_fixed[rng.choice(np.arange(1, n_kf), size=n_fixed_extra, replace=False)] = 1
    P.set("pose_qt", pose_qt)
    P.set("pose_fixed", pose_fixed)
    P.set("pose_cam", np.tile(f32([FX, FY, CX, CY, bf]), (n_kf, 1)))
    P.set("point_xyz", f32(X_true + rng.normal(size=X_true.shape) * 0.03))

    mono = ~is_stereo
    th_mono = float(np.float32(np.sqrt(5.991)))
    th_stereo = float(np.float32(np.sqrt(7.815)))
    P.set("mono_point", sel_pt[mono]); P.set("mono_pose", sel_pose[mono])
    P.set("mono_obs", f32(np.stack([u_n[mono], v_n[mono]], axis=1)))
    P.set("mono_info", inv_sigma2[mono]); P.set("mono_huber", np.full(mono.sum(), th_mono))
    P.set("stereo_point", sel_pt[is_stereo]); P.set("stereo_pose", sel_pose[is_stereo])
    P.set("stereo_obs", f32(np.stack([u_n[is_stereo], v_n[is_stereo], ur[is_stereo]], axis=1)))
    P.set("stereo_info", inv_sigma2[is_stereo]); P.set("stereo_huber", np.full(is_stereo.sum(), th_stereo))

    # ---- planes ---------------------------------------------------------------------------------
    planes_true = np.zeros((n_pl, 4))
    pl_plane, pl_pose, pl_kind, pl_meas = [], [], [], []
    for i in range(n_pl):
        while True:
            n = rng.geometry_normal(size=3)
            n /= np.linalg.norm(n)
            p0 = rng.uniform(ROOM_LO, ROOM_HI)
            c = plane_normalize(np.concatenate([n, [-n @ p0]]))
            dist = np.abs(cam_pos @ c[:3] + c[3])
            ok = np.nonzero(dist > 0.5)[0]
            if c[3] > 0.5 and ok.size >= min(20, n_kf - 1):
                break
        planes_true[i] = c
        anchor = rng.choice(ok)
        d = np.abs(ok - anchor)
        d = np.minimum(d, n_kf - d)
        obs_kf = ok[np.argsort(d, kind="stable")[: min(20, ok.size)]]
        for j in np.sort(obs_kf):
            m = plane_to_camera(Rcw[j], tcw[j], c)
            m = plane_oplus(m, np.array([rng.normal() * np.deg2rad(0.5), rng.normal() * np.deg2rad(0.5), rng.normal() * 0.01]))
            pl_plane.append(i); pl_pose.append(j); pl_kind.append(0); pl_meas.append(f32(m))
        for kind in (1, 2):
            for _ in range(with_vp_planes):
                j = int(rng.choice(ok))
                m = plane_to_camera(Rcw[j], tcw[j], c)
                if kind == 1:  # a plane perpendicular to this one, as seen from the camera
                    a = np.cross(m[:3], rng.geometry_normal(size=3)); a /= np.linalg.norm(a)
                    m = plane_normalize(np.concatenate([a, [rng.uniform(0.5, 3.0)]]))
                else:
                    m = plane_normalize(np.concatenate([m[:3], [rng.uniform(0.5, 3.0)]]))
                m = plane_oplus(m, np.array([rng.normal() * np.deg2rad(0.5), rng.normal() * np.deg2rad(0.5), 0.0]))
                pl_plane.append(i); pl_pose.append(j); pl_kind.append(kind); pl_meas.append(f32(m))
    plane_init = np.array([f32(plane_oplus(planes_true[i], np.array([rng.normal() * np.deg2rad(0.5), rng.normal() * np.deg2rad(0.5), rng.normal() * 0.01])))
                           for i in range(n_pl)]).reshape(n_pl, 4)
    P.set("plane_coef", plane_init)
    n_ple = len(pl_plane) if on["plane_3d"] else 0
    angle_info = 3282.8 / (1.0 * 1.0)
    dist_info = 100.0 * 100.0
    vp_info = 3282.8 / (0.5 * 0.5)
    kinds = np.array(pl_kind[:n_ple], dtype=np.uint8)
    info = np.zeros((n_ple, 3))
    if n_ple:
        info[kinds == 0] = [angle_info, angle_info, dist_info]
        info[kinds != 0] = [vp_info, vp_info, 0.0]
    hub = np.where(kinds == 0, float(np.float32(np.sqrt(500.0))), float(np.float32(np.sqrt(200.0)))) if n_ple else np.zeros(0)
    P.set("pl_plane", np.array(pl_plane[:n_ple], dtype=np.int32)); P.set("pl_pose", np.array(pl_pose[:n_ple], dtype=np.int32))
    P.set("pl_kind", kinds); P.set("pl_meas", np.array(pl_meas[:n_ple]).reshape(n_ple, 4))
    P.set("pl_info", info); P.set("pl_huber", hub)

    # ---- cuboids ----------------------------------------------------------------------------------
    cub_qt = np.zeros((n_cb, 7)); cub_scale = np.zeros((n_cb, 3))
    cub_qt_true = np.zeros((n_cb, 7)); cub_scale_true = np.zeros((n_cb, 3))
    bbox_pose, bbox_cub, bbox_meas, corner_meas = [], [], [], []
    signs = np.array([[1, 1, -1, -1, 1, 1, -1, -1], [1, -1, -1, 1, 1, -1, -1, 1], [-1, -1, -1, -1, 1, 1, 1, 1]], dtype=float)
    K = f32([[FX, 0, CX], [0, FY, CY], [0, 0, 1]])
    for i in range(n_cb):
        for _ in range(200):
            yaw = rng.uniform(-np.pi, np.pi)
            ctr = rng.uniform(ROOM_LO + [2.0, 1.5, 0.3], ROOM_HI - [2.0, 1.5, 1.5])
            hs = rng.uniform(0.1, 0.5, size=3)
            Rwo = np.array([[np.cos(yaw), -np.sin(yaw), 0], [np.sin(yaw), np.cos(yaw), 0], [0, 0, 1]])
            corners = (Rwo @ (signs * hs[:, None])).T + ctr
            ok = []
            for j in range(n_kf):
                uu, vv, zz = project(Rcw[j], tcw[j], corners)
                if zz.min() > 0.5 and uu.min() > 6 and uu.max() < W_IMG - 6 and vv.min() > 6 and vv.max() < H_IMG - 6:
                    ok.append(j)
            if len(ok) >= min(10, n_kf // 2):
                break
        ok = np.array(ok)
        anchor = rng.choice(ok)
        d = np.abs(ok - anchor); d = np.minimum(d, n_kf - d)
        obs_kf = np.sort(ok[np.argsort(d, kind="stable")[:10]])
        q = np.array([0, 0, np.sin(yaw / 2), np.cos(yaw / 2)])
        if q[3] < 0:
            q = -q
        cub_qt_true[i] = np.concatenate([q, ctr]); cub_scale_true[i] = hs
        for j in obs_kf:
            uu, vv, zz = project(Rcw[j], tcw[j], corners)
            bb = np.array([(uu.max() + uu.min()) / 2, (vv.max() + vv.min()) / 2, uu.max() - uu.min(), vv.max() - vv.min()])
            bb_n = bb + rng.normal(size=4) * 2.0
            bbox_pose.append(j); bbox_cub.append(i); bbox_meas.append(bb_n)
            corner_meas.append((np.stack([uu, vv], axis=1) + rng.normal(size=(8, 2)) * 2.0).reshape(-1))
        yaw0 = yaw + rng.normal() * np.deg2rad(2.0)
        q0 = np.array([0, 0, np.sin(yaw0 / 2), np.cos(yaw0 / 2)])
        if q0[3] < 0:
            q0 = -q0
        c0 = ctr + np.array([rng.normal() * 0.03, 0.0, rng.normal() * 0.03])
        cub_qt[i] = np.concatenate([q0, c0]); cub_scale[i] = hs + rng.normal(size=3) * 0.02
    P.set("cuboid_qt", cub_qt); P.set("cuboid_scale", cub_scale)
    P.set("cuboid_flags", np.full(n_cb, 3, np.uint8)); P.set("cuboid_fixedscale", np.zeros((n_cb, 3)))
    n_bb = len(bbox_pose) if on["cuboid_2d"] else 0
    meas_quality, ba_weight_bbox, ba_weight_corner = 0.7, 1.0, 1.0
    P.set("bbox_pose", np.array(bbox_pose[:n_bb], dtype=np.int32)); P.set("bbox_cuboid", np.array(bbox_cub[:n_bb], dtype=np.int32))
    P.set("bbox_meas", np.array(bbox_meas[:n_bb]).reshape(n_bb, 4))
    P.set("bbox_info", np.full((n_bb, 4), (ba_weight_bbox * meas_quality) ** 2))
    P.set("bbox_huber", np.full(n_bb, float(np.float32(np.sqrt(80.0)))))
    P.set("bbox_K", np.tile(K.reshape(-1), (n_bb, 1)))
    n_co = len(bbox_pose) if on["corners_2d"] else 0
    P.set("corner_pose", np.array(bbox_pose[:n_co], dtype=np.int32)); P.set("corner_cuboid", np.array(bbox_cub[:n_co], dtype=np.int32))
    P.set("corner_meas", np.array(corner_meas[:n_co]).reshape(n_co, 16))
    P.set("corner_info", np.full((n_co, 16), (ba_weight_corner * meas_quality) ** 2))
    P.set("corner_huber", np.full(n_co, float(np.float32(np.sqrt(10.0)))))
    P.set("corner_K", np.tile(K.reshape(-1), (n_co, 1)))

    # point-cuboid unary edges: 30 frozen points per cuboid (Optimizer.cc:2642 needs > 10)
    n_pc = n_cb if on["pt_obj_3d"] else 0
    offs, pts = [0], []
    for i in range(n_pc):
        yaw = 2 * np.arctan2(cub_qt_true[i, 2], cub_qt_true[i, 3])
        Rwo = np.array([[np.cos(yaw), -np.sin(yaw), 0], [np.sin(yaw), np.cos(yaw), 0], [0, 0, 1]])
        local = rng.uniform(-1.15, 1.15, size=(30, 3)) * cub_scale_true[i]
        pts.append(f32(local @ Rwo.T + cub_qt_true[i, 4:] + rng.normal(size=(30, 3)) * 0.03))
        offs.append(offs[-1] + 30)
    P.set("ptcub_cuboid", np.arange(n_pc, dtype=np.int32)); P.set("ptcub_offsets", np.array(offs, dtype=np.int32))
    P.set("ptcub_points", np.concatenate(pts).reshape(-1, 3) if pts else np.zeros((0, 3)))
    P.set("ptcub_margin_ratio", np.ones(n_pc)); P.set("ptcub_prior_half", np.zeros((n_pc, 3)))
    P.set("ptcub_info", np.ones((n_pc, 3)))

    # cuboid-plane edges: constant-error terms (G2O_Plane3D.h:470-473)
    n_cp = min(n_cb, n_pl) if on["cuboid_plane"] else 0
    P.set("cubpl_cuboid", np.arange(n_cp, dtype=np.int32))
    P.set("cubpl_plane", rng.permutation(n_pl)[:n_cp].astype(np.int32) if n_cp else np.zeros(0, np.int32))
    P.set("cubpl_meas", rng.normal(size=(n_cp, 3)) * 0.01)
    P.set("cubpl_info", np.tile([3282.8 / (2.0 * 2.0), 3282.8 / (2.0 * 2.0), 100.0 * 100.0], (n_cp, 1)))
    P.set("cubpl_huber", np.full(n_cp, float(np.float32(np.sqrt(500.0)))))

    # 9-D camera-object edges (EdgeSE3Cuboid, Optimizer::LocalBACameraPointCuboids2D, Optimizer.cc:1767-1800): one per cuboid observation, the
    # measured cuboid in the CAMERA frame; every other measurement names another face "front" (a quarter / half turn about the box's z axis
    # with the x / y half extents swapped for the quarter turns), which is what cuboid::min_log_error exists for.  Own random stream: the
    # arrays above do not depend on this option.
    if with_cuboid3d and n_cb > 0:
        r3 = np.random.default_rng([int(seed), 3])
        c3_pose, c3_cub, c3_meas = [], [], []
        for e in range(len(bbox_pose)):
            j, i = int(bbox_pose[e]), int(bbox_cub[e])
            yaw = 2 * np.arctan2(cub_qt_true[i, 2], cub_qt_true[i, 3])
            turn = int(r3.integers(0, 4)) if e % 2 else 0
            yaw_m = yaw + turn * np.pi / 2 + r3.normal() * np.deg2rad(3.0) * noise_scale
            Rwo = np.array([[np.cos(yaw_m), -np.sin(yaw_m), 0], [np.sin(yaw_m), np.cos(yaw_m), 0], [0, 0, 1]])
            Rco = Rcw[j] @ Rwo
            tco = Rcw[j] @ cub_qt_true[i, 4:] + tcw[j] + r3.normal(size=3) * 0.05 * noise_scale
            hs = cub_scale_true[i].copy()
            if turn % 2:
                hs[[0, 1]] = hs[[1, 0]]
            hs = hs + r3.normal(size=3) * 0.02 * noise_scale
            q = quat_from_matrix(Rco)
            if q[3] < 0:
                q = -q
            c3_pose.append(j); c3_cub.append(i); c3_meas.append(np.concatenate([q, tco, hs]))
        n3 = len(c3_pose)
        P.set("cub3d_pose", np.array(c3_pose, dtype=np.int32)); P.set("cub3d_cuboid", np.array(c3_cub, dtype=np.int32))
        P.set("cub3d_meas", np.array(c3_meas).reshape(n3, 10))
        P.set("cub3d_info", np.full((n3, 9), (1.0 * 0.75) ** 2))   # (ba_weight_SE3 * 0.75)^2, Optimizer.cc:1785-1789
        P.set("cub3d_huber", np.full(n3, 900.0))                   # thHuberSE3 itself, Optimizer.cc:1793
    P.meta = dict(config=config if isinstance(config, str) else "custom", seed=seed,
                  truth=dict(Rcw=Rcw, tcw=tcw, points=X_true, planes=planes_true, cuboid_qt=cub_qt_true, cuboid_scale=cub_scale_true))
    return P
