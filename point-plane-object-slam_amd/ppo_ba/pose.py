"""ctypes mirror of include/ppo_pose.h (pose-only optimisation, Optimizer::PoseOptimization) + a synthetic frame generator.

Tests and bench only, like binding.py: no compute of its own.  `optimize(lib, frames)` works with the product library
(binding.load_library()) and with the oracle (oracle_py.load(), prefix "oracle_") because both take the same structs.
"""
import ctypes as C

import numpy as np

from .binding import c_bp, c_dp, c_ip


class CPoseProblem(C.Structure):
    _fields_ = [("n_frames", C.c_int32), ("pose_qt", c_dp), ("cam", c_dp), ("edge_offsets", c_ip), ("xw", c_dp), ("obs", c_dp),
                ("inv_sigma2", c_dp)]


class CPoseOptions(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("device_id", C.c_int32), ("rounds", C.c_int32), ("iters_per_round", C.c_int32),
                ("robust_rounds", C.c_int32), ("min_correspondences", C.c_int32), ("min_edges", C.c_int32),
                ("chi2_mono", C.c_float), ("chi2_stereo", C.c_float), ("huber_mono", C.c_float), ("huber_stereo", C.c_float),
                ("lm_tau", C.c_double), ("lm_max_trials", C.c_int32), ("lm_lower", C.c_double), ("lm_upper", C.c_double)]


class CPoseStats(C.Structure):
    _fields_ = [("rounds_run", C.c_int32), ("lm_iterations", C.c_int32), ("lm_trials", C.c_int32), ("n_bad", C.c_int32),
                ("chi2_initial", C.c_double), ("chi2_final", C.c_double), ("lambda_final", C.c_double)]


class CPoseResult(C.Structure):
    _fields_ = [("pose_qt", c_dp), ("outlier", c_bp), ("n_inliers", c_ip), ("chi2", c_dp), ("stats", C.POINTER(CPoseStats))]


def default_options(**over):
    """The reference's constants (src/Optimizer.cc:281-282, 372, 377-382, 415, 448)."""
    o = CPoseOptions()
    o.abi_version, o.device_id = 1, 0
    o.rounds, o.iters_per_round, o.robust_rounds, o.min_correspondences, o.min_edges = 4, 10, 3, 3, 10
    o.chi2_mono, o.chi2_stereo = 5.991, 7.815
    o.huber_mono, o.huber_stereo = np.float32(np.sqrt(5.991)), np.float32(np.sqrt(7.815))
    o.lm_tau, o.lm_max_trials, o.lm_lower, o.lm_upper = 1e-5, 10, 1.0 / 3.0, 2.0 / 3.0
    for k, v in over.items():
        setattr(o, k, v)
    return o


class Frames:
    """SoA batch of frames: pose_qt [F,7], cam [F,5], edge_offsets [F+1], xw [E,3], obs [E,3] (u, v, u_right or -1), inv_sigma2 [E]."""

    def __init__(self, pose_qt, cam, edge_offsets, xw, obs, inv_sigma2):
        self.pose_qt = np.ascontiguousarray(pose_qt, np.float64).reshape(-1, 7)
        self.cam = np.ascontiguousarray(cam, np.float64).reshape(-1, 5)
        self.edge_offsets = np.ascontiguousarray(edge_offsets, np.int32)
        self.xw = np.ascontiguousarray(xw, np.float64).reshape(-1, 3)
        self.obs = np.ascontiguousarray(obs, np.float64).reshape(-1, 3)
        self.inv_sigma2 = np.ascontiguousarray(inv_sigma2, np.float64).reshape(-1)

    @property
    def n_frames(self):
        return self.pose_qt.shape[0]

    @property
    def n_edges(self):
        return self.xw.shape[0]

    def as_struct(self):
        p = CPoseProblem()
        p.n_frames = self.n_frames
        p.pose_qt = self.pose_qt.ctypes.data_as(c_dp)
        p.cam = self.cam.ctypes.data_as(c_dp)
        p.edge_offsets = self.edge_offsets.ctypes.data_as(c_ip)
        p.xw = self.xw.ctypes.data_as(c_dp)
        p.obs = self.obs.ctypes.data_as(c_dp)
        p.inv_sigma2 = self.inv_sigma2.ctypes.data_as(c_dp)
        return p


def optimize(lib, frames, options=None, prefix="ppo_"):
    """Call <prefix>pose_optimize; returns a dict of numpy arrays + the per-frame stats as dicts.  Raises on an error code."""
    fn = getattr(lib, prefix + "pose_optimize")
    fn.argtypes = [C.POINTER(CPoseProblem), C.POINTER(CPoseOptions), C.POINTER(CPoseResult)]
    fn.restype = C.c_int
    o = options if options is not None else default_options()
    F, E = frames.n_frames, frames.n_edges
    pose = np.zeros((F, 7))
    outlier = np.zeros(E, np.uint8)
    n_inl = np.zeros(F, np.int32)
    chi2 = np.zeros(E)
    stats = (CPoseStats * max(F, 1))()
    r = CPoseResult()
    r.pose_qt = pose.ctypes.data_as(c_dp)
    r.outlier = outlier.ctypes.data_as(c_bp)
    r.n_inliers = n_inl.ctypes.data_as(c_ip)
    r.chi2 = chi2.ctypes.data_as(c_dp)
    r.stats = stats
    p = frames.as_struct()
    rc = fn(C.byref(p), C.byref(o), C.byref(r))
    if rc != 0:
        raise RuntimeError("%spose_optimize failed: %d" % (prefix, rc))
    st = [{k: getattr(stats[f], k) for k, _ in CPoseStats._fields_} for f in range(F)]
    return {"pose_qt": pose, "outlier": outlier, "n_inliers": n_inl, "chi2": chi2, "stats": st}


# ---- synthetic frames -----------------------------------------------------------------------------------------------------
def _quat_of(R):
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
        q = np.zeros(4)
        q[i] = 0.25 * s
        q[j] = (R[j, i] + R[i, j]) / s
        q[k] = (R[k, i] + R[i, k]) / s
        q[3] = (R[k, j] - R[j, k]) / s
    return q if q[3] >= 0 else -q


def _rot(w):
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-12:
        return np.eye(3) + K
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * K @ K


def make_frames(n_frames=1, n_obs=400, stereo_fraction=0.5, outlier_fraction=0.1, seed=0, pose_noise=(0.02, 0.05), bf=40.0):
    """Frames the way the Tracking thread hands them over: map points (float32 positions) seen by a camera whose pose estimate
    is off by pose_noise (rad, m); pixel noise N(0, 1 px) * 1.2^octave, inv_sigma2 = 1.2^(-2 octave) (src/ORBextractor.cc:421-430);
    a fraction of gross mismatches.  n_obs may be an int or a per-frame list."""
    rng = np.random.default_rng(seed)
    counts = [n_obs] * n_frames if np.isscalar(n_obs) else list(n_obs)
    fx, fy, cx, cy = 480.0, 480.0, 320.0, 240.0
    poses, cams, off, xw, obs, inv = [], [], [0], [], [], []
    for f in range(n_frames):
        R = _rot(rng.normal(size=3) * 0.4)
        t = rng.normal(size=3) * 1.5
        n = counts[f]
        pc = np.stack([rng.uniform(-1, 1, n), rng.uniform(-0.8, 0.8, n), np.ones(n)], 1) * rng.uniform(0.5, 8.0, (n, 1))
        Xw = ((pc - t) @ R).astype(np.float32).astype(np.float64)  # R^T (pc - t); the reference keeps map points in float32
        pcf = Xw @ R.T + t
        octave = rng.integers(0, 8, n)
        sig = 1.2 ** octave
        u = fx * pcf[:, 0] / pcf[:, 2] + cx + rng.normal(size=n) * sig
        v = fy * pcf[:, 1] / pcf[:, 2] + cy + rng.normal(size=n) * sig
        ur = u - bf / pcf[:, 2] + rng.normal(size=n) * sig
        is_stereo = rng.uniform(size=n) < stereo_fraction
        bad = rng.uniform(size=n) < outlier_fraction
        u = np.where(bad, rng.uniform(0, 640, n), u)
        v = np.where(bad, rng.uniform(0, 480, n), v)
        ur = np.where(is_stereo, np.where(bad, u - rng.uniform(1, 60, n), ur), -1.0)
        dR = _rot(rng.normal(size=3) * pose_noise[0])
        q = _quat_of(dR @ R)
        poses.append(np.concatenate([q, dR @ t + rng.normal(size=3) * pose_noise[1]]))
        cams.append([fx, fy, cx, cy, bf])
        off.append(off[-1] + n)
        xw.append(Xw)
        obs.append(np.stack([u, v, ur], 1).astype(np.float32).astype(np.float64))  # cv::KeyPoint / mvuRight are float32
        inv.append((1.0 / sig ** 2).astype(np.float32).astype(np.float64))
    cat = lambda xs, w: np.concatenate(xs) if xs and sum(len(x) for x in xs) else np.zeros((0, w))
    return Frames(np.array(poses).reshape(-1, 7), np.array(cams).reshape(-1, 5), off, cat(xw, 3), cat(obs, 3),
                  np.concatenate(inv) if inv and sum(len(x) for x in inv) else np.zeros(0))
