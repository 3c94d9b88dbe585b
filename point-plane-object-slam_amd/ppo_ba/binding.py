"""ctypes mirror of include/ppo_ba.h.

Describes ``ppo_ba_problem`` / ``ppo_ba_options`` / ``ppo_ba_result`` to Python and loads the HIP product library
(csrc/libppo_ba.so).  This module is plumbing for tests and bench.py; it contains no solver logic and no CPU
fallback: if the HIP library is missing, loading fails loudly.
"""
import ctypes as C
import os
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.dirname(HERE)
REPO_ROOT = os.path.dirname(PKG_ROOT)
LIB_PATH = os.environ.get("PPO_BA_LIB") or os.path.join(PKG_ROOT, "csrc", "libppo_ba.so")  # (PPO_BA_LIB: development aid, A/B builds)

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int32)
c_bp = C.POINTER(C.c_uint8)

# (field name, ctype, per-row width or None for scalars)
_PROBLEM_FIELDS = [
    ("n_poses", C.c_int32), ("pose_qt", c_dp), ("pose_fixed", c_bp), ("pose_cam", c_dp),
    ("n_points", C.c_int32), ("point_xyz", c_dp), ("point_fixed", c_bp),
    ("n_planes", C.c_int32), ("plane_coef", c_dp),
    ("n_cuboids", C.c_int32), ("cuboid_qt", c_dp), ("cuboid_scale", c_dp), ("cuboid_flags", c_bp), ("cuboid_fixedscale", c_dp),
    ("n_mono", C.c_int32), ("mono_point", c_ip), ("mono_pose", c_ip), ("mono_obs", c_dp), ("mono_info", c_dp), ("mono_huber", c_dp),
    ("n_stereo", C.c_int32), ("stereo_point", c_ip), ("stereo_pose", c_ip), ("stereo_obs", c_dp), ("stereo_info", c_dp), ("stereo_huber", c_dp),
    ("n_plane_edges", C.c_int32), ("pl_plane", c_ip), ("pl_pose", c_ip), ("pl_kind", c_bp), ("pl_meas", c_dp), ("pl_info", c_dp), ("pl_huber", c_dp),
    ("n_bbox", C.c_int32), ("bbox_pose", c_ip), ("bbox_cuboid", c_ip), ("bbox_meas", c_dp), ("bbox_info", c_dp), ("bbox_huber", c_dp), ("bbox_K", c_dp),
    ("n_corner", C.c_int32), ("corner_pose", c_ip), ("corner_cuboid", c_ip), ("corner_meas", c_dp), ("corner_info", c_dp), ("corner_huber", c_dp), ("corner_K", c_dp),
    ("n_ptcub", C.c_int32), ("ptcub_cuboid", c_ip), ("ptcub_offsets", c_ip), ("ptcub_points", c_dp), ("ptcub_margin_ratio", c_dp),
    ("ptcub_prior_half", c_dp), ("ptcub_info", c_dp), ("ptcub_huber", c_dp),
    ("n_cubpl", C.c_int32), ("cubpl_cuboid", c_ip), ("cubpl_plane", c_ip), ("cubpl_meas", c_dp), ("cubpl_info", c_dp), ("cubpl_huber", c_dp),
    ("n_cub3d", C.c_int32), ("cub3d_pose", c_ip), ("cub3d_cuboid", c_ip), ("cub3d_meas", c_dp), ("cub3d_info", c_dp), ("cub3d_huber", c_dp),
]


class CProblem(C.Structure):
    _fields_ = _PROBLEM_FIELDS


class COptions(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("iters_pass1", C.c_int32), ("iters_pass2", C.c_int32),
        ("lm_tau", C.c_double), ("lm_max_trials", C.c_int32), ("lm_lower", C.c_double), ("lm_upper", C.c_double),
        ("numeric_delta", C.c_double), ("jacobian_mode", C.c_int32),
        ("chi2_mono", C.c_double), ("chi2_stereo", C.c_double), ("chi2_plane", C.c_double), ("chi2_plane_vp", C.c_double),
        ("th_bbox_norm", C.c_double), ("th_corner_norm", C.c_double),
        ("deterministic", C.c_int32), ("solver", C.c_int32), ("eliminate_cuboids", C.c_int32),
        ("device_id", C.c_int32), ("rank", C.c_int32), ("world_size", C.c_int32),
        ("comm_id", C.c_uint8 * 128), ("verbose", C.c_int32), ("th_cub3d_norm", C.c_double), ("solve_on_owner", C.c_int32), ("lm_control", C.c_int32),
    ]


class CIterStat(C.Structure):
    _fields_ = [
        ("pass_", C.c_int32), ("iteration", C.c_int32), ("trials", C.c_int32), ("accepted", C.c_int32),
        ("chi2_before", C.c_double), ("chi2_after", C.c_double), ("lambda_", C.c_double),
        ("result", C.c_int32), ("pad_", C.c_int32),
    ]


MAX_ITER_STATS = 64


class CStats(C.Structure):
    _fields_ = [
        ("n_iters", C.c_int32), ("n_trials", C.c_int32), ("iters", CIterStat * MAX_ITER_STATS),
        ("n_reduced", C.c_int32), ("n_active_edges_pass1", C.c_int32), ("n_active_edges_pass2", C.c_int32),
        ("wall_ms_total", C.c_double), ("wall_ms_pass1", C.c_double), ("wall_ms_pass2", C.c_double), ("wall_ms_classify", C.c_double),
        ("ms_linearize", C.c_double), ("ms_schur", C.c_double), ("ms_factor", C.c_double), ("ms_backsub", C.c_double), ("ms_chi2", C.c_double),
        ("deterministic_fell_back", C.c_int32), ("pad_", C.c_int32),
    ]


_RESULT_FIELDS = [
    ("pose_qt", c_dp), ("point_xyz", c_dp), ("plane_coef", c_dp), ("cuboid_qt", c_dp), ("cuboid_scale", c_dp),
    ("mono_chi2", c_dp), ("mono_level", c_bp), ("mono_depth_positive", c_bp),
    ("stereo_chi2", c_dp), ("stereo_level", c_bp), ("stereo_depth_positive", c_bp),
    ("pl_chi2", c_dp), ("pl_level", c_bp),
    ("bbox_chi2", c_dp), ("bbox_errnorm", c_dp), ("bbox_level", c_bp),
    ("corner_chi2", c_dp), ("corner_errnorm", c_dp), ("corner_level", c_bp),
    ("ptcub_chi2", c_dp), ("cubpl_chi2", c_dp),
    ("cub3d_chi2", c_dp), ("cub3d_errnorm", c_dp), ("cub3d_level", c_bp),
]


class CResult(C.Structure):
    _fields_ = _RESULT_FIELDS


_DTYPES = {c_dp: np.float64, c_ip: np.int32, c_bp: np.uint8}

# rows / width of every array of ppo_ba_problem, keyed by the count field it follows
_WIDTH = {
    "pose_qt": ("n_poses", 7), "pose_fixed": ("n_poses", 1), "pose_cam": ("n_poses", 5),
    "point_xyz": ("n_points", 3), "point_fixed": ("n_points", 1),
    "plane_coef": ("n_planes", 4),
    "cuboid_qt": ("n_cuboids", 7), "cuboid_scale": ("n_cuboids", 3), "cuboid_flags": ("n_cuboids", 1), "cuboid_fixedscale": ("n_cuboids", 3),
    "mono_point": ("n_mono", 1), "mono_pose": ("n_mono", 1), "mono_obs": ("n_mono", 2), "mono_info": ("n_mono", 1), "mono_huber": ("n_mono", 1),
    "stereo_point": ("n_stereo", 1), "stereo_pose": ("n_stereo", 1), "stereo_obs": ("n_stereo", 3), "stereo_info": ("n_stereo", 1), "stereo_huber": ("n_stereo", 1),
    "pl_plane": ("n_plane_edges", 1), "pl_pose": ("n_plane_edges", 1), "pl_kind": ("n_plane_edges", 1), "pl_meas": ("n_plane_edges", 4),
    "pl_info": ("n_plane_edges", 3), "pl_huber": ("n_plane_edges", 1),
    "bbox_pose": ("n_bbox", 1), "bbox_cuboid": ("n_bbox", 1), "bbox_meas": ("n_bbox", 4), "bbox_info": ("n_bbox", 4), "bbox_huber": ("n_bbox", 1), "bbox_K": ("n_bbox", 9),
    "corner_pose": ("n_corner", 1), "corner_cuboid": ("n_corner", 1), "corner_meas": ("n_corner", 16), "corner_info": ("n_corner", 16),
    "corner_huber": ("n_corner", 1), "corner_K": ("n_corner", 9),
    "ptcub_cuboid": ("n_ptcub", 1), "ptcub_margin_ratio": ("n_ptcub", 1), "ptcub_prior_half": ("n_ptcub", 3), "ptcub_info": ("n_ptcub", 3), "ptcub_huber": ("n_ptcub", 1),
    "cubpl_cuboid": ("n_cubpl", 1), "cubpl_plane": ("n_cubpl", 1), "cubpl_meas": ("n_cubpl", 3), "cubpl_info": ("n_cubpl", 3), "cubpl_huber": ("n_cubpl", 1),
    "cub3d_pose": ("n_cub3d", 1), "cub3d_cuboid": ("n_cub3d", 1), "cub3d_meas": ("n_cub3d", 10), "cub3d_info": ("n_cub3d", 9), "cub3d_huber": ("n_cub3d", 1),
}
_OPTIONAL = {"point_fixed", "cuboid_flags", "cuboid_fixedscale", "pl_kind", "ptcub_prior_half", "ptcub_huber"}


class Problem:
    """Flattened factor graph held as numpy arrays; ``arrays`` maps ppo_ba_problem field names to ndarrays."""

    def __init__(self, **arrays):
        self.arrays = {}
        self.meta = {}
        for k, v in arrays.items():
            self.set(k, v)

    def set(self, name, value):
        ftype = dict(_PROBLEM_FIELDS)[name]
        self.arrays[name] = np.ascontiguousarray(value, dtype=_DTYPES[ftype])

    def count(self, count_field):
        for name, (cf, w) in _WIDTH.items():
            if cf == count_field and name in self.arrays and name not in _OPTIONAL:
                return self.arrays[name].size // w
        return 0

    def __getitem__(self, k):
        return self.arrays[k]

    def as_struct(self):
        s = CProblem()
        counts = {}
        for name, ftype in _PROBLEM_FIELDS:
            if ftype is C.c_int32:
                counts[name] = self.count(name)
                setattr(s, name, counts[name])
        for name, ftype in _PROBLEM_FIELDS:
            if ftype is C.c_int32:
                continue
            a = self.arrays.get(name)
            if name in ("ptcub_offsets", "ptcub_points"):
                if a is None:
                    a = np.zeros(1 if name == "ptcub_offsets" else 0, dtype=_DTYPES[ftype])
                    self.arrays[name] = a
            elif a is None:
                if name in _OPTIONAL:
                    continue
                cf, w = _WIDTH[name]
                if counts[cf] != 0:
                    raise ValueError("missing array " + name)
                a = np.zeros(0, dtype=_DTYPES[ftype])
                self.arrays[name] = a
            else:
                cf, w = _WIDTH[name]
                if a.size != counts[cf] * w:
                    raise ValueError("array %s has %d elements, expected %d" % (name, a.size, counts[cf] * w))
            setattr(s, name, a.ctypes.data_as(ftype))
        self._keepalive = s
        return s

    def sizes(self):
        st = self.as_struct()
        return {n: getattr(st, n) for n, t in _PROBLEM_FIELDS if t is C.c_int32}

    # ---- graph dump / replay (ppo_ba_problem_save / _load of the product library, host only) -----------------------------------
    def save(self, path):
        lib = load_library()
        lib.ppo_ba_problem_save.argtypes = [C.POINTER(CProblem), C.c_char_p]
        st = self.as_struct()
        rc = lib.ppo_ba_problem_save(C.byref(st), str(path).encode())
        if rc != 0:
            raise IOError("ppo_ba_problem_save(%s) failed: %d" % (path, rc))

    @classmethod
    def load(cls, path):
        lib = load_library()
        lib.ppo_ba_problem_load.argtypes = [C.c_char_p, C.POINTER(C.POINTER(CProblem))]
        lib.ppo_ba_problem_free.argtypes = [C.POINTER(CProblem)]
        lib.ppo_ba_problem_free.restype = None
        pp = C.POINTER(CProblem)()
        rc = lib.ppo_ba_problem_load(str(path).encode(), C.byref(pp))
        if rc != 0:
            raise IOError("ppo_ba_problem_load(%s) failed: %d" % (path, rc))
        try:
            st = pp.contents
            counts = {n: getattr(st, n) for n, t in _PROBLEM_FIELDS if t is C.c_int32}
            P = cls()
            for name, ftype in _PROBLEM_FIELDS:
                if ftype is C.c_int32:
                    continue
                ptr = getattr(st, name)
                if not ptr:
                    continue
                if name == "ptcub_offsets":
                    n = counts["n_ptcub"] + 1
                elif name == "ptcub_points":
                    n = 3 * st.ptcub_offsets[counts["n_ptcub"]] if counts["n_ptcub"] > 0 else 0
                else:
                    cf, w = _WIDTH[name]
                    n = counts[cf] * w
                P.set(name, np.ctypeslib.as_array(ptr, shape=(n,)).copy() if n > 0 else np.zeros(0, dtype=_DTYPES[ftype]))
            return P
        finally:
            lib.ppo_ba_problem_free(pp)


class Result:
    def __init__(self, sizes):
        z = lambda n, w=1, dt=np.float64: np.zeros((n, w) if w > 1 else (n,), dtype=dt)
        self.pose_qt = z(sizes["n_poses"], 7)
        self.point_xyz = z(sizes["n_points"], 3)
        self.plane_coef = z(sizes["n_planes"], 4)
        self.cuboid_qt = z(sizes["n_cuboids"], 7)
        self.cuboid_scale = z(sizes["n_cuboids"], 3)
        self.mono_chi2 = z(sizes["n_mono"]); self.mono_level = z(sizes["n_mono"], 1, np.uint8); self.mono_depth_positive = z(sizes["n_mono"], 1, np.uint8)
        self.stereo_chi2 = z(sizes["n_stereo"]); self.stereo_level = z(sizes["n_stereo"], 1, np.uint8); self.stereo_depth_positive = z(sizes["n_stereo"], 1, np.uint8)
        self.pl_chi2 = z(sizes["n_plane_edges"]); self.pl_level = z(sizes["n_plane_edges"], 1, np.uint8)
        self.bbox_chi2 = z(sizes["n_bbox"]); self.bbox_errnorm = z(sizes["n_bbox"]); self.bbox_level = z(sizes["n_bbox"], 1, np.uint8)
        self.corner_chi2 = z(sizes["n_corner"]); self.corner_errnorm = z(sizes["n_corner"]); self.corner_level = z(sizes["n_corner"], 1, np.uint8)
        self.ptcub_chi2 = z(sizes["n_ptcub"]); self.cubpl_chi2 = z(sizes["n_cubpl"])
        self.cub3d_chi2 = z(sizes["n_cub3d"]); self.cub3d_errnorm = z(sizes["n_cub3d"]); self.cub3d_level = z(sizes["n_cub3d"], 1, np.uint8)

    def as_struct(self):
        s = CResult()
        for name, ftype in _RESULT_FIELDS:
            setattr(s, name, getattr(self, name).ctypes.data_as(ftype))
        return s


def stats_to_dict(st):
    iters = []
    for i in range(min(st.n_iters, MAX_ITER_STATS)):
        it = st.iters[i]
        iters.append(dict(pass_=it.pass_, iteration=it.iteration, trials=it.trials, accepted=it.accepted,
                          chi2_before=it.chi2_before, chi2_after=it.chi2_after, lambda_=it.lambda_, result=it.result))
    d = {k: getattr(st, k) for k, _ in CStats._fields_ if k != "iters"}
    d["iters"] = iters
    return d


class _Backend:
    """Shared driver for a library exporting <prefix>_create/run/read/destroy over the common structs."""

    def __init__(self, lib, prefix):
        self.lib = lib
        self.prefix = prefix
        f = lambda n: getattr(lib, prefix + n)
        self._create, self._run, self._read, self._destroy = f("create"), f("run"), f("read"), f("destroy")
        self._reset, self._linearize = f("reset"), f("linearize")
        self._create.argtypes = [C.POINTER(CProblem), C.POINTER(COptions), C.POINTER(C.c_void_p)]
        self._create.restype = C.c_int
        self._run.argtypes = [C.c_void_p, c_bp, C.POINTER(CStats)]
        self._run.restype = C.c_int
        self._read.argtypes = [C.c_void_p, C.POINTER(CResult)]
        self._read.restype = C.c_int
        self._destroy.argtypes = [C.c_void_p]
        self._destroy.restype = None
        self._reset.argtypes = [C.c_void_p]
        self._reset.restype = C.c_int
        self._linearize.argtypes = [C.c_void_p, c_dp]
        self._linearize.restype = C.c_int

    def create(self, problem, options):
        h = C.c_void_p()
        ps = problem.as_struct()
        t0 = time.perf_counter()
        rc = self._create(C.byref(ps), C.byref(options), C.byref(h))
        c_ms = 1e3 * (time.perf_counter() - t0)
        if rc != 0:
            raise RuntimeError("%screate failed: %d" % (self.prefix, rc))
        s = Solver(self, h, problem)
        s.c_ms["create"] = c_ms  # (inside the C call: what a C++ caller pays -- the ctypes marshalling around it is this harness's, not the library's)
        return s


class Solver:
    def __init__(self, backend, handle, problem):
        self.backend, self.h, self.problem = backend, handle, problem
        self.sizes = problem.sizes()
        self.c_ms = {}  # milliseconds inside the last C call of each kind (create / run / read)

    def run(self, stop_flag=None):
        st = CStats()
        sf = stop_flag.ctypes.data_as(c_bp) if stop_flag is not None else None
        t0 = time.perf_counter()
        rc = self.backend._run(self.h, sf, C.byref(st))
        self.c_ms["run"] = 1e3 * (time.perf_counter() - t0)
        if rc < 0:
            raise RuntimeError("%srun failed: %d" % (self.backend.prefix, rc))
        self.last_rc = rc
        return stats_to_dict(st)

    def read(self):
        r = Result(self.sizes)
        rs = r.as_struct()
        t0 = time.perf_counter()
        rc = self.backend._read(self.h, C.byref(rs))
        self.c_ms["read"] = 1e3 * (time.perf_counter() - t0)
        if rc != 0:
            raise RuntimeError("%sread failed: %d" % (self.backend.prefix, rc))
        return r

    def reset(self):
        rc = self.backend._reset(self.h)
        if rc != 0:
            raise RuntimeError("reset failed: %d" % rc)

    def linearize(self):
        chi = C.c_double()
        rc = self.backend._linearize(self.h, C.byref(chi))
        if rc != 0:
            raise RuntimeError("linearize failed: %d" % rc)
        return chi.value

    def close(self):
        if self.h:
            self.backend._destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def default_options(**over):
    """The reference's constants (comments on ppo_ba_options in include/ppo_ba.h)."""
    o = COptions()
    o.abi_version = 3
    o.iters_pass1, o.iters_pass2 = 5, 10
    o.lm_tau, o.lm_max_trials, o.lm_lower, o.lm_upper = 1e-5, 10, 1.0 / 3.0, 2.0 / 3.0
    o.numeric_delta, o.jacobian_mode = 1e-9, 0
    o.chi2_mono, o.chi2_stereo, o.chi2_plane, o.chi2_plane_vp = 5.991, 7.815, 500.0, 200.0
    o.th_bbox_norm, o.th_corner_norm, o.th_cub3d_norm = 80.0, 10.0, 900.0
    o.deterministic, o.solver, o.eliminate_cuboids = 1, 0, 1
    o.device_id, o.rank, o.world_size, o.verbose = 0, 0, 1, 0
    for k, v in over.items():
        setattr(o, k, v)
    return o


_lib = None


TEST_LIB_PATH = os.path.join(PKG_ROOT, "csrc", "libppo_ba_test.so")  # built with -DPPO_BA_TEST_HOOKS (csrc/ppo_dev.hpp)
_test_lib = None


def _declare(lib):
    lib.ppo_ba_strerror.restype = C.c_char_p
    lib.ppo_ba_strerror.argtypes = [C.c_int]
    lib.ppo_ba_version.restype = C.c_int
    lib.ppo_ba_linearize_bytes.restype = C.c_int64
    lib.ppo_ba_linearize_bytes.argtypes = [C.c_void_p]
    lib.ppo_ba_time_linearize.argtypes = [C.c_void_p, C.c_int, c_dp]
    lib.ppo_ba_time_factor.argtypes = [C.c_void_p, C.c_int, c_dp]
    lib.ppo_ba_overlap_schedule.argtypes = [C.c_void_p, c_ip, C.c_int32]
    lib.ppo_ba_time_kernel.argtypes = [C.c_void_p, C.c_int, C.c_int, c_dp, C.POINTER(C.c_int64)]
    lib.ppo_ba_debug_reduced.argtypes = [C.c_void_p, c_dp, c_dp, c_dp, c_ip]
    lib.ppo_ba_partition.argtypes = [C.POINTER(CProblem), C.c_int32, c_ip, c_ip]
    lib.ppo_ba_comm_unique_id.argtypes = [c_bp]
    lib.ppo_ba_default_options.argtypes = [C.POINTER(COptions)]
    lib.ppo_ba_default_options.restype = None
    return lib


def load_test_library():
    """The build with the development / test switches compiled in (A/B environment switches, loopback communicator)."""
    global _test_lib
    if _test_lib is None:
        if not os.path.exists(TEST_LIB_PATH):
            raise RuntimeError("libppo_ba_test.so is missing (%s): run `python -c 'import __graft_entry__ as g; g.build()'`" % TEST_LIB_PATH)
        _test_lib = _declare(C.CDLL(TEST_LIB_PATH))
    return _test_lib


def load_library(path=LIB_PATH):
    """Load the HIP product library.  Raises if it has not been built -- there is no CPU fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(path):
            raise RuntimeError("libppo_ba.so is missing (%s): run `python -c 'import __graft_entry__ as g; g.build()'`" % path)
        lib = C.CDLL(path)  # RTLD_LOCAL: the HIP runtime it pulls in must not shadow the one bundled with torch
        lib.ppo_ba_strerror.restype = C.c_char_p
        lib.ppo_ba_strerror.argtypes = [C.c_int]
        lib.ppo_ba_version.restype = C.c_int
        lib.ppo_ba_linearize_bytes.restype = C.c_int64
        lib.ppo_ba_linearize_bytes.argtypes = [C.c_void_p]
        lib.ppo_ba_time_linearize.argtypes = [C.c_void_p, C.c_int, c_dp]
        lib.ppo_ba_time_factor.argtypes = [C.c_void_p, C.c_int, c_dp]
        lib.ppo_ba_overlap_schedule.argtypes = [C.c_void_p, c_ip, C.c_int32]
        lib.ppo_ba_time_kernel.argtypes = [C.c_void_p, C.c_int, C.c_int, c_dp, C.POINTER(C.c_int64)]
        lib.ppo_ba_debug_reduced.argtypes = [C.c_void_p, c_dp, c_dp, c_dp, c_ip]
        lib.ppo_ba_partition.argtypes = [C.POINTER(CProblem), C.c_int32, c_ip, c_ip]
        lib.ppo_ba_comm_unique_id.argtypes = [c_bp]
        lib.ppo_ba_default_options.argtypes = [C.POINTER(COptions)]
        lib.ppo_ba_default_options.restype = None
        _lib = lib
    return _lib


def hip_backend(test_hooks=False):
    return _Backend(load_test_library() if test_hooks else load_library(), "ppo_ba_")
