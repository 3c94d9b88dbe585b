"""CPU-side checks of the drop-in boundary: the library loads, exports every symbol include/ppo_ba.h declares,
the host-only entry points work, and the product path fails loudly (no CPU fallback) when there is no GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from ppo_ba import binding as B
from ppo_ba import synth

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(REPO, "include", "ppo_ba.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ppo_ba_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_exported():
    lib = B.load_library()
    names = declared_functions()
    assert "ppo_ba_create" in names and "ppo_ba_run" in names and "ppo_ba_read" in names and len(names) >= 14
    for n in names:
        assert hasattr(lib, n), "libppo_ba.so does not export " + n


def test_the_c_abi_is_the_only_dynamic_symbol_set():
    """The library is loaded into a process that already holds g2o, Eigen, OpenCV and possibly other ROCm libraries: nothing but the entry points of
    include/ppo_ba.h and include/ppo_pose.h may be a dynamic symbol (csrc/ppo_ba.map; -fvisibility=hidden) -- no rocprim trampoline, no ppo:: function,
    no weak STL instantiation that could interpose on, or be interposed by, the host's."""
    import subprocess
    declared = set(declared_functions())
    pose = re.sub(r"/\*.*?\*/", "", open(os.path.join(REPO, "include", "ppo_pose.h")).read(), flags=re.S)
    declared |= set(re.findall(r"\b(ppo_pose_[a-z_0-9]+)\s*\(", pose))
    for name in ("libppo_ba.so", "libppo_ba_test.so"):
        path = os.path.join(REPO, "point-plane-object-slam_amd", "csrc", name)
        out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
        exported = {line.split()[-1] for line in out.splitlines() if line.strip()}
        assert exported == declared, (name, sorted(exported ^ declared))


def test_warmup_fails_loudly_without_a_gpu():
    """ppo_ba_warmup goes through ppo_ba_create: no device, no silent success."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = B.load_library()
    lib.ppo_ba_warmup.restype = C.c_int
    lib.ppo_ba_warmup.argtypes = [C.c_int]
    assert lib.ppo_ba_warmup(0) == -4  # PPO_BA_ERR_HIP


def test_version_and_strerror():
    lib = B.load_library()
    assert lib.ppo_ba_version() == 3  # 3: ppo_ba_stats.deterministic_fell_back, options.deterministic defaults to 1
    assert lib.ppo_ba_strerror(0) == b"ok"
    assert b"HIP" in lib.ppo_ba_strerror(-4)


def test_default_options_match_reference_constants():
    lib = B.load_library()
    o = B.COptions()
    lib.ppo_ba_default_options(C.byref(o))
    assert (o.iters_pass1, o.iters_pass2, o.lm_max_trials) == (5, 10, 10)  # Optimizer.cc:2728,2837; levenberg.cpp:51
    assert o.lm_tau == 1e-5 and o.numeric_delta == 1e-9
    assert (o.chi2_mono, o.chi2_stereo, o.chi2_plane, o.chi2_plane_vp) == (5.991, 7.815, 500.0, 200.0)
    assert (o.th_bbox_norm, o.th_corner_norm, o.th_cub3d_norm) == (80.0, 10.0, 900.0)  # Parameters.cc:63-65
    p = B.default_options()
    for k, _ in B.COptions._fields_:
        if k not in ("comm_id",):
            assert getattr(o, k) == getattr(p, k), k


def test_struct_sizes_match_header():
    # sizes derived from the C declarations: guards against silent layout drift between header and ctypes mirror
    assert C.sizeof(B.CIterStat) == 4 * 4 + 3 * 8 + 2 * 4
    assert C.sizeof(B.CStats) == 8 + 64 * C.sizeof(B.CIterStat) + 4 * 4 + 9 * 8 + 2 * 4
    assert C.sizeof(B.CResult) == 24 * 8
    # ... and against the library itself: sizeof() of every ABI struct as the C compiler laid it out
    lib = B.load_library()
    lib.ppo_ba_sizeof.restype = C.c_int64
    lib.ppo_ba_sizeof.argtypes = [C.c_int32]
    for which, mirror in enumerate((B.CProblem, B.COptions, B.CStats, B.CResult, B.CIterStat)):
        assert lib.ppo_ba_sizeof(which) == C.sizeof(mirror), (which, mirror.__name__)
    assert lib.ppo_ba_sizeof(99) == -1


def test_partition_is_contiguous_and_balanced():
    lib = B.load_library()
    P = synth.make_problem("small", seed=3)
    ps = P.as_struct()
    for world in (1, 2, 3, 8):
        fp = np.zeros(world + 1, np.int32)
        fl = np.zeros(world + 1, np.int32)
        rc = lib.ppo_ba_partition(C.byref(ps), world, fp.ctypes.data_as(B.c_ip), fl.ctypes.data_as(B.c_ip))
        assert rc == 0
        assert fp[0] == 0 and fp[-1] == ps.n_points and np.all(np.diff(fp) >= 0)
        assert fl[0] == 0 and fl[-1] == ps.n_planes and np.all(np.diff(fl) >= 0)
        counts = np.bincount(P["mono_point"], minlength=ps.n_points)
        per_rank = [counts[fp[r]:fp[r + 1]].sum() for r in range(world)]
        assert max(per_rank) - min(per_rank) <= 0.2 * counts.sum() / world + 16


def test_bad_arguments_are_rejected_before_touching_the_device():
    lib = B.load_library()
    P = synth.make_problem("tiny")
    ps = P.as_struct()
    h = C.c_void_p()
    o = B.default_options()
    assert lib.ppo_ba_create(None, C.byref(o), C.byref(h)) == -1
    bad = B.default_options(abi_version=99)
    assert lib.ppo_ba_create(C.byref(ps), C.byref(bad), C.byref(h)) == -1
    P2 = synth.make_problem("tiny")
    a = P2["mono_pose"].copy()
    a[0] = 10_000
    P2.set("mono_pose", a)
    ps2 = P2.as_struct()
    assert lib.ppo_ba_create(C.byref(ps2), C.byref(o), C.byref(h)) == -1
    P3 = synth.make_problem("tiny")
    x = P3["point_xyz"].copy()
    x[5] = np.nan
    P3.set("point_xyz", x)
    ps3 = P3.as_struct()
    assert lib.ppo_ba_create(C.byref(ps3), C.byref(o), C.byref(h)) == -2
    # options that exist in the struct but are not implemented fail loudly instead of being ignored
    assert lib.ppo_ba_create(C.byref(ps), C.byref(B.default_options(deterministic=2)), C.byref(h)) == -1
    assert lib.ppo_ba_create(C.byref(ps), C.byref(B.default_options(jacobian_mode=2)), C.byref(h)) == -1  # 0 numeric, 1 analytic
    assert lib.ppo_ba_create(C.byref(ps), C.byref(B.default_options(rank=1, world_size=1)), C.byref(h)) == -1
    assert lib.ppo_ba_create(C.byref(ps), C.byref(B.default_options(rank=0, world_size=0)), C.byref(h)) == -1
    lib.ppo_ba_destroy(None)  # NULL-safe


def test_no_cpu_fallback():
    """Without a GPU the product must refuse to run (PPO_BA_ERR_HIP); with one it must succeed."""
    lib = B.load_library()
    P = synth.make_problem("tiny")
    ps = P.as_struct()
    h = C.c_void_p()
    o = B.default_options()
    rc = lib.ppo_ba_create(C.byref(ps), C.byref(o), C.byref(h))
    if os.path.exists("/dev/kfd"):  # a ROCm device node: the GPU box
        assert rc == 0
        lib.ppo_ba_destroy(h)
    else:
        assert rc == -4 and not h.value


def test_product_library_carries_no_experiment_code():
    """The one-launch Cholesky experiment, the timing-only macros and the loopback communicator exist only in the
    test-hooks build (csrc/libppo_ba_test.so): neither symbol tables nor device code objects of libppo_ba.so name them."""
    import subprocess
    so = os.path.join(REPO, "point-plane-object-slam_amd", "csrc", "libppo_ba.so")
    sym = subprocess.run(["nm", "-C", so], capture_output=True, text=True).stdout
    blob = open(so, "rb").read()
    for name in ("chol_fwd_all", "k_loop_reduce", "LoopGroup", "loop_reduce"):
        assert name not in sym, name
        assert name.encode() not in blob, name
    for env in (b"PPO_BA_CHOL_ONELAUNCH", b"PPO_BA_CHOL_GRAPH", b"PPO_BA_COMM_SELFTEST", b"PPO_BA_LIN_PARTS"):
        assert env not in blob, env


def test_overlap_schedule_keeps_its_invariants():
    """The schedule of the overlapped reduction (host-only rule, no GPU): for random chunk distributions over the column blocks
    fin is non-decreasing, ends at the number of blocks, never lets a panel launch read a block that is not final (fin[k] >= k + 1), starts
    with the requested blocks, and never gives a launch more chunks than its share unless the k + 1 rule forces it."""
    lib = B.load_library()
    lib.ppo_ba_overlap_plan.argtypes = [C.c_int32, B.c_ip, C.c_int32, C.c_double, C.c_double, B.c_ip]
    rng = np.random.default_rng(7)
    for case in range(300):
        nb = int(rng.integers(1, 80))
        per_block = rng.integers(0, 400, size=nb) * (rng.random(nb) < 0.9)
        cb = np.concatenate([[0], np.cumsum(per_block)]).astype(np.int32)
        first = int(rng.integers(1, 4))
        share0, share = float(rng.choice([0.0, 0.05, 0.14, 0.4, 1.0])), float(rng.choice([0.0, 0.05, 0.1, 0.3, 1.0]))
        fin = np.zeros(nb + 1, np.int32)
        assert lib.ppo_ba_overlap_plan(nb, cb.ctypes.data_as(B.c_ip), first, share0, share, fin.ctypes.data_as(B.c_ip)) == 0
        assert fin[0] == min(nb, max(1, first)) and fin[nb] == nb and np.all(np.diff(fin) >= 0), (case, fin)
        total = int(cb[nb])
        for k in range(nb):
            assert fin[k] >= min(nb, k + 1), (case, k, fin)
            carried = int(cb[fin[k + 1]] - cb[fin[k]])
            budget = int((share0 if k == 0 else share) * total)
            need = min(nb, k + 2)
            if carried > budget:  # only the blocks the next launch must find final, never more
                assert fin[k + 1] <= max(need, fin[k]) or int(cb[fin[k + 1]] - cb[fin[k]]) == int(cb[max(need, fin[k])] - cb[fin[k]]), (case, k, fin)
    bad = np.array([0, 5, 3], np.int32)
    fin = np.zeros(3, np.int32)
    assert lib.ppo_ba_overlap_plan(2, bad.ctypes.data_as(B.c_ip), 1, 0.1, 0.1, fin.ctypes.data_as(B.c_ip)) != 0

