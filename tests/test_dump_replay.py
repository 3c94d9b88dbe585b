"""Graph dump / replay: ppo_ba_problem_save / _load of the product library are host-only, so the round
trip is checked here without a GPU; the loaded graph must give the oracle exactly the same run as the original."""
import os

import numpy as np
import pytest

from ppo_ba import binding as B
from ppo_ba import synth


def test_round_trip_preserves_every_array(tmp_path):
    P = synth.make_problem("small", seed=3, with_corners=True, stereo_frac=0.2)
    path = tmp_path / "window.ppoba"
    P.save(path)
    Q = B.Problem.load(path)
    assert P.sizes() == Q.sizes()
    for name, a in P.arrays.items():
        if a.size == 0:
            continue
        assert name in Q.arrays, name
        assert a.dtype == Q.arrays[name].dtype
        np.testing.assert_array_equal(a.ravel(), Q.arrays[name].ravel(), err_msg=name)


def test_loaded_graph_replays_identically_on_the_oracle(tmp_path, oracle):
    P = synth.make_problem("tiny", seed=4)
    path = tmp_path / "w.ppoba"
    P.save(path)
    Q = B.Problem.load(path)
    sa, sb = oracle.create(P, B.default_options()), oracle.create(Q, B.default_options())
    ta, tb = sa.run(), sb.run()
    assert [i["chi2_after"] for i in ta["iters"]] == [i["chi2_after"] for i in tb["iters"]]
    np.testing.assert_array_equal(sa.read().pose_qt, sb.read().pose_qt)


def test_truncated_and_foreign_files_are_rejected(tmp_path):
    P = synth.make_problem("tiny", seed=1)
    path = tmp_path / "w.ppoba"
    P.save(path)
    raw = open(path, "rb").read()
    bad = tmp_path / "bad.ppoba"
    open(bad, "wb").write(raw[: len(raw) // 2])
    with pytest.raises(IOError):
        B.Problem.load(bad)
    open(bad, "wb").write(b"NOTPPOBA" + raw[8:])
    with pytest.raises(IOError):
        B.Problem.load(bad)
    # a count that no longer matches its arrays (n_mono patched upwards) must not get through either
    i = raw.index(b"n_mono") + len(b"n_mono") + 4 + 8
    patched = raw[:i] + (np.int32(P.sizes()["n_mono"] + 5)).tobytes() + raw[i + 4:]
    open(bad, "wb").write(patched)
    with pytest.raises(IOError):
        B.Problem.load(bad)
    # a record that claims more elements than the file has bytes left is refused BEFORE anything is allocated for it
    j = raw.index(b"pose_qt") + len(b"pose_qt") + 4
    huge = raw[:j] + np.uint64(1 << 32).tobytes() + raw[j + 8:]
    open(bad, "wb").write(huge)
    with pytest.raises(IOError):
        B.Problem.load(bad)
    with pytest.raises(IOError):
        B.Problem.load(tmp_path / "does_not_exist.ppoba")


def test_end_of_run_exports_follow_the_reference_formats(tmp_path):
    """ppo_ba_export_tum / _cuboids / _planes: the text files System::SaveKeyFrameTrajectoryTUM, SaveCuboidOptimized and
    SavePlaneOptimized write (reference src/System.cc:402-494), from the arrays a ppo_ba_result holds."""
    import ctypes as C
    lib = B.load_library()
    rng = np.random.default_rng(5)
    n = 4
    q = rng.normal(size=(n, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True); q[q[:, 3] < 0] *= -1
    t = rng.normal(size=(n, 3))
    pose = np.ascontiguousarray(np.hstack([q, t]))
    pose[0] = [0, 0, 0, 1, 0, 0, 0]
    stamps = np.array([0.0, 0.033333, 1.5, 1305031102.175304])
    path = str(tmp_path / "KeyFrameTrajectory.txt").encode()
    lib.ppo_ba_export_tum.argtypes = [C.c_char_p, C.c_int32, B.c_dp, B.c_dp]
    assert lib.ppo_ba_export_tum(path, n, stamps.ctypes.data_as(B.c_dp), pose.ctypes.data_as(B.c_dp)) == 0
    rows = [ln.split() for ln in open(path.decode())]
    assert len(rows) == n and all(len(r) == 8 for r in rows)
    assert rows[0] == ["0.000000"] + ["0.0000000"] * 6 + ["1.0000000"]
    assert rows[3][0] == "1305031102.175304"
    for i in range(n):
        x, y, z, w = q[i] if i else (0, 0, 0, 1)
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                      [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        centre = -R.T @ (t[i] if i else np.zeros(3))
        got = np.array([float(v) for v in rows[i][1:]])
        np.testing.assert_allclose(got[:3], centre, atol=2e-7 + 1e-6 * np.abs(centre).max())
        np.testing.assert_allclose(got[3:], [-x, -y, -z, w], atol=2e-7)  # camera-to-world rotation = conjugate, w >= 0
    # cuboids: x y z roll pitch yaw sx sy sz of an upright box turned by a known yaw
    yaw = 0.7
    cq = np.array([[0, 0, np.sin(yaw / 2), np.cos(yaw / 2), 1.0, 2.0, 3.0]])
    cs = np.array([[0.1, 0.2, 0.3]])
    cpath = str(tmp_path / "CuboidPose.txt").encode()
    lib.ppo_ba_export_cuboids.argtypes = [C.c_char_p, C.c_int32, C.POINTER(C.c_char_p), B.c_dp, B.c_dp]
    names = (C.c_char_p * 1)(b"chair")
    assert lib.ppo_ba_export_cuboids(cpath, 1, names, cq.ctypes.data_as(B.c_dp), cs.ctypes.data_as(B.c_dp)) == 0
    assert open(cpath.decode()).read() == "chair 1.000000 2.000000 3.000000 0.000000 0.000000 0.700000 0.100000 0.200000 0.300000\n"
    assert lib.ppo_ba_export_cuboids(cpath, 1, None, cq.ctypes.data_as(B.c_dp), cs.ctypes.data_as(B.c_dp)) == 0
    assert open(cpath.decode()).read().startswith("object 1.000000")
    ppath = str(tmp_path / "PlanePose.txt").encode()
    pl = np.array([[0.0, 0.0, 1.0, 1.25], [0.6, 0.8, 0.0, 2.0]])
    lib.ppo_ba_export_planes.argtypes = [C.c_char_p, C.c_int32, B.c_dp]
    assert lib.ppo_ba_export_planes(ppath, 2, pl.ctypes.data_as(B.c_dp)) == 0
    assert open(ppath.decode()).read() == "0 0.000000 0.000000 1.000000 1.250000\n1 0.600000 0.800000 0.000000 2.000000\n"
    assert lib.ppo_ba_export_planes(b"/nonexistent-dir/x.txt", 2, pl.ctypes.data_as(B.c_dp)) == -1
