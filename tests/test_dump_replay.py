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
    with pytest.raises(IOError):
        B.Problem.load(tmp_path / "does_not_exist.ppoba")
