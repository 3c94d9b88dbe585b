"""TEST INFRASTRUCTURE ONLY: ctypes loader for oracle/libppo_oracle.so (the CPU restatement).

Imported by tests/, by __graft_entry__.smoke() and by bench.py's cpu_baseline leg -- never by the product
package.  The oracle takes the same C structs as the product library, so the struct descriptions are reused
from the package's binding module.
"""
import ctypes as C
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(REPO, "point-plane-object-slam_amd"))
from ppo_ba import binding as B  # noqa: E402

LIB = os.path.join(HERE, "libppo_oracle.so")
_lib = None
c_dp = B.c_dp


def build():
    subprocess.check_call(["make", "-s", "-C", HERE])


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        lib = C.CDLL(LIB)
        lib.oracle_ba_reduced.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_int, C.c_int, c_dp, c_dp, B.c_ip]
        lib.oracle_ba_num_landmarks.argtypes = [C.c_void_p]
        lib.oracle_ldlt_solve.argtypes = [c_dp, C.c_int, c_dp, c_dp]
        lib.oracle_edge_eval.argtypes = [C.c_void_p, C.c_int, C.c_int, c_dp, c_dp, c_dp]
        lib.oracle_se3_exp.argtypes = [c_dp, c_dp]
        lib.oracle_pose_oplus.argtypes = [c_dp, c_dp, c_dp]
        lib.oracle_plane_oplus.argtypes = [c_dp, c_dp, c_dp]
        lib.oracle_plane_ominus.argtypes = [c_dp, c_dp, C.c_int, c_dp]
        lib.oracle_plane_transform.argtypes = [c_dp, c_dp, c_dp]
        lib.oracle_cuboid_oplus.argtypes = [c_dp, c_dp, c_dp, C.c_uint, c_dp, c_dp, c_dp]
        lib.oracle_cuboid_project.argtypes = [c_dp, c_dp, c_dp, c_dp, c_dp, c_dp]
        lib.oracle_huber.argtypes = [C.c_double, C.c_double, c_dp]
        for n in ("oracle_se3_exp", "oracle_pose_oplus", "oracle_plane_oplus", "oracle_plane_ominus", "oracle_plane_transform",
                  "oracle_cuboid_oplus", "oracle_cuboid_project", "oracle_huber"):
            getattr(lib, n).restype = None
        _lib = lib
    return _lib


def backend():
    return B._Backend(load(), "oracle_ba_")
