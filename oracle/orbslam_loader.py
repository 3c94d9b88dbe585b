#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (build container only): load the reference's OWN prebuilt lib/libORB_SLAM2.so with ctypes.

The library was linked against 26 shared objects this image does not have (OpenCV 3.3, PCL 1.8, VTK 6.3, Pangolin, boost 1.65,
GLEW, DBoW2, tictoc_profiler).  None of them is touched by the functions pinned here -- g2o::Plane3D / VertexPlane / EdgePlane*,
g2o::cuboid / VertexCuboid / EdgeSE3Cuboid*, BlockSolver<BlockSolverTraits<-1,-1>>, LinearSolverDense<MatrixXd> -- whose code
depends on libg2o.so (the reference's own prebuilt Thirdparty/g2o/lib/libg2o.so, which loads as it is), libstdc++, libm and
Eigen templates that are compiled INTO libORB_SLAM2.so.  To get the dynamic linker past the missing DT_NEEDED entries this module

  1. writes, into a fresh temporary directory, one EMPTY shared object per missing soname (no code, no symbols), and
  2. one `libsink.so` that defines every symbol libORB_SLAM2.so imports and no present library provides (cv::*, pcl::*, pangolin::*,
     gl*, vtk*, DBoW2::* ...) as a TRAP: while `sink_quiet` is non-zero (only during dlopen, for the library's static
     initialisers, and at exit for its static destructors) a trap returns 0 and counts the call; once armed it prints the symbol's
     name and abort()s.

So nothing stands in for reference code: any pinned call that reached outside reference instructions (+ libstdc++ / libm / libc)
would kill the process with the offending name.  `armed_hits()` is 0 by construction for every number written to tests/golden/.
Nothing produced here is kept or travels: the temporary directory is removed at exit, and the GPU box never runs this module
(/root/reference does not exist there).
"""
import atexit
import ctypes as C
import os
import shutil
import subprocess
import tempfile

REF = "/root/reference"
ORB = os.path.join(REF, "lib", "libORB_SLAM2.so")
G2O = os.path.join(REF, "Thirdparty", "g2o", "lib", "libg2o.so")
_SYSTEM = ["/lib/x86_64-linux-gnu/libstdc++.so.6", "/lib/x86_64-linux-gnu/libm.so.6", "/lib/x86_64-linux-gnu/libc.so.6",
           "/lib/x86_64-linux-gnu/libgcc_s.so.1", "/lib/x86_64-linux-gnu/libpthread.so.0"]

_state = {}


def available():
    return os.path.exists(ORB) and os.path.exists(G2O) and shutil.which("gcc") is not None


def _symbols(path, flag):
    """Dynamic symbols of one library; for --undefined-only the WEAK ones (__gmon_start__, _ITM_*: meant to stay null) are left out."""
    out = subprocess.check_output(["nm", "-D", flag, path], text=True)
    rows = [ln.split() for ln in out.splitlines() if ln.strip()]
    return {r[-1].split("@")[0] for r in rows if not (flag == "--undefined-only" and r[-2] in ("w", "v"))}


def _missing_sonames():
    out = subprocess.check_output(["ldd", ORB], text=True, env=dict(os.environ, LD_LIBRARY_PATH=""))
    return [ln.split()[0] for ln in out.splitlines() if "not found" in ln]


def load():
    """Returns (libg2o, libORB_SLAM2, sink) as ctypes handles, trap ARMED."""
    if _state:
        return _state["g2o"], _state["orb"], _state["sink"]
    tmp = tempfile.mkdtemp(prefix="ppo_orbslam_pin_")
    atexit.register(shutil.rmtree, tmp, True)
    need = sorted(_symbols(ORB, "--undefined-only") - set().union(*[_symbols(p, "--defined-only") for p in _SYSTEM + [G2O]]))
    src = ["#include <stdio.h>", "#include <stdlib.h>", "int sink_quiet = 1; long sink_quiet_hits = 0; long sink_armed_hits = 0;",
           "static long sink_hit(const char* name) {",
           "  if (sink_quiet) { ++sink_quiet_hits; return 0; }",
           "  ++sink_armed_hits; fprintf(stderr, \"orbslam_loader: TRAP -- a pinned call reached the absent symbol %s\\n\", name); abort(); }"]
    for i, s in enumerate(need):
        src.append('long sink_fn_%d(void) __asm__("%s"); long sink_fn_%d(void) { return sink_hit("%s"); }' % (i, s, i, s))
    with open(os.path.join(tmp, "sink.c"), "w") as fh:
        fh.write("\n".join(src) + "\n")
    sink_path = os.path.join(tmp, "libsink.so")
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-O0", "-o", sink_path, os.path.join(tmp, "sink.c")])
    open(os.path.join(tmp, "empty.c"), "w").close()
    sink = C.CDLL(sink_path, mode=C.RTLD_GLOBAL)
    g2o = C.CDLL(G2O, mode=C.RTLD_GLOBAL)
    keep = []
    for so in _missing_sonames():
        if so == "libg2o.so":
            continue  # the real one is loaded above under this soname
        p = os.path.join(tmp, so)
        subprocess.check_call(["gcc", "-shared", "-fPIC", "-Wl,-soname," + so, "-o", p, os.path.join(tmp, "empty.c")])
        keep.append(C.CDLL(p, mode=C.RTLD_GLOBAL))  # the linker then finds the soname already loaded
    orb = C.CDLL(ORB, mode=os.RTLD_LAZY | C.RTLD_GLOBAL)
    quiet = C.c_int.in_dll(sink, "sink_quiet")
    init_hits = C.c_long.in_dll(sink, "sink_quiet_hits").value
    quiet.value = 0  # ARM: from here on a call into an absent dependency aborts with its name

    def _disarm():  # static destructors of libORB_SLAM2.so run at exit and may touch cv:: / pcl:: objects
        quiet.value = 1
    atexit.register(_disarm)
    _state.update(g2o=g2o, orb=orb, sink=sink, keep=keep, n_trap_symbols=len(need), init_hits=init_hits, tmp=tmp)
    return g2o, orb, sink


def armed_hits():
    return C.c_long.in_dll(_state["sink"], "sink_armed_hits").value if _state else 0


def info():
    return {"trap_symbols": _state.get("n_trap_symbols"), "quiet_hits_during_dlopen": _state.get("init_hits"), "armed_hits": armed_hits()}


if __name__ == "__main__":
    load()
    print(info())
