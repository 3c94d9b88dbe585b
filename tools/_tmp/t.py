import sys, time, ctypes as C
sys.path.insert(0,'point-plane-object-slam_amd')
import numpy as np
from ppo_ba import binding as B, pose
lib = B.load_library()
fr = pose.make_frames(n_frames=1, n_obs=400, seed=1)
fn = lib.ppo_pose_optimize
fn.argtypes = [C.POINTER(pose.CPoseProblem), C.POINTER(pose.CPoseOptions), C.POINTER(pose.CPoseResult)]
o = pose.default_options()
F, E = 1, fr.n_edges
ps = np.zeros((F, 7)); outl = np.zeros(E, np.uint8); ni = np.zeros(F, np.int32); chi2 = np.zeros(E); stats = (pose.CPoseStats * 1)()
r = pose.CPoseResult(); r.pose_qt = ps.ctypes.data_as(B.c_dp); r.outlier = outl.ctypes.data_as(B.c_bp); r.n_inliers = ni.ctypes.data_as(B.c_ip); r.chi2 = chi2.ctypes.data_as(B.c_dp); r.stats = stats
p = fr.as_struct()
for _ in range(5): fn(C.byref(p), C.byref(o), C.byref(r))
t0 = time.perf_counter()
for _ in range(200): fn(C.byref(p), C.byref(o), C.byref(r))
print("bare C call: %.1f us" % ((time.perf_counter() - t0) / 200 * 1e6))
t0 = time.perf_counter()
for _ in range(200): pose.optimize(lib, fr)
print("python wrapper: %.1f us" % ((time.perf_counter() - t0) / 200 * 1e6))
