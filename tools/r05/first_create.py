"""development aid: phases of the FIRST ppo_ba_create of a process after ppo_ba_warmup (options.verbose prints them), BASELINE size by default"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "point-plane-object-slam_amd"))
from ppo_ba import binding as B, synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
P = synth.make_problem(cfg, seed=1)
be = B.hip_backend()
be.lib.ppo_ba_warmup.restype = C.c_int; be.lib.ppo_ba_warmup.argtypes = [C.c_int]
t0 = time.perf_counter(); assert be.lib.ppo_ba_warmup(0) == 0; t1 = time.perf_counter()
s = be.create(P, B.default_options(verbose=1)); t2 = time.perf_counter()
s.close()
s = be.create(P, B.default_options(verbose=0)); t3 = time.perf_counter()
s.close()
print("%s: warm-up %.1f ms, first create %.2f ms, second create %.2f ms" % (cfg, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2)))
