import sys, time
sys.path.insert(0, "/root/repo/point-plane-object-slam_amd")
from ppo_ba import binding as B, synth
P = synth.make_problem("cfg4", seed=1)
for i in range(3):
    t0 = time.perf_counter()
    s = B.hip_backend().create(P, B.default_options(verbose=1 if i == 2 else 0))
    t1 = time.perf_counter()
    st = s.run()
    t2 = time.perf_counter()
    r = s.read()
    t3 = time.perf_counter()
    s.close()
    print("create %.1f ms  run %.1f ms  read %.1f ms  close %.1f ms" % (1e3*(t1-t0), 1e3*(t2-t1), 1e3*(t3-t2), 1e3*(time.perf_counter()-t3)))
