#!/usr/bin/env python3
"""bench.py -- LM iterations/sec of the local-BA hot path on the BASELINE.json graph.

A *step* is one complete Optimizer::LocalBACameraPlaneCuboids-style call on the resident graph: reset of the
device estimates (device-to-device), 5 LM iterations, on-device outlier classification, 10 more LM iterations
(src/Optimizer.cc:2727-2838).  `value` = LM iterations executed / wall time, `ms_per_step` = wall ms per call;
the per-iteration time is in config.ms_per_lm_iter.  The graph is uploaded once before the timed region.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg4] [--no-cpu-baseline]
For N > 1 launch under torch.distributed.run (one rank per GPU); the landmark range is sharded over the ranks
and the reduced pose system is all-reduced with RCCL once per solve attempt (strong scaling of ONE graph, BASELINE config 5).
`--config cfg5w` is the weak-scaling companion: the same 200 keyframes with 50 000 points PER rank (400 000 on 8
GPUs); `value` stays LM iterations/s of that N-times larger graph, so perfect weak scaling is a constant `value`.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, "point-plane-object-slam_amd"))

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def cpu_baseline(P, B, seconds_budget=12.0):
    """The oracle (a port of the reference algorithm, 1 thread) on the same graph: the full 5+10 protocol, repeated
    until about `seconds_budget` seconds of CPU work have been timed (at least one complete call)."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import oracle_py  # test infrastructure: only this leg touches it
    be = oracle_py.backend()
    s = be.create(P, B.default_options())
    iters = trials = calls = 0
    dt = 0.0
    while calls == 0 or (dt < seconds_budget and dt / calls * (calls + 1) < 2.5 * seconds_budget):
        s.reset()
        t0 = time.perf_counter()
        st = s.run()
        dt += time.perf_counter() - t0
        iters += st["n_iters"]
        trials += st["n_trials"]
        calls += 1
    s.close()
    return dict(value=iters / dt, unit="LM iters/s", cores=1, kind="port", ms_per_call=1e3 * dt / calls,
                sample="oracle/libppo_oracle.so (g2o algorithm restated, fixed-size blocks, -O3 -march=native, 1 thread): %d complete 5+10 calls "
                       "on the same graph = %d LM iterations (%d solve attempts) in %.1f s; host: %s, %d logical CPUs" % (
                           calls, iters, trials, dt, cpu_model(), os.cpu_count() or 0))


FP64_PEAK_TFLOPS = 78.6  # MI355X FP64 (vector = matrix) peak: half the 157.3 TF FP32 rate of /opt/skills/guides/MI355X_MICROARCH.md


def cpu_baseline_openmp(P, B, seconds_budget=6.0):
    """Secondary, clearly labelled "best CPU" line: the oracle built with OpenMP -- edge loop of the linearisation with
    per-thread accumulators, landmark loop of the Schur complement with per-thread copies of S, a blocked right-looking Cholesky
    (64-wide panels, two fork-joins per panel) instead of the unblocked pivoted LDLT, residual pass over the edges.  NOT how the
    reference runs -- g2o's OpenMP is off (Thirdparty/g2o/config.h:4) and the whole BA sits on the LocalMapping thread -- and not
    part of cpu_baseline."""
    path = os.path.join(REPO, "oracle", "libppo_oracle_omp.so")
    if not os.path.exists(path):
        return None
    # (a container's CPU quota, not the number of logical CPUs it can see, is what the threads share: 32 spinning OpenMP threads
    #  under a 16-CPU quota ran this oracle seven times SLOWER than one thread)
    quota = os.cpu_count() or 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = max(1, int(int(q) / int(per)))
    except Exception:
        pass
    threads = max(1, min(8, os.cpu_count() or 1, quota))
    os.environ.setdefault("OMP_NUM_THREADS", str(threads))
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    lib = C.CDLL(path)
    # (torch loaded libgomp long ago, with one thread per VISIBLE CPU -- 256 on the GPU box, whose container may use 16: the environment
    #  variable is too late, the count is set through the library)
    threads = int(lib.oracle_set_threads(threads))
    be = B._Backend(lib, "oracle_ba_")
    s = be.create(P, B.default_options())
    iters = calls = 0
    dt = 0.0
    while calls == 0 or (dt < seconds_budget and dt / calls * (calls + 1) < 2.0 * seconds_budget):
        s.reset()
        t0 = time.perf_counter()
        st = s.run()
        dt += time.perf_counter() - t0
        iters += st["n_iters"]
        calls += 1
    s.close()
    return dict(value=iters / dt, unit="LM iters/s", cores=threads, kind="port+openmp", ms_per_call=1e3 * dt / calls,
                sample="oracle/libppo_oracle_omp.so: %d complete 5+10 calls = %d LM iterations in %.1f s; OpenMP over the edges (linearisation, residuals), "
                       "the landmarks (Schur complement) and the tiles of a blocked Cholesky; same schedule and results to rounding as the 1-thread "
                       "port (tests/test_oracle_known_answers.py); the reference itself is single-threaded" % (calls, iters, dt))


def committed_traffic(k_bytes):
    """HBM bytes per launch of the roofline kernel from the committed PMC passes (profiles/rNN_lin_traffic.json, made by
    tools/pmc_traffic.sh on this exact workload: FETCH_SIZE / WRITE_SIZE in separate --pmc runs), and the file they came from;
    (None, None) when absent or made for a different graph.  The counters are not collected inside this run.  The NEWEST round's file
    is tried first; a newer file that cannot be used (another graph, a schema without the two fields) is named on stderr."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r[0-9][0-9]_lin_traffic.json")),
                   key=lambda f: int(re.search(r"r(\d+)_", os.path.basename(f)).group(1)), reverse=True)
    for f in files:
        name = os.path.basename(f)
        try:
            j = json.load(open(f))
            if "hbm_bytes_per_launch" not in j or "algorithmic_bytes" not in j:
                raise KeyError("no algorithmic_bytes / hbm_bytes_per_launch on top level")
            if int(j["algorithmic_bytes"]) != int(k_bytes):
                raise ValueError("made for %d algorithmic bytes, this workload has %d" % (int(j["algorithmic_bytes"]), int(k_bytes)))
            return float(j["hbm_bytes_per_launch"]), "profiles/" + name
        except Exception as e:  # noqa: BLE001
            sys.stderr.write("bench.py: profiles/%s skipped for roofline.traffic: %s\n" % (name, e))
    return None, None


def pose_only_figures(B, with_cpu):
    """Secondary figures (never part of `value`): pose-only optimisation (include/ppo_pose.h, Optimizer::PoseOptimization) of
    synthetic frames, whole call with host buffers (pack + PCIe + one kernel + read-back); the oracle timed beside it."""
    from ppo_ba import pose
    lib = B.load_library()
    out = {}
    for label, fr in (("single_frame_500_obs", pose.make_frames(n_frames=1, n_obs=500, seed=21)),
                      ("batch_256_frames_500_obs", pose.make_frames(n_frames=256, n_obs=500, seed=22))):
        pose.optimize(lib, fr)
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps):
            pose.optimize(lib, fr)
        ms = 1e3 * (time.perf_counter() - t0) / reps
        out[label] = {"ms_per_call": ms, "frames_per_s": fr.n_frames / (ms * 1e-3)}
        if with_cpu:
            sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle"))
            import oracle_py
            small = fr if fr.n_frames == 1 else pose.make_frames(n_frames=16, n_obs=500, seed=22)
            t0 = time.perf_counter()
            n = 0
            while time.perf_counter() - t0 < 0.5:
                pose.optimize(oracle_py.load(), small, prefix="oracle_")
                n += small.n_frames
            out[label]["cpu_oracle_frames_per_s"] = n / (time.perf_counter() - t0)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="cfg4")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--solver", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--with-corners", action="store_true", help="add the 16-D cuboid corner edges (off in the BASELINE configuration)")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary figures (pose-only, concurrent windows, phase times): "
                    "used for the rocprofv3 runs so that the kernel statistics hold the headline workload only")
    ap.add_argument("--lin-reps", type=int, default=200)
    ap.add_argument("--lm-control", type=int, default=0, help="0 (default): the LM decisions by a kernel behind every trial (options.lm_control = PPO_LM_DEVICE), "
                    "1: by the host after a round trip per trial.  Sharded runs always take the host path")
    ap.add_argument("--deterministic", type=int, default=1, help="options.deterministic for the timed handle (1 = the library's default: bit-reproducible integer sums)")
    ap.add_argument("--solve-on-owner", type=int, default=0, help="N > 1: 1 = reduce (S, b) to rank 0, solve there, broadcast x (options.solve_on_owner)")
    a = ap.parse_args()

    import numpy as np
    import torch
    from ppo_ba import binding as B
    from ppo_ba import synth

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world != a.gpus:
        sys.exit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d bench.py --gpus %d ..." % (a.gpus, a.gpus))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: libppo_ba has no CPU path")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    lib = B.load_library()
    P = synth.make_problem(a.config, seed=a.seed, with_corners=a.with_corners, world_size=world)  # (cfg5w: 50 000 points per rank)
    weak = isinstance(synth.CONFIGS.get(a.config), dict) and synth.CONFIGS[a.config].get("points_per_rank", False)
    sizes = P.sizes()
    opt = B.default_options(device_id=local_rank, rank=rank, world_size=world, solver=a.solver, solve_on_owner=a.solve_on_owner if world > 1 else 0, lm_control=a.lm_control, deterministic=a.deterministic)
    if world > 1:
        ident = np.zeros(128, np.uint8)
        if rank == 0:
            assert lib.ppo_ba_comm_unique_id(ident.ctypes.data_as(B.c_bp)) == 0
        t = torch.from_numpy(ident).cuda()
        dist.broadcast(t, 0)
        ident = t.cpu().numpy()
        for i in range(128):
            opt.comm_id[i] = int(ident[i])
    # what System's constructor would do once: code objects, stream, mapped block, host team and block cache up front ...
    lib.ppo_ba_warmup.restype = C.c_int
    lib.ppo_ba_warmup.argtypes = [C.c_int]
    t_w = time.perf_counter()
    rc_w = lib.ppo_ba_warmup(local_rank)
    warmup_ms = 1e3 * (time.perf_counter() - t_w)
    if rc_w != 0:
        sys.exit("ppo_ba_warmup failed: %d" % rc_w)
    # ... so that the first create of the process costs what every later one does
    t_c = time.perf_counter()
    solver = B.hip_backend().create(P, opt)
    create_ms = 1e3 * (time.perf_counter() - t_c)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        solver.reset()
        return solver.run()

    for _ in range(a.warmup):
        st = step()
    barrier()
    t0 = time.perf_counter()
    iters = trials = 0
    for _ in range(a.steps):
        st = step()
        iters += st["n_iters"]
        trials += st["n_trials"]
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # Which mode the parity claim belongs to (single rank; outside the timed region).  Round 6: `value` above IS the mode the strict comparisons of
    # tests/ run (against the oracle at full size, the random windows, the reference's own runs): options.deterministic = 1, the library's default.
    # Next to it the OTHER setting of the switch (floating-point atomics, opt-in): its rate on this workload, and how far its run ends from the
    # timed one per quantity of BASELINE.md's bar.
    deterministic_mode = None
    other_mode = None
    if world == 1 and not a.no_extras:
        r_def = solver.read()
        sd_ = B.hip_backend().create(P, B.default_options(device_id=local_rank, solver=a.solver, deterministic=0 if a.deterministic else 1))
        for _ in range(2):
            sd_.reset(); sd_.run()
        torch.cuda.synchronize()
        t_d = time.perf_counter()
        n_d = 0
        for _ in range(max(5, a.steps // 2)):
            sd_.reset()
            n_d += sd_.run()["n_iters"]
        dt_d = time.perf_counter() - t_d
        r_det = sd_.read()
        sd_.close()

        def _qangle(qa, qb):
            d = np.minimum(np.linalg.norm(qa - qb, axis=1), np.linalg.norm(qa + qb, axis=1))
            return float((2 * d).max()) if len(d) else 0.0
        dev = {"pose_translation_rmse_m": float(np.sqrt(np.mean(np.sum((r_def.pose_qt[:, 4:] - r_det.pose_qt[:, 4:]) ** 2, axis=1)))),
               "pose_rotation_max_rad": _qangle(r_def.pose_qt[:, :4], r_det.pose_qt[:, :4]),
               "point_rmse_m": float(np.sqrt(np.mean(np.sum((r_def.point_xyz - r_det.point_xyz) ** 2, axis=1)))) if r_def.point_xyz.size else 0.0,
               "plane_coef_max_abs": float(np.abs(r_def.plane_coef - r_det.plane_coef).max()) if r_def.plane_coef.size else 0.0,
               "cuboid_9_max_abs": float(max(np.abs(r_def.cuboid_qt - r_det.cuboid_qt).max(), np.abs(r_def.cuboid_scale - r_det.cuboid_scale).max())) if r_def.cuboid_qt.size else 0.0}
        bars = {"pose_translation_rmse_m": 1e-5, "pose_rotation_max_rad": 1e-6, "point_rmse_m": 1e-5, "plane_coef_max_abs": 1e-6, "cuboid_9_max_abs": 1e-5}
        other_mode = {"deterministic": 0 if a.deterministic else 1, "lm_iters_per_s": n_d / dt_d, "ms_per_lm_iter": 1e3 * dt_d / max(n_d, 1),
                      "its_run_deviation_from_the_timed_run": dev, "baseline_bars": bars}
        if a.deterministic:
            deterministic_mode = {"is_the_timed_mode": True, "lm_iters_per_s": iters / dt, "ms_per_lm_iter": 1e3 * dt / max(iters, 1),
                                  "floating_point_atomics_mode": other_mode}
        else:
            deterministic_mode = {"is_the_timed_mode": False, "lm_iters_per_s": n_d / dt_d, "ms_per_lm_iter": 1e3 * dt_d / max(n_d, 1),
                                  "default_run_deviation_from_it": dev, "baseline_bars": bars}

    # secondary figure: the same workload with the OTHER setting of options.lm_control (single rank; outside the timed region)
    lm_control = {"in_use": "host" if (a.lm_control or world > 1 or a.solver == 1) else "device"}
    if world == 1 and not a.no_extras:
        so_ = B.hip_backend().create(P, B.default_options(device_id=local_rank, solver=a.solver, lm_control=0 if a.lm_control else 1))
        for _ in range(2):
            so_.reset(); so_.run()
        torch.cuda.synchronize()
        t_o = time.perf_counter()
        n_o = 0
        for _ in range(max(5, a.steps // 2)):
            so_.reset()
            n_o += so_.run()["n_iters"]
        dt_o = time.perf_counter() - t_o
        so_.close()
        lm_control["other"] = {"lm_control": "device" if a.lm_control else "host", "lm_iters_per_s": n_o / dt_o}

    # roofline of the Jacobian/assembly kernel -- k_linearize_all, the ONE fused launch that evaluates residuals and
    # Jacobians of every edge class and writes the block-sparse system (SURVEY.md 8(d) "linearise + assemble") -- timed
    # alone with HIP events on the library's stream; with its zeroing kernel in front it is the "linearise group".
    lin_ms = C.c_double()
    solver.reset()
    assert lib.ppo_ba_time_linearize(solver.h, a.lin_reps, C.byref(lin_ms)) == 0
    lin_bytes = int(lib.ppo_ba_linearize_bytes(solver.h))
    k_ms, k_bytes = C.c_double(), C.c_int64()
    assert lib.ppo_ba_time_kernel(solver.h, 6, a.lin_reps, C.byref(k_ms), C.byref(k_bytes)) == 0
    kernels = {}
    n_red = C.c_int32()
    lib.ppo_ba_debug_reduced(solver.h, None, None, None, C.byref(n_red))
    b_red = np.zeros(max(1, n_red.value))
    assert lib.ppo_ba_debug_reduced(solver.h, None, b_red.ctypes.data_as(B.c_dp), None, None) == 0  # builds W, Z, C for the Schur kernels
    for name, which in (("k_linearize_all", 6), ("k_lin_poses", 0), ("k_lin_points", 1), ("k_schur_pairs", 3), ("k_backsub", 4), ("k_chi2_points", 5)):
        ms_, by_ = C.c_double(), C.c_int64()
        if lib.ppo_ba_time_kernel(solver.h, which, 50, C.byref(ms_), C.byref(by_)) == 0 and ms_.value > 0:
            kernels[name] = {"us": 1e3 * ms_.value, "algorithmic_bytes": int(by_.value), "GBps": by_.value / (ms_.value * 1e-3) / 1e9}
    fac_ms = C.c_double()
    assert lib.ppo_ba_time_factor(solver.h, 10, C.byref(fac_ms)) == 0
    fin = (C.c_int32 * 260)()
    n_panels = int(lib.ppo_ba_overlap_schedule(solver.h, fin, 260))
    reduction = {"overlapped_with_factorisation": n_panels > 0}
    if n_panels > 0:
        sched = [int(fin[k]) for k in range(n_panels + 1)]
        reduction.update({"panel_launches": n_panels, "column_blocks_final_at_launch": sched[:next((k for k, f in enumerate(sched) if f >= n_panels), n_panels) + 1],
                          "note": "pose-pair chunks ordered by 64-column block of S; only the first block(s) are reduced ahead of the dense factorisation, the "
                                  "panel launches carry the rest as extra workgroups (csrc/ppo_chol.hip StepOverlap); factor_solve_ms is the dense solve alone"})
    achieved = k_bytes.value / (k_ms.value * 1e-3) / 1e9 if k_ms.value > 0 else 0.0
    traffic, traffic_source = committed_traffic(int(k_bytes.value))
    ld_dense = C.c_int32()
    lib.ppo_ba_debug_reduced(solver.h, None, None, None, C.byref(ld_dense))  # dimension of the system the dense solve factors

    # N > 1: what one solve attempt spends in its exchange steps, each timed alone with HIP events (outside the timed region), the other
    # variant of the (S, b) exchange beside the one in use, and the complete call in that other variant -- so that ONE lease of a multi-GPU
    # node answers "which exchange, and what share of an iteration is it"
    multi = {}
    if world > 1:
        coll = np.zeros(4)
        assert lib.ppo_ba_time_collectives(solver.h, 20, coll.ctypes.data_as(B.c_dp)) == 0
        solver.reset()
        multi = {"us_per_solve_attempt": {"sb_allreduce_incl_pack_unpack": 1e3 * coll[0], "sb_reduce_to_rank0_plus_bcast_x": 1e3 * coll[1],
                                          "scalar_allreduce": 1e3 * coll[2], "dense_solve_every_rank": 1e3 * fac_ms.value},
                 "packed_sb_bytes": int(coll[3]), "exchange_in_use": "reduce+bcast" if a.solve_on_owner else "allreduce",
                 "note": "strong scaling of ONE window: linearisation, Schur terms, back-substitution shard; the dense solve (factor_solve_ms) does not"}
        other = B.default_options(device_id=local_rank, rank=rank, world_size=world, solver=a.solver, solve_on_owner=0 if a.solve_on_owner else 1)
        for i in range(128):
            other.comm_id[i] = opt.comm_id[i]
        so = B.hip_backend().create(P, other)
        so.run()
        barrier()
        t_o = time.perf_counter()
        n_o = 0
        for _ in range(max(3, a.steps // 4)):
            so.reset()
            n_o += so.run()["n_iters"]
        barrier()
        dt_o = time.perf_counter() - t_o
        if dist is not None:
            tt = torch.tensor([dt_o], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt_o = float(tt.item())
        so.close()
        multi["other_exchange"] = {"exchange": "allreduce" if a.solve_on_owner else "reduce+bcast", "lm_iters_per_s": n_o / dt_o}

    # what a per-call user pays around the LM iterations (host flattening + PCIe upload, read-back): steady state, i.e. with
    # the library's block cache warm as in a running SLAM process; never part of `value`
    call_ms = {}
    if world == 1:
        samples = []
        for _ in range(5):  # (median of five: a single call is at the mercy of the container's CPU quota)
            solver.close()
            t_a = time.perf_counter()
            solver = B.hip_backend().create(P, opt)
            t_b = time.perf_counter()
            solver.run()
            t_c2 = time.perf_counter()
            solver.read()
            t_d = time.perf_counter()
            samples.append((1e3 * (t_b - t_a), 1e3 * (t_c2 - t_b), 1e3 * (t_d - t_c2), 1e3 * (t_d - t_a)))
        med = [sorted(x[i] for x in samples)[len(samples) // 2] for i in range(4)]
        call_ms = {"create": med[0], "run": med[1], "read": med[2], "total": med[3],
                   "note": "median of 5 calls through the Python harness (read includes allocating the result arrays); examples/replay.cpp measures the C++ caller"}

    extras = world == 1 and not a.no_extras
    pose_only = pose_only_figures(B, not a.no_cpu_baseline) if extras else {}
    # secondary figure, never `value`: two independent windows of the same size driven by two host threads on this GPU
    # (distinct handles are thread-compatible); a single window is latency-bound, so two of them overlap
    concurrent = {}
    if extras:
        import threading
        hs = [solver, B.hip_backend().create(P, opt)]
        hs[1].run()
        counts = [0, 0]

        def work(i):
            for _ in range(10):
                hs[i].reset()
                counts[i] += len(hs[i].run()["iters"])
        th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
        t_c0 = time.perf_counter()
        for t_ in th:
            t_.start()
        for t_ in th:
            t_.join()
        concurrent = {"handles": 2, "lm_iters_per_s_aggregate": sum(counts) / (time.perf_counter() - t_c0)}
        hs[1].close()
    # secondary figure, never `value`: the same window with options.jacobian_mode = PPO_JAC_ANALYTIC (exact derivatives for the plane and
    # cuboid-projection edges instead of the reference's delta = 1e-9 central differences; opt-in, include/ppo_ba.h)
    analytic = {}
    if extras and (sizes["n_plane_edges"] or sizes["n_bbox"]):
        ao = B.default_options(device_id=local_rank, solver=a.solver, jacobian_mode=1)
        sa = B.hip_backend().create(P, ao)
        sa.run()
        n_it = 0
        t_a0 = time.perf_counter()
        for _ in range(10):
            sa.reset()
            n_it += sa.run()["n_iters"]
        t_a1 = time.perf_counter()
        sa.reset()
        ka_ms, ka_b = C.c_double(), C.c_int64()
        assert lib.ppo_ba_time_kernel(sa.h, 6, a.lin_reps, C.byref(ka_ms), C.byref(ka_b)) == 0
        analytic = {"lm_iters_per_s": n_it / (t_a1 - t_a0), "k_linearize_all_us": 1e3 * ka_ms.value,
                    "note": "opt-in mode; parity against the numeric oracle: tests/test_parity_gpu.py::test_analytic_jacobians_*"}
        sa.close()
    # secondary figure, never `value`: the reference's REAL operating point (ICL-sized windows, 157.48 ms per call
    # published for its CPU path) -- what one call costs all inclusive, a fresh handle per call as in a running SLAM process
    icl = {}
    if extras:
        for label, cfg in (("15kf_800pt_4pl_2cub", dict(n_kf=15, n_points=800, n_planes=4, n_cuboids=2)),
                           ("30kf_2000pt_10pl_3cub", dict(n_kf=30, n_points=2000, n_planes=10, n_cuboids=3))):
            Pw = synth.make_problem(cfg, seed=1)
            ow = B.default_options(device_id=local_rank, solver=a.solver)
            rows = []
            for i in range(13):
                t_0 = time.perf_counter()
                sw = B.hip_backend().create(Pw, ow)
                t_1 = time.perf_counter()
                stw = sw.run()
                t_2 = time.perf_counter()
                sw.read()
                t_3 = time.perf_counter()
                sw.close()
                if i >= 3:
                    rows.append((t_1 - t_0, t_2 - t_1, t_3 - t_2, t_3 - t_0, 1e-3 * sw.c_ms["create"], 1e-3 * sw.c_ms["run"], 1e-3 * sw.c_ms["read"]))
            med = [1e3 * sorted(r[k] for r in rows)[len(rows) // 2] for k in range(7)]
            # create / run / read: INSIDE the C calls (what the adapter in Optimizer.cc pays; examples/replay.cpp measures the same from C++:
            # profiles/r06_icl_from_cpp.txt); *_with_harness: the same calls with this script's ctypes marshalling of ~60 arrays around them
            icl[label] = {"create_ms": med[4], "run_ms": med[5], "read_ms": med[6], "call_ms": med[4] + med[5] + med[6],
                          "create_ms_with_harness": med[0], "read_ms_with_harness": med[2], "call_ms_with_harness": med[3], "lm_iters": stw["n_iters"],
                          "solve_attempts": stw["n_trials"],  # (> lm_iters: rejected trials, each a complete reduce + solve + evaluate)
                          "us_per_lm_iter": 1e3 * med[5] / max(stw["n_iters"], 1), "us_per_solve_attempt": 1e3 * med[5] / max(stw["n_trials"], 1)}
            # ... and repeated runs on ONE handle (reset + run), with the LM decisions on the device (default) and on the host
            for key, lmc in (("run_ms_repeated_lm_on_device", 0), ("run_ms_repeated_lm_on_host", 1)):
                sw = B.hip_backend().create(Pw, B.default_options(device_id=local_rank, solver=a.solver, lm_control=lmc))
                for _ in range(3):
                    sw.reset(); sw.run()
                t_0 = time.perf_counter()
                for _ in range(20):
                    sw.reset(); sw.run()
                icl[label][key] = 1e3 * (time.perf_counter() - t_0) / 20
                sw.close()
        icl["note"] = "median of 10 calls, a fresh handle per call (create + run + read); the reference publishes 157.48 ms per call for its CPU path on ICL-NUIM"
    # where the device time of one 5+10 call goes (ppo_ba_stats.ms_*, HIP events on the stream, outside the timed region)
    phases = {}
    if extras:
        vo = B.default_options(solver=a.solver, verbose=1)
        err = os.dup(2)
        devnull = os.open(os.devnull, os.O_WRONLY)
        os.dup2(devnull, 2)  # verbose also prints the phase times of ppo_ba_create on stderr
        try:
            vs = B.hip_backend().create(P, vo)
            vs.run()
            vs.reset()
            tv = vs.run()
            vs.close()
        finally:
            os.dup2(err, 2)
            os.close(devnull)
            os.close(err)
        phases = {k: tv[k] for k in ("ms_linearize", "ms_schur", "ms_factor", "ms_backsub", "ms_chi2")}
        phases["wall_ms_total"] = tv["wall_ms_total"]

    if rank == 0:
        out = {
            "metric": "LM iters/sec on 200KF/50k-pt/200-plane/50-cuboid local BA",
            "value": iters / dt, "unit": "LM iters/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "weak" if weak else "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": "%s seed %d: %d KF / %d points / %d planes / %d cuboids; %d mono + %d plane + %d bbox + %d point-cuboid + %d cuboid-plane edges; "
                            "reduced system %d in the reference's partition (poses + cuboids); dense system actually factored: %d (cuboids eliminated "
                            "too, leading dimension %d); step = one 5+10 LM call (reset + run)" % (
                                a.config, a.seed, sizes["n_poses"], sizes["n_points"], sizes["n_planes"], sizes["n_cuboids"], sizes["n_mono"],
                                sizes["n_plane_edges"], sizes["n_bbox"], sizes["n_ptcub"], sizes["n_cubpl"], st["n_reduced"], ld_dense.value,
                                (ld_dense.value + 63) // 64 * 64),
                "dense_dim": int(ld_dense.value), "dense_ld": int((ld_dense.value + 63) // 64 * 64), "reference_reduced_dim": int(st["n_reduced"]),
                "lm_iters_per_step": iters / a.steps, "solve_attempts_per_step": trials / a.steps,
                "ms_per_lm_iter": 1e3 * dt / max(iters, 1), "solver_backend": int(a.solver),
                "factor_solve_ms": fac_ms.value, "schur_reduction": reduction, "parallelism": "landmark-range x%d" % world,
                "deterministic_mode": deterministic_mode,
                "parity_mode_note": ("value is the library's default mode, options.deterministic = 1 (bit-reproducible integer sums, overlapped reduction, LM decisions on "
                                     "the device): the mode the strict parity tests run (oracle at full size, random windows, the reference's own runs: "
                                     "tests/test_parity_gpu.py, tests/test_orbslam_lm_pin.py, tests/test_orbslam_63_pin.py).  config.deterministic_mode."
                                     "floating_point_atomics_mode is the opt-in options.deterministic = 0, held to the deterministic result by "
                                     "test_atomics_mode_stays_near_the_deterministic_result") if a.deterministic else
                                    "value was measured with --deterministic 0 (floating-point atomics, opt-in): NOT the mode the strict parity tests run",
                "warmup_ms_incl_module_load": warmup_ms, "first_create_ms_after_warmup": create_ms, "steady_state_call_ms": call_ms, "kernels": kernels,
                "pose_only": pose_only, "device_ms_per_phase_of_one_call": phases,
                "two_windows_concurrently": concurrent, "analytic_jacobians": analytic, "multi_gpu": multi, "icl_sized_windows": icl, "lm_control": lm_control,
            },
            "roofline": {"bound": "hbm", "kernel": "k_linearize_all", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "algorithmic_bytes": int(k_bytes.value), "launch_ms": k_ms.value,
                         "linearise_group": {"kernels": "k_zero_ranges + k_linearize_all",
                                             "algorithmic_bytes": lin_bytes, "launch_ms": lin_ms.value,
                                             "achieved": lin_bytes / (lin_ms.value * 1e-3) / 1e9}},
        }
        # the kernel group that owns the clock: the dense solve of the reduced pose system (chol_step x panels + chol_back_all), FP64 on the
        # MFMA / vector units.  flops = n^3 / 3 + 2 n^2 at the dimension actually factored; launch_ms = the solve alone after a complete
        # reduction (ppo_ba_time_factor).  A latency chain, not a throughput kernel: the fraction says how far from a GEMM it is.
        n_d = float(ld_dense.value)
        dense_flops = n_d ** 3 / 3.0 + 2.0 * n_d ** 2
        dense_tf = dense_flops / (fac_ms.value * 1e-3) / 1e12 if fac_ms.value > 0 else 0.0
        out["roofline_dense"] = {"bound": "mfma", "kernel": "chol_step x %d + chol_back_all" % ((ld_dense.value + 63) // 64), "dtype": "f64",
                                 "flops": dense_flops, "launch_ms": fac_ms.value, "achieved": dense_tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                 "frac": dense_tf / FP64_PEAK_TFLOPS, "share_of_lm_iteration": fac_ms.value / (1e3 * dt / max(trials, 1)),
                                 "note": "the panel chain is issue- and latency-bound (one workgroup's 16x16 register factorisations per panel, DESIGN.md 4.1), "
                                         "not FLOP-bound: share_of_lm_iteration is the figure to lower"}
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(P, B)
            omp = cpu_baseline_openmp(P, B)
            if omp and omp["value"] <= out["cpu_baseline"]["value"]:
                omp = {"omitted": "the OpenMP build of the oracle ran slower than the single-threaded one on this host (%.2f vs %.2f LM iters/s, %d threads)"
                                  % (omp["value"], out["cpu_baseline"]["value"], omp["cores"])}
            if omp:
                out["config"]["cpu_secondary_openmp"] = omp
        print(json.dumps(out))
    solver.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
