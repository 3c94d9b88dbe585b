"""bench.py's N > 1 branch on CPU: torch.distributed.run with two ranks, gloo, the library stubbed (tests/bench_plumbing_worker.py).
The first contact of that branch with an 8-GPU node must not also be its first execution: argument check, process group, id broadcast,
per-rank options, barrier + MAX-reduced time, rank-0-only JSON, `scaling`."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
WORKER = os.path.join(HERE, "bench_plumbing_worker.py")


def _launch(n, port, log, *args):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", str(port),
           WORKER, "--gpus", str(n), "--steps", "4", "--warmup", "1", "--no-cpu-baseline", *args]
    env = dict(os.environ, PPO_PLUMBING_LOG=str(log), OMP_NUM_THREADS="1")
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)


@pytest.mark.parametrize("cfg,scaling", [("cfg4", "strong"), ("cfg5w", "weak")])
def test_two_rank_bench_plumbing(tmp_path, cfg, scaling):
    log = tmp_path / "plumb"
    res = _launch(2, 29631 if cfg == "cfg4" else 29633, log, "--config", cfg)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout  # rank 0 only, ONE line
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 4 and j["warmup"] == 1 and j["scaling"] == scaling and j["higher_is_better"] is True
    assert j["unit"] == "LM iters/s" and j["dtype"] == "f64" and j["vs_baseline"] is None and "cpu_baseline" not in j
    assert j["config"]["parallelism"] == "landmark-range x2" and "roofline" in j
    mg = j["config"]["multi_gpu"]  # the exchange steps of one solve attempt, and the other exchange variant beside the one in use
    assert mg["exchange_in_use"] == "allreduce" and mg["other_exchange"]["exchange"] == "reduce+bcast" and mg["other_exchange"]["lm_iters_per_s"] > 0
    assert set(mg["us_per_solve_attempt"]) == {"sb_allreduce_incl_pack_unpack", "sb_reduce_to_rank0_plus_bcast_x", "scalar_allreduce", "dense_solve_every_rank"}
    # the stub's rank 1 takes 30 ms per step, rank 0 10 ms: the reported time is the MAX over the ranks (both wait at the barrier)
    assert j["ms_per_step"] >= 29.0, j["ms_per_step"]
    assert abs(j["value"] - 15 * 4 / (j["ms_per_step"] * 4e-3)) <= 1e-6 * j["value"]  # whole-job LM iterations / that time
    recs = [json.load(open(str(log) + ".rank%d" % r)) for r in range(2)]
    want = bytes((7 * i + 3) % 251 for i in range(128)).hex()
    for r, rec in enumerate(recs):
        assert (rec["rank"], rec["world_size"], rec["device_id"]) == (r, 2, r)
        assert rec["comm_id"] == want  # rank 0's ncclUniqueId reached every rank's options
    if cfg == "cfg5w":  # 50 000 points PER rank in the real configuration: the generator is told the world size
        assert recs[0]["n_points"] == recs[1]["n_points"] == 2 * 120


def test_gpus_flag_must_match_the_launcher():
    env = dict(os.environ, PPO_PLUMBING_LOG="/tmp/unused")
    env.pop("WORLD_SIZE", None)
    res = subprocess.run([sys.executable, WORKER, "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode != 0 and "torch.distributed.run" in res.stderr
