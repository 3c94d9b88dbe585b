"""Two (or more) REAL RCCL ranks, one process per GPU: self-skips on a box with fewer than two GPUs (the pool's test boxes have one;
the sharded code path itself runs on one GPU through the loopback communicator, tests/test_parity_gpu.py)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _gpus():
    # (counted in a child process: importing torch here would load its bundled HIP runtime and RCCL into the pytest process, next
    #  to ROCm's that libppo_ba.so is bound to -- the one-rank RCCL self-test of test_parity_gpu.py then meets the wrong pair)
    try:
        out = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count() if torch.cuda.is_available() else 0)"],
                             capture_output=True, text=True, timeout=300)
        return int(out.stdout.strip().splitlines()[-1])
    except Exception:  # noqa: BLE001
        return 0


def _launch(n, *args):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(29517 + n), os.path.join(HERE, "mgpu_worker.py"), *args]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("MGPU ")][-1]
    return json.loads(line[5:])


@pytest.mark.parametrize("on_owner", [0, 1])
@pytest.mark.parametrize("cfg", ["small", "cfg3"])
def test_two_rccl_ranks_give_the_single_rank_answer(cfg, on_owner):
    if _gpus() < 2:
        pytest.skip("needs two GPUs")
    out = _launch(2, cfg, "-1", str(on_owner))
    r0 = out[0]
    assert all(o["rc"] == 0 for o in out)
    assert all(o["iters"] == r0["iters"] for o in out)  # every rank walks the same schedule
    assert r0["same_schedule"] and r0["levels_equal"]
    assert r0["dchi2"] < 1e-6 and r0["t_rmse"] < 1e-5 and r0["p_rmse"] < 1e-5, r0


def test_stop_flag_on_one_rccl_rank_aborts_all_of_them():
    if _gpus() < 2:
        pytest.skip("needs two GPUs")
    out = _launch(2, "small", "1")
    assert all(o["rc"] == 1 for o in out), out  # PPO_BA_ABORTED everywhere, nobody left in a collective
    assert all(o["iters"] == out[0]["iters"] for o in out)
