"""Worker of tests/test_bench_plumbing.py: runs bench.py's main() on CPU ranks under torch.distributed.run with everything that needs a
GPU replaced by stand-ins -- gloo instead of RCCL, no-op device calls, and a stub of the library whose solver "runs" for a
rank-dependent time.  What executes for real is bench.py's own rank plumbing: the --gpus / WORLD_SIZE check, the process group, the
ppo_ba_comm_unique_id broadcast into options.comm_id, the per-rank options (device_id / rank / world_size), the barrier-bracketed
timed region, the MAX-reduction of the elapsed time, rank-0-only printing, `scaling`.  TEST AID: nothing here is a product path."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "point-plane-object-slam_amd"))
from ppo_ba import binding as B  # noqa: E402
from ppo_ba import synth  # noqa: E402

RANK = int(os.environ.get("RANK", "0"))
STEP_S = 0.010 * (1 + 2 * RANK)  # rank r "computes" for 10 ms * (1 + 2 r) per step: the slowest rank must set the reported time
ID = bytes((7 * i + 3) % 251 for i in range(128))
LOG = os.environ["PPO_PLUMBING_LOG"] + ".rank%d" % RANK


def _set(ref, v):
    ref._obj.value = v


class StubLib:
    """The entry points bench.py calls on the library object (ctypes signatures: byref outputs)."""

    def ppo_ba_comm_unique_id(self, out):
        assert RANK == 0, "only rank 0 may ask for the id"
        C.memmove(out, ID, 128)
        return 0

    class _Fn:  # (bench.py sets restype / argtypes on the entry point before calling it)
        def __call__(self, device):
            return 0

    ppo_ba_warmup = _Fn()

    def ppo_ba_time_linearize(self, h, reps, ms):
        _set(ms, 0.03)
        return 0

    def ppo_ba_linearize_bytes(self, h):
        return 50_000_000

    def ppo_ba_time_kernel(self, h, which, reps, ms, by):
        _set(ms, 0.028)
        _set(by, 50_000_000)
        return 0

    def ppo_ba_debug_reduced(self, h, S, b, x, n):
        if n is not None:
            _set(n, 1194)
        return 0

    def ppo_ba_time_factor(self, h, reps, ms):
        _set(ms, 0.28)
        return 0

    def ppo_ba_overlap_schedule(self, h, fin, n):
        return 0

    def ppo_ba_time_collectives(self, h, reps, out):
        for i, v in enumerate((0.09, 0.05, 0.02, 5.9e6)):
            out[i] = v
        return 0


class StubSolver:
    h = None

    def __init__(self, problem, opt):
        self.sizes = problem.sizes()
        rec = {"rank": int(opt.rank), "world_size": int(opt.world_size), "device_id": int(opt.device_id), "comm_id": bytes(opt.comm_id[i] for i in range(128)).hex(),
               "n_points": int(self.sizes["n_points"])}
        json.dump(rec, open(LOG, "w"))

    def reset(self):
        pass

    def run(self):
        time.sleep(STEP_S)
        return {"n_iters": 15, "n_trials": 15, "n_reduced": 1644, "iters": [{}] * 15}

    def read(self):
        return None

    def close(self):
        pass


class StubBackend:
    def create(self, problem, opt):
        return StubSolver(problem, opt)


def main():
    # ---- stand-ins for the device ----
    torch.cuda.is_available = lambda: True
    torch.cuda.set_device = lambda d: None
    torch.cuda.synchronize = lambda *a, **k: None
    torch.Tensor.cuda = lambda self, *a, **k: self
    real_tensor = torch.tensor
    torch.tensor = lambda data, **kw: real_tensor(data, **{k: v for k, v in kw.items() if k != "device"})
    real_init = dist.init_process_group
    dist.init_process_group = lambda backend, **kw: real_init("gloo")  # (RCCL needs GPUs; the collective calls bench.py makes are the same)
    B.load_library = lambda *a, **k: StubLib()
    B.hip_backend = lambda *a, **k: StubBackend()
    real_make = synth.make_problem
    synth.make_problem = lambda cfg, **kw: real_make("tiny" if cfg == "cfg4" else dict(n_kf=6, n_points=120, n_planes=4, n_cuboids=2, points_per_rank=True), **kw)
    if "cfg5w" in sys.argv:
        synth.CONFIGS["cfg5w"] = dict(synth.CONFIGS["cfg5w"])
    import bench
    bench.main()


if __name__ == "__main__":
    main()
