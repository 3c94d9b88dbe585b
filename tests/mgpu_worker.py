"""Worker of tests/test_multigpu.py: one process per GPU under torch.distributed.run, real RCCL between the ranks.

Every rank builds the same seeded graph, creates a solver for its landmark range (options.rank / world_size / comm_id from
ncclGetUniqueId on rank 0, broadcast over the torch process group), runs the 5 + 10 protocol and reads the result back (read
gathers the landmarks from their owners).  Rank 0 also solves the graph alone and prints one JSON line with the comparison."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "point-plane-object-slam_amd"))
from ppo_ba import binding as B  # noqa: E402
from ppo_ba import synth  # noqa: E402


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    cfg = sys.argv[1] if len(sys.argv) > 1 else "small"
    stop_rank = int(sys.argv[2]) if len(sys.argv) > 2 else -1  # >= 0: that rank raises its stop flag before the run
    on_owner = int(sys.argv[3]) if len(sys.argv) > 3 else 0    # options.solve_on_owner: reduce to rank 0 + broadcast of x instead of the all-reduce
    P = synth.make_problem(cfg, seed=1)
    lib = B.load_library()
    ident = np.zeros(128, np.uint8)
    if rank == 0:
        assert lib.ppo_ba_comm_unique_id(ident.ctypes.data_as(B.c_bp)) == 0
    t = torch.from_numpy(ident).cuda()
    dist.broadcast(t, 0)
    ident = t.cpu().numpy()
    opt = B.default_options(device_id=local, rank=rank, world_size=world, solve_on_owner=on_owner)
    for i in range(128):
        opt.comm_id[i] = int(ident[i])
    s = B.hip_backend().create(P, opt)
    flag = np.zeros(1, np.uint8)
    if rank == stop_rank:
        flag[0] = 1
    st = s.run(stop_flag=flag)
    rc = s.last_rc
    r = s.read()
    s.close()
    out = {"rank": rank, "rc": rc, "iters": [(i["trials"], i["accepted"], i["result"]) for i in st["iters"]]}
    if rank == 0 and stop_rank < 0:
        s1 = B.hip_backend().create(P, B.default_options(device_id=local))
        st1, r1 = s1.run(), s1.read()
        s1.close()
        out["same_schedule"] = out["iters"] == [(i["trials"], i["accepted"], i["result"]) for i in st1["iters"]]
        out["dchi2"] = abs(st["iters"][-1]["chi2_after"] - st1["iters"][-1]["chi2_after"]) / st1["iters"][-1]["chi2_after"]
        out["t_rmse"] = float(np.sqrt(np.mean(np.sum((r.pose_qt[:, 4:] - r1.pose_qt[:, 4:]) ** 2, axis=1))))
        out["p_rmse"] = float(np.sqrt(np.mean(np.sum((r.point_xyz - r1.point_xyz) ** 2, axis=1))))
        out["levels_equal"] = bool(np.array_equal(r.mono_level, r1.mono_level))
    gathered = [None] * world
    dist.all_gather_object(gathered, out)
    if rank == 0:
        print("MGPU " + json.dumps(gathered))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
