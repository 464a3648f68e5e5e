"""world_size-2 gloo tests of the data-parallel pieces (CPU): reduce_dict, the num_boxes all-reduce of
the set criterion (sum, / world, clamp >= 1; reference mdetr.py:997-1001) and the flat-buffer gradient
all-reduce (mean) of toist_amd.parallel.GradSync."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from toist_amd import dist as tdist
    from toist_amd import functions, parallel
    from toist_amd.mdetr import SetCriterion
    res = {}
    red = tdist.reduce_dict({"b": torch.tensor(float(rank + 1)), "a": torch.tensor(10.0 * (rank + 1))})
    res["reduce"] = (float(red["a"]), float(red["b"]))
    crit = SetCriterion(None, 255, matcher=None, eos_coef=0.1, losses=["labels"], temperature=0.07)
    targets = [{"labels": torch.ones(3 if rank == 0 else 0)}]
    res["num_boxes"] = float(crit._num_boxes(targets, torch.device("cpu")))
    res["num_boxes_empty"] = float(crit._num_boxes([{"labels": torch.ones(0)}], torch.device("cpu")))
    sync = parallel.GradSync()
    flat = torch.full((10,), float(rank + 1))
    with sync:
        assert functions.GRAD_SYNC is not None
        functions.GRAD_SYNC(flat)
        sync.finish()
    assert functions.GRAD_SYNC is None
    res["flat"] = flat.tolist()
    lin = torch.nn.Linear(2, 2)
    torch.manual_seed(rank)
    torch.nn.init.normal_(lin.weight)
    parallel.broadcast_parameters(lin)
    res["w"] = lin.weight.detach().flatten().tolist()
    out[rank] = res
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_two_gloo():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    r0, r1 = out[0], out[1]
    assert r0["reduce"] == r1["reduce"] == (15.0, 1.5)
    assert r0["num_boxes"] == r1["num_boxes"] == 1.5   # (3 + 0) / 2
    assert r0["num_boxes_empty"] == 1.0                # clamp(min=1)
    assert r0["flat"] == r1["flat"] == [1.5] * 10      # mean of 1 and 2
    assert r0["w"] == r1["w"]
