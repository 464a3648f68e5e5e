"""torch.distributed helpers with the reference's names (/root/reference/util/dist.py:136-229).

One process per GPU; backend "nccl" on PyTorch-ROCm IS RCCL (xGMI), "gloo" for CPU tests."""
import os

import torch
import torch.distributed as dist


def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank():
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def is_main_process():
    return get_rank() == 0


def reduce_dict(input_dict, average=True):
    """All-reduce a dict of scalar tensors in sorted-key order; mean when `average` (dist.py:93-117)."""
    world = get_world_size()
    if world < 2:
        return input_dict
    with torch.no_grad():
        names = sorted(input_dict.keys())
        values = torch.stack([input_dict[k] for k in names], dim=0)
        dist.all_reduce(values)
        if average:
            values = values / world
        return dict(zip(names, values))


def init_distributed_mode(args):
    """Read RANK / WORLD_SIZE / LOCAL_RANK from the environment and start the RCCL process group."""
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ:
        args.rank = int(os.environ["RANK"])
        args.world_size = int(os.environ["WORLD_SIZE"])
        args.gpu = int(os.environ.get("LOCAL_RANK", 0))
    else:
        args.distributed = False
        return
    args.distributed = True
    backend = "nccl" if torch.cuda.is_available() else "gloo"
    if torch.cuda.is_available():
        torch.cuda.set_device(args.gpu)
    dist.init_process_group(backend=backend, init_method=getattr(args, "dist_url", "env://"), world_size=args.world_size,
                            rank=args.rank)
    dist.barrier()
