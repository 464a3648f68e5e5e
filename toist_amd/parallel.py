"""Data-parallel gradient exchange for one-process-per-GPU training (RCCL over xGMI).

Replaces the DistributedDataParallel wrap of /root/reference/main.py:335-337.  Every backward program
(heads, decoder, encoder, text encoder, backbone) writes its parameter gradients into one flat fp32
buffer; as soon as a program finishes, its buffer is all-reduced (mean) asynchronously on RCCL's
stream while the remaining programs keep computing.  `finish()` joins before clip / optimizer.
Five large collectives per step instead of hundreds of 25 MB buckets: xGMI links are point-to-point,
so few large messages keep every link busy.
"""
import torch
import torch.distributed as dist

from . import functions


class GradSync:
    def __init__(self, process_group=None):
        self.group = process_group
        self.pending = []
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1

    def __enter__(self):
        if self.world > 1:
            functions.GRAD_SYNC = self._launch
        return self

    def __exit__(self, *exc):
        functions.GRAD_SYNC = None
        return False

    def _launch(self, flat):
        op = _mean_op(self.group)
        work = dist.all_reduce(flat, op=op, group=self.group, async_op=True)
        self.pending.append((work, flat, op))

    def finish(self):
        """Wait for every outstanding all-reduce (call after loss.backward(), before clipping)."""
        for work, flat, op in self.pending:
            work.wait()
            if op == dist.ReduceOp.SUM:
                flat.div_(self.world)
        self.pending = []


def _mean_op(group=None):
    """RCCL averages in the collective; gloo (CPU tests, smoke runs) sums and the caller divides."""
    return dist.ReduceOp.AVG if dist.get_backend(group) == "nccl" else dist.ReduceOp.SUM


class _MeanHandle:
    def __init__(self, works, flats, op, world):
        self.works, self.flats, self.op, self.world = works, flats, op, world

    def wait(self):
        for w in self.works:
            w.wait()
        if self.op == dist.ReduceOp.SUM:
            for f in self.flats:
                f.div_(self.world)


def all_reduce_mean_async(flats, group=None):
    """Start the mean all-reduce of a list of flat gradient buffers on the collective stream; .wait() joins."""
    op = _mean_op(group)
    works = [dist.all_reduce(f, op=op, group=group, async_op=True) for f in flats]
    return _MeanHandle(works, list(flats), op, dist.get_world_size(group))


def all_reduce_mean(flats, group=None):
    """Blocking mean all-reduce of a list of flat gradient buffers."""
    all_reduce_mean_async(flats, group).wait()


def enable_backward_cuts(model, on=True):
    """Make `model` (MDETR or DETRsegm) cut the autograd graph at the outputs of the backbone and of the text encoder:
    loss.backward() then leaves those two programs to backward_cut(memory_cache, "text" / "backbone"), so the gradients
    of every finished segment can be all-reduced underneath the segments still running."""
    getattr(model, "detr", model).split_backward = bool(on)


def backward_cut(memory_cache, name):
    """Run the backward pass of the segment cut off under `name` ("text" or "backbone") from the gradients left at the cut."""
    cut = memory_cache.get("_native", {}).get("cuts", {}).get(name)
    if not cut:
        return
    pairs = [(src, leaf.grad) for src, leaf in zip(*cut) if leaf.grad is not None and src.requires_grad]
    if pairs:
        torch.autograd.backward([s for s, _ in pairs], [g for _, g in pairs])


def broadcast_parameters(module, src=0, group=None):
    """Make every rank start from rank `src`'s parameters and buffers (what DDP does at wrap time)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) < 2:
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src=src, group=group)
