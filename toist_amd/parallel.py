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


def all_reduce_mean(flats, group=None):
    """Blocking mean all-reduce of a list of flat gradient buffers (used between the two hipGraphs of bench.py)."""
    op = _mean_op(group)
    works = [dist.all_reduce(f, op=op, group=group, async_op=True) for f in flats]
    for w in works:
        w.wait()
    if op == dist.ReduceOp.SUM:
        world = dist.get_world_size(group)
        for f in flats:
            f.div_(world)


def broadcast_parameters(module, src=0, group=None):
    """Make every rank start from rank `src`'s parameters and buffers (what DDP does at wrap time)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) < 2:
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src=src, group=group)
