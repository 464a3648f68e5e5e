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
    """with GradSync(model): loss.backward(); sync.finish() -- every program's flat gradient buffer is all-reduced (mean) as soon as
    its backward has finished; finish() joins them and then reduces whatever the flat buffers did not cover: parameters
    differentiated by plain autograd, and programs that accumulated into gradients kept from an earlier backward (gradient
    accumulation: wrap only the LAST micro-step, like DDP.no_sync() around the others).  Without `model` only flat buffers are
    reduced (the round-1 behaviour).

    Which stand-alone gradients exist may differ between ranks (a conditionally used module, an empty micro-batch: the reference wraps
    its model with find_unused_parameters=True, main.py:335-337).  Every rank must nevertheless issue the same collectives, so the ranks
    agree on the set first: one small SUM all-reduce of per-parameter one-hot counts (no gradient | in a flat buffer | stand-alone).
      * stand-alone on any rank           -> every rank reduces it stand-alone (a rank without a gradient contributes zeros)
      * flat on one rank, stand-alone on another -> RuntimeError on EVERY rank (the flat buffer has already been averaged with a slot that
        does not hold the other rank's gradient; all ranks see the same counts, so none is left waiting in a collective)
    static_set=True: the agreement of the first finish() is kept -- no collective and no host read in later steps; a rank whose own
    states change raises.  Use it when every rank runs the same programs every step (the training loop of engine.py:54-101)."""

    def __init__(self, model=None, process_group=None, static_set=False):
        if model is not None and not isinstance(model, (torch.nn.Module, list, tuple)):      # GradSync(process_group) of round 1
            model, process_group = None, model
        self.models = [] if model is None else (list(model) if isinstance(model, (list, tuple)) else [model])
        self.group = process_group
        self.pending = []
        self.ranges = []
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.static_set = static_set
        self._agreed = None          # (local states, indices of the parameters every rank reduces stand-alone)

    def __enter__(self):
        if self.world > 1:
            functions.GRAD_SYNC = self._launch
        return self

    def __exit__(self, *exc):
        functions.GRAD_SYNC = None
        return False

    def _launch(self, flat):
        if flat is None:
            return
        op = _mean_op(self.group)
        work = dist.all_reduce(flat, op=op, group=self.group, async_op=True)
        self.pending.append((work, flat, op))
        self.ranges.append((flat.data_ptr(), flat.data_ptr() + flat.numel() * flat.element_size()))

    def _local_states(self, params):
        state = []
        for p in params:
            g = p.grad
            if g is None:
                state.append(0)
            else:
                a = g.data_ptr()
                state.append(1 if any(lo <= a < hi for lo, hi in self.ranges) else 2)
        return state

    def _agree(self, params, state):
        """indices of the parameters that every rank reduces stand-alone; raises on every rank for a flat / stand-alone conflict"""
        if self.static_set and self._agreed is not None:
            if self._agreed[0] != state:
                raise RuntimeError("GradSync(static_set=True): this rank's gradient states changed since the first step; use the default (agreement every step)")
            return self._agreed[1]
        dev = next((p.grad.device for p in params if p.grad is not None), params[0].device)
        onehot = torch.zeros(len(params), 3, dtype=torch.int32)
        onehot[torch.arange(len(params)), torch.tensor(state, dtype=torch.int64)] = 1
        counts = onehot.to(dev)
        dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=self.group)
        counts = counts.cpu()
        conflict = ((counts[:, 1] > 0) & (counts[:, 2] > 0)).nonzero().flatten().tolist()
        if conflict:
            raise RuntimeError("GradSync: %d parameter gradient(s) travelled in a flat program buffer on some ranks and stand-alone on others "
                               "(first index %d): the ranks ran different backward schedules; zero the gradients on every rank before the step "
                               "or accumulate on all of them" % (len(conflict), conflict[0]))
        alone = (counts[:, 2] > 0).nonzero().flatten().tolist()
        if self.static_set:
            self._agreed = (list(state), alone)
        return alone

    def finish(self):
        """Wait for every outstanding all-reduce, then reduce the gradients no flat buffer carried (call after loss.backward(),
        before clipping)."""
        for work, flat, op in self.pending:
            work.wait()
            if op == dist.ReduceOp.SUM:
                flat.div_(self.world)
        rest = []
        if self.world > 1 and self.models:
            params = [q for m in self.models for q in m.parameters() if q.requires_grad]
            if params:
                try:
                    for i in self._agree(params, self._local_states(params)):
                        p = params[i]
                        if p.grad is None:
                            p.grad = torch.zeros_like(p)
                        rest.append(p.grad)
                finally:
                    self.pending, self.ranges = [], []
        if rest:
            all_reduce_mean(rest, self.group)
        self.pending, self.ranges = [], []


def _mean_op(group=None):
    """RCCL averages in the collective; gloo (CPU tests, smoke runs) sums and the caller divides."""
    return dist.ReduceOp.AVG if dist.get_backend(group) == "nccl" else dist.ReduceOp.SUM


class _MeanHandle:
    def __init__(self, works, flats, op, world, wire=None):
        self.works, self.flats, self.op, self.world, self.wire = works, flats, op, world, wire

    def wait(self):
        for w in self.works:
            w.wait()
        if self.wire is not None:                  # bf16 on the wire: back into the fp32 buffers the optimizer reads
            for f, t in zip(self.flats, self.wire):
                f.copy_(t)
        if self.op == dist.ReduceOp.SUM:
            for f in self.flats:
                f.div_(self.world)


def all_reduce_mean_async(flats, group=None, bf16=False):
    """Start the mean all-reduce of a list of flat gradient buffers on the collective stream; .wait() joins.  bf16=True halves the
    bytes on the xGMI links (the per-link-bound ring all-reduce of the 170 MB backbone buffer is the one collective nothing is left
    to overlap with): the buffers travel as bfloat16 copies and are written back as fp32 (the reference's DDP reduces in fp32, so
    this is an option, off by default)."""
    op = _mean_op(group)
    wire = [f.to(torch.bfloat16) for f in flats] if bf16 else None
    works = [dist.all_reduce(t, op=op, group=group, async_op=True) for t in (wire if bf16 else flats)]
    return _MeanHandle(works, list(flats), op, dist.get_world_size(group), wire)


def measure_all_reduce(flats, group=None, iters=5, bf16=False):
    """Achieved bandwidth of the mean all-reduce of `flats`, alone on the machine: {"bytes", "ms", "algbw_GBps", "busbw_GBps"} with
    busbw = algbw * 2 (n - 1) / n (the per-link traffic of a ring), the number to hold against ~153 GB/s per xGMI link."""
    import time
    world = dist.get_world_size(group)
    nbytes = sum(f.numel() * (2 if bf16 else f.element_size()) for f in flats)
    scratch = [torch.zeros_like(f) for f in flats]
    all_reduce_mean_async(scratch, group, bf16).wait()
    torch.cuda.synchronize()
    dist.barrier(group)
    t0 = time.perf_counter()
    for _ in range(iters):
        all_reduce_mean_async(scratch, group, bf16).wait()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    alg = nbytes / dt / 1e9
    return {"bytes": nbytes, "ms": round(1000 * dt, 3), "algbw_GBps": round(alg, 1), "busbw_GBps": round(alg * 2 * (world - 1) / world, 1)}


def all_reduce_mean(flats, group=None):
    """Blocking mean all-reduce of a list of flat gradient buffers."""
    all_reduce_mean_async(flats, group).wait()


def enable_backward_cuts(model, on=True):
    """Make `model` (MDETR or DETRsegm) cut the autograd graph at the outputs of the backbone and of the text encoder -- and, for the
    detection model, between the three stage programs of the ResNet body (layer4 | layer3 | stem .. layer2): loss.backward() then
    leaves those programs to backward_cut(memory_cache, name) for name in BACKWARD_CUTS, so the gradients of every finished segment
    can be all-reduced underneath the segments still running (only layer2's 5 MB are left without cover)."""
    getattr(model, "detr", model).split_backward = bool(on)


BACKWARD_CUTS = ("text", "backbone", "backbone.layer3", "backbone.layer2")     # the order a step runs them in after loss.backward()


def backward_cut(memory_cache, name):
    """Run the backward pass of the segment cut off under `name` from the gradients left at the cut: "text" (RoBERTa + resizer),
    "backbone" (ResNet layer4 -- or the whole body when it ran as one program), then "backbone.layer3" and "backbone.layer2"
    (stem .. layer2; frozen below layer2) when the body ran as three stage programs (detection model with cuts enabled).  Each
    segment's flat gradient buffer is complete when its call returns: all-reduce it while the next segment runs."""
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


class DistributedDataParallel(torch.nn.Module):
    """Constructor / `.module` / forward surface of torch.nn.parallel.DistributedDataParallel (as used at
    /root/reference/main.py:335-337, 344-346) on top of the flat-buffer gradient exchange: parameters are broadcast from rank 0 at
    wrap time; during backward every program's flat gradient buffer is all-reduced (mean) asynchronously as soon as the program is
    done, and a callback queued on the autograd engine joins the collectives (and reduces gradients that did not travel in a flat
    buffer) before loss.backward() returns -- so clip_grad_norm_ / optimizer.step() see averaged gradients exactly as with torch's
    DDP.  torch's own DistributedDataParallel(model) also works (functions.PARAM_GRADS, "autograd" path); this class is the faster
    route: 5 large in-place collectives instead of per-parameter bucket copies.  `find_unused_parameters` is accepted and ignored
    (unused parameters simply have no gradient); no_sync() skips the exchange for gradient accumulation."""

    def __init__(self, module, device_ids=None, output_device=None, find_unused_parameters=False, process_group=None, broadcast_buffers=True,
                 **unused):
        super().__init__()
        self.module = module
        self.process_group = process_group
        self.require_backward_grad_sync = True
        self._sync = GradSync(module, process_group)
        self._queued = False
        broadcast_parameters(module, 0, process_group)

    def forward(self, *args, **kwargs):
        functions.GRAD_SYNC = None          # a forward whose backward never ran (NaN skip, exception, no_sync) must not leave its hook behind
        if self.require_backward_grad_sync and self._sync.world > 1 and torch.is_grad_enabled():
            functions.GRAD_SYNC = self._on_flat
        return self.module(*args, **kwargs)

    def _on_flat(self, flat):
        if not self._queued:
            self._queued = True
            torch.autograd.Variable._execution_engine.queue_callback(self._finish)
        self._sync._launch(flat)

    def _finish(self):
        functions.GRAD_SYNC = None
        self._queued = False
        self._sync.finish()

    class _NoSync:
        def __init__(self, ddp):
            self.ddp = ddp

        def __enter__(self):
            self.old, self.ddp.require_backward_grad_sync = self.ddp.require_backward_grad_sync, False

        def __exit__(self, *exc):
            self.ddp.require_backward_grad_sync = self.old
            return False

    def no_sync(self):
        return DistributedDataParallel._NoSync(self)
