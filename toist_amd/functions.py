"""Bridge between torch.autograd and the explicit forward/backward programs of toist_amd.engine.

`run_program` executes a forward program on the HIP kernels and registers ONE autograd node whose
backward replays the program's tape.  torch.autograd only carries tensors between these coarse nodes
(backbone, text encoder, encoder, decoder, heads); it never differentiates through individual ops.

Parameter gradients leave a program in one of two ways (PARAM_GRADS):
  "attach"   (default) the views of the program's flat fp32 buffer are attached to `.grad` directly and the
             GRAD_SYNC hook (toist_amd.parallel.GradSync / parallel.DistributedDataParallel) all-reduces the
             flat buffer in place -- no AccumulateGrad nodes run, so torch's DDP hooks would NOT fire;
  "autograd" the node returns them to torch.autograd, AccumulateGrad stores them and every hook registered
             on a parameter fires: this is what torch.nn.parallel.DistributedDataParallel(model) needs
             (/root/reference/main.py:335-337).  It is selected automatically while a torch DDP forward is
             active, so a driver that wraps the model as the reference does gets reduced gradients.
"""
import torch

from . import engine, kernels

# Optional callable(flat_grad_buffer or None) invoked when a program's backward has produced all of its
# parameter gradients (set by toist_amd.parallel.GradSync / DistributedDataParallel).
GRAD_SYNC = None

# Callables(params tuple) invoked on the program's stream when a program's backward has attached all of its parameter gradients
# (toist_amd.optim.FusedClipAdamWEMA: the squared gradient norm of an "early_norm" group is taken right there, beside the rest of the backward pass).
AFTER_BACKWARD = []

# Stream the programs launched right now were forked from (set by MDETR.encode around the text branch).
# Their parameter gradients are attached to .grad directly, not through AccumulateGrad nodes, so the
# autograd engine does not know it has to join their stream at the end of backward(): each forked
# program's backward makes REJOIN wait for it instead.
REJOIN = None

PARAM_GRADS = "auto"     # "auto": "autograd" inside a torch DistributedDataParallel forward, "attach" otherwise


def _via_autograd():
    if PARAM_GRADS != "auto":
        return PARAM_GRADS == "autograd"
    try:
        from torch.nn.parallel import DistributedDataParallel as _DDP
        return getattr(_DDP, "_active_ddp_module", None) is not None
    except Exception:  # pragma: no cover
        return False


class _TapeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, body, names, cache, n_in, tape_kw, *tensors):
        inputs = tensors[:n_in]
        params = tensors[n_in:]
        tape_kw = dict(tape_kw)
        ctx.rejoin = tape_kw.pop('rejoin')
        ctx.via_autograd = tape_kw.pop('via_autograd')
        need = tape_kw.pop('need_grads')  # grad mode is off inside Function.forward: decided by the caller
        transforms = tape_kw.pop('transforms')
        store_once = tape_kw.pop('store_once')
        named = dict(zip(names, params))
        trainable = {n: p.requires_grad for n, p in named.items()}
        ps = engine.ParamSet(named, trainable, need_grads=need, bf16_cache=cache, transforms=transforms, store_once=store_once)
        tape = engine.Tape(**tape_kw)
        tape.fresh_views = ps.fresh_views() if need else []
        in_vars = [None if t is None else engine.Var(t, needs_grad=(need and t.requires_grad and t.is_floating_point()))
                   for t in inputs]
        out_vars, extra = body(tape, ps, *in_vars)
        if need:
            ctx.tape, ctx.ps, ctx.in_vars, ctx.out_vars = tape, ps, in_vars, out_vars
            ctx.params = params
        else:
            tape.steps = []
        ctx.n_in = n_in
        ctx.label = names[0].split(".")[0] if names else "program"
        ctx.n_par = len(params)
        ctx.extra = extra
        non_diff = [v.data for v in out_vars if not v.needs_grad]
        if non_diff:
            ctx.mark_non_differentiable(*non_diff)
        return tuple(v.data for v in out_vars)

    @staticmethod
    def backward(ctx, *grads):
        tape, ps = ctx.tape, ctx.ps
        if kernels.STAMPS is not None:
            kernels.stamp("bwd." + ctx.label + ".start")
        for v, g in zip(ctx.out_vars, grads):
            if g is not None and v.needs_grad:
                if v.raw_grad:           # the program inspects it first (segmentation: an untouched zero sentinel = "only the mask losses read pred_masks")
                    v.grad = g
                    continue
                if g.dtype != v.data.dtype:
                    g = g.to(v.data.dtype)
                v.grad = g.contiguous()
        tape.backward()
        if kernels.STAMPS is not None:
            kernels.stamp("bwd." + ctx.label + ".end")
        in_grads = []
        for v in ctx.in_vars:
            if v is None or not v.needs_grad:
                in_grads.append(None)
            else:
                in_grads.append(v.take_grad())
        n_par = len(ctx.params)
        if ctx.via_autograd:
            # torch DistributedDataParallel (or any other per-parameter hook) is listening: hand the gradients to autograd
            par_grads = [g if (g is not None and p.requires_grad) else None for p, g in zip(ctx.params, ps.grads())]
            if ctx.rejoin is not None:
                ctx.rejoin.wait_stream(torch.cuda.current_stream())
            ctx.tape = ctx.ps = ctx.in_vars = ctx.out_vars = ctx.params = None
            del tape, ps
            return (None, None, None, None, None, *in_grads, *par_grads)
        # Parameter gradients live in ONE flat fp32 buffer per program; they are attached to .grad
        # directly (no autograd accumulation copies), so a data-parallel all-reduce can run in place
        # on the flat buffer while the rest of the backward pass proceeds (toist_amd/parallel.py).
        for p, g in zip(ctx.params, ps.grads()):
            if g is None:
                continue
            if p.grad is None:
                p.grad = g
            else:
                p.grad.add_(g)
                ps.flat = None  # accumulated into an older buffer: GradSync.finish() reduces those gradients one by one
        if GRAD_SYNC is not None:
            GRAD_SYNC(ps.flat)      # None: this program accumulated into older gradients (the hook still learns that a backward ran)
        elif AFTER_BACKWARD:        # (a gradient all-reduce would change the values after this point)
            for hook in AFTER_BACKWARD:
                hook(ctx.params)
        if ctx.rejoin is not None:
            ctx.rejoin.wait_stream(torch.cuda.current_stream())
        ctx.tape = ctx.ps = ctx.in_vars = ctx.out_vars = ctx.params = None
        return (None, None, None, None, None, *in_grads, *([None] * n_par))


def run_program(body, named_params, inputs, cache=None, training=False, drop_p=0.0, seed=0, transforms=None, group_wgrads=False, store_once=None):
    """body(tape, ps, *input_vars) -> (list_of_output_vars, extra).  Returns (outputs_tuple)."""
    names = tuple(named_params.keys())
    params = tuple(named_params.values())
    need = torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (*inputs, *params))
    tape_kw = dict(training=training, drop_p=drop_p, seed=seed, need_grads=need, transforms=transforms, rejoin=REJOIN, group_wgrads=group_wgrads,
                   via_autograd=need and _via_autograd(), store_once=store_once)
    return _TapeFn.apply(body, names, cache, len(inputs), tape_kw, *inputs, *params)
