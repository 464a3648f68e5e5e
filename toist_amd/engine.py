"""Explicit forward/backward programs for the TOIST hot path, built from the HIP kernel calls of
toist_amd.ops / toist_amd.kernels.

There is no tracing compiler and no per-op torch autograd here: a forward pass appends its backward
closures to a Tape, and the enclosing torch.autograd.Function (see toist_amd/functions.py) replays
them in reverse.  Activations are bf16, statistics / parameter gradients fp32.  Activation-function
and FrozenBN backward steps are folded into the epilogue of the GEMM that produces the gradient.
"""
import math

import weakref

import torch

from . import kernels as k
from . import ops
from .knobs import knob

BF16 = torch.bfloat16


class Var:
    """An activation (bf16, [rows, features] or NHWC) and its gradient slot."""

    __slots__ = ("data", "grad", "needs_grad", "drop", "gdrop", "plus", "raw_grad")

    def __init__(self, data, needs_grad=True):
        self.data = data
        self.grad = None
        self.needs_grad = needs_grad
        self.raw_grad = False   # True: the program's backward wants the output gradient exactly as autograd delivered it (no dtype cast, no .contiguous())
        self.plus = None    # data + a constant embedding (pos / query_pos), when the producing layernorm emitted it in the same pass
        self.drop = None    # (p, seed) when data = residual + dropout(branch): the branch gradient is dropout(grad) with that mask
        self.gdrop = None   # that masked gradient, when the consumer's backward produced it in the same pass (layernorm)

    def take_branch_grad(self, g):
        """dropout-masked copy of the gradient g for the `dropout(branch)` term (one extra launch unless the consumer made it)."""
        gd, self.gdrop = self.gdrop, None
        if gd is None:
            gd = torch.empty_like(g)
            k.dropout(g, self.drop[0], self.drop[1], gd)
        return gd

    def take_grad(self):
        g, self.grad = self.grad, None
        return g


# Independent branches of the step are forked onto side HIP streams so that, inside the captured hipGraph,
# they become parallel branches: the small GEMMs of one branch fill the ramp-up / tail bubbles of the other.
#   "capture": only while a hipGraph is being captured;  "on": always (tests);  "off": never.
# Measured on MI355X (B=8, 640x640): text branch || image branch +8.7 % images/s (+14 % once the rest of the step had shrunk).  Forking the weight-gradient
# GEMMs from the data-gradient chain was measured too (0 % to -6 %: those kernels already fill the chip and
# every fork is a cross-stream edge of the graph) and is deliberately not done.
OVERLAP = knob("TOIST_OVERLAP", "capture")   # "capture": fork the text branch inside captured graphs only; "on" / "off"
FUSED_ATTENTION = True   # head-dim-32 attention cores run as one flash-style launch each way (csrc/attn2.hip); False = batched score GEMMs + softmax kernels
FUSED_BLOCKS = True      # encoder / decoder layers: packed in_proj in one launch, decoder K/V of all layers grouped, LayerNorm emits y + pos
_SIDE = {}
SIDE_PRIORITY = knob("TOIST_SIDE_PRIORITY", 0)   # -1 = high: the small kernels of a side branch get free CU slots first


# Work that must run at the head of the text branch, on its stream, before RoBERTa reads its weights: the late parameter groups of
# toist_amd.optim.FusedClipAdamWEMA (the text encoder's AdamW + EMA launch of the PREVIOUS step runs beside the ResNet forward).
_TEXT_PRELUDE = []


def register_text_prelude(opt, late_params):
    """opt.flush_late() will run at the head of every text branch whose encoder owns one of `late_params` (other models are not touched)"""
    _TEXT_PRELUDE[:] = [(r, ids) for r, ids in _TEXT_PRELUDE if r() is not None]
    _TEXT_PRELUDE.append((weakref.ref(opt), frozenset(id(p) for p in late_params)))


def run_text_prelude(param):
    """`param`: any parameter of the text encoder about to run"""
    for r, ids in _TEXT_PRELUDE:
        opt = r()
        if opt is not None and id(param) in ids:
            opt.flush_late()


def side_stream(device, name):
    key = (device.index, name)
    s = _SIDE.get(key)
    if s is None:
        s = _SIDE[key] = torch.cuda.Stream(device=device, priority=SIDE_PRIORITY)
    return s


def overlap_enabled():
    if OVERLAP == "on":
        return True
    return OVERLAP == "capture" and torch.cuda.is_current_stream_capturing()


GROUP_WGRADS = knob("TOIST_GROUP_WGRADS", True)   # same-shape weight gradients of a program run as grouped launches at its end


class Tape:
    def __init__(self, training, drop_p=0.0, seed=0, group_wgrads=False):
        self.groups = {} if (group_wgrads and GROUP_WGRADS) else None   # (shapes, geometry) -> [(dy, x, out, rscale)]
        self.steps = []
        self.training = training
        self.drop_p = drop_p if training else 0.0
        self._seed = seed * 1000003 + 12345
        self.keep = []  # tensors that must outlive the forward (API outputs etc.)
        self.fresh_views = []   # ParamViews with store-once gradient slots (ParamSet)

    def next_seed(self):
        self._seed += 7919
        return self._seed & 0x7FFFFFFFFFFF

    def record(self, fn):
        self.steps.append(fn)

    def conv_wgrad(self, dy, x, w_shape, out, rscale, stride=1, pad=0, view=None):
        """Weight gradient of a convolution: now, or (programs with group_wgrads) collected with the other convolutions of the same
        shape and launched grouped when the program's backward ends.  The operands are never written again by this backward.
        view = the ParamView whose gradient slot `out` is: a store-once slot (view.fresh) is overwritten, not accumulated into."""
        store = view is not None and view.fresh and not view.written
        if view is not None:
            view.mark_written()
        if self.groups is None:
            ops.conv2d_wgrad(dy, x, w_shape, stride=stride, pad=pad, out=out, rscale=rscale, defer=True, accumulate=not store)
        else:
            self.groups.setdefault((tuple(dy.shape), tuple(x.shape), tuple(w_shape), stride, pad, store), []).append((dy, x, out, rscale))

    def linear_wgrad(self, dy, x, W, b, owned=True):
        """Weight / bias gradient of an nn.Linear: now, or grouped with the same-shape layers of the program.  `owned` = no later step
        of this backward writes into dy (a gradient that doubles as a residual's accumulator is not owned and is consumed at once).
        A store-once slot (W.fresh, first launch on it) is overwritten instead of accumulated into; bias slots always accumulate."""
        bias_out = b.g if b is not None else None
        store = W.fresh and not W.written
        W.mark_written()
        if self.groups is None or not owned:
            ops.linear_wgrad(dy, x, out=W.g, bias_out=bias_out, defer=True, accumulate=not store)
        else:
            key = ("linear", tuple(dy.shape), dy.stride(0), tuple(x.shape), x.stride(0), W.g.stride(0), bias_out is None, store)
            self.groups.setdefault(key, []).append((dy, x, W.g, bias_out))

    def backward(self):
        for fn in reversed(self.steps):
            fn()
        self.steps = []
        if self.groups:
            for key, items in self.groups.items():
                if key[0] == "linear":
                    ops.linear_wgrad_group(items, accumulate=not key[-1])
                else:
                    ops.conv2d_wgrad_group(items, key[2], stride=key[3], pad=key[4], accumulate=not key[5])
            self.groups = {}
        k.flush_reductions()   # deferred split-K partials of this program -> parameter gradients
        for v in self.fresh_views:      # store-once rows no weight-gradient launch reached (their branch received no gradient): zero, as an accumulating slot would read
            for a, b in v.unwritten_rows():
                if v.g.dim() == 0:
                    v.g.zero_()
                else:
                    v.g[a:b].zero_()


# ---- bf16 compute copies of the fp32 master weights ------------------------------------------------------------
# A copy is valid while the master's version counter, storage and the global WEIGHT_EPOCH are unchanged.  The
# epoch exists because fused optimizers (torch's fused AdamW, the HIP tail in toist_amd/optim.py) update
# parameters without touching the version counter: any optimizer step bumps it (global torch hook below), the
# HIP tail re-validates the copies it has rewritten itself.  Copies are refreshed IN PLACE so their addresses
# stay valid inside captured hipGraphs and in the optimizer's tensor table.
WEIGHT_EPOCH = 0
COPY_GEN = 0          # bumped whenever a copy is (re)allocated: holders of raw pointers rebuild their tables
COPIES = {}           # master data_ptr -> ComputeCopy


def bump_weight_epoch():
    global WEIGHT_EPOCH
    WEIGHT_EPOCH += 1


try:  # every torch optimizer step invalidates the compute copies
    from torch.optim.optimizer import register_optimizer_step_post_hook as _reg_post_hook
    _reg_post_hook(lambda *a, **k: bump_weight_epoch())
except Exception:  # pragma: no cover - older torch: copies are refreshed every training forward instead
    _reg_post_hook = None


class ComputeCopy:
    __slots__ = ("version", "ptr", "epoch", "w", "row_scale", "elementwise", "master")


def prune_copies():
    """Drop registry entries whose master tensor is gone (their address may since belong to another tensor)."""
    for ptr in [ptr for ptr, ent in COPIES.items() if ent.master() is None]:
        del COPIES[ptr]


def copy_of(t):
    """The registered compute copy of exactly this tensor object, or None (an address match alone is not enough: the
    caching allocator hands a freed parameter's address to the next model)."""
    ent = COPIES.get(t.data_ptr())
    return ent if ent is not None and ent.master() is t and ent.ptr == t.data_ptr() else None


def _cast_bf16(m):
    return m.detach().to(BF16)


_cast_bf16.elementwise = True


def compute_copy(t, make, cache, name):
    global COPY_GEN
    ent = cache.get(name) if cache is not None else None
    # a frozen tensor (requires_grad False) is outside every optimizer: only its version counter / storage can invalidate the copy
    if ent is not None and ent.version == t._version and ent.ptr == t.data_ptr() and (ent.epoch == WEIGHT_EPOCH or not t.requires_grad) \
            and _reg_post_hook is not None:
        return ent.w
    w = make(t)
    pinned = bool(getattr(make, "pinned", False))     # the transform owns the copy's storage (packed_cast): the entry must BE that view
    if ent is not None and pinned and ent.w.data_ptr() != w.data_ptr():
        ent = None                                    # a copy built by another transform of the same parameter: rebuild around the view
    if ent is not None and pinned:
        pass                                          # make() has refreshed the view in place
    elif ent is not None and ent.w.shape == w.shape and ent.w.stride() == w.stride():
        ent.w.copy_(w)
    else:
        ent = ComputeCopy()
        ent.w = w
        # elementwise: the copy has the master's physical element order, so the optimizer tail can rewrite it
        ent.elementwise = bool(getattr(make, "elementwise", False)) and w.shape == t.shape and w.stride() == t.stride()
        ent.row_scale = getattr(make, "row_scale", None)
        COPY_GEN += 1
        if cache is not None:
            cache[name] = ent
    ent.version, ent.ptr, ent.epoch, ent.master = t._version, t.data_ptr(), WEIGHT_EPOCH, weakref.ref(t)
    if cache is not None:
        COPIES[t.data_ptr()] = ent
    return ent.w


def named_cache(owner, key, build):
    """Per-module memo of an ordered {name: Parameter} dict (module traversals cost ~1 ms per step when repeated every
    forward).  Parameters are moved / loaded in place by nn.Module, so the objects stay valid; deepcopy copies the memo
    together with the parameters it points to."""
    memo = owner.__dict__.setdefault("_named_memo", {})
    ent = memo.get(key)
    if ent is None:
        ent = memo[key] = build()
    return ent


class ParamView:
    """bf16 compute copy + fp32 gradient slot of one nn.Parameter (or a row slice of one).

    Store-once bookkeeping lives on the ROOT view as a list of written row ranges: a slice made by rows() reads and writes the same
    record, so a second weight-gradient launch on an already stored range accumulates (whatever view object it came through), and
    rows no launch reached are zeroed at the end of the backward pass (unwritten_rows)."""

    __slots__ = ("w", "g", "f32", "fresh", "parent", "_lo", "_hi", "_ranges")

    def __init__(self, w_bf16, grad_f32, f32=None, fresh=False, parent=None, lo=None, hi=None):
        self.w = w_bf16    # bf16 tensor used by the kernels (None for fp32-only params)
        self.g = grad_f32  # fp32 gradient slot: zero-initialised accumulator, or (fresh) uninitialised memory a weight-gradient GEMM overwrites
        self.f32 = f32     # fp32 master values for params consumed in fp32 (biases, LN affine)
        self.fresh = fresh         # store-once gradient (see ParamSet): the first weight-gradient launch on a row range stores instead of accumulating
        self.parent = parent
        self._lo, self._hi = lo, hi   # row range inside the root (None: all of it)
        self._ranges = []          # root only: row ranges [a, b) covered by weight-gradient launches of the current backward; (None, None) = everything

    def _root(self):
        v = self
        while v.parent is not None:
            v = v.parent
        return v

    def _span(self):
        """this view's row range in root coordinates"""
        lo, hi, v = 0, None, self
        chain = []
        while v is not None:
            chain.append(v)
            v = v.parent
        for v in reversed(chain):              # root first: nested rows() offsets add up
            if v._lo is not None:
                lo, hi = lo + v._lo, lo + v._hi
        return lo, hi

    @property
    def written(self):
        """True when every row of this view has been stored by a weight-gradient launch of the current backward."""
        root = self._root()
        if any(a is None for a, _ in root._ranges):
            return True
        lo, hi = self._span()
        if hi is None:
            n = root.g.shape[0] if root.g is not None and root.g.dim() > 0 else 1
            lo, hi = 0, n
        pos = lo
        for a, b in sorted(root._ranges):
            if a > pos:
                break
            pos = max(pos, b)
        return pos >= hi

    @written.setter
    def written(self, value):
        if value:
            self.mark_written()
        else:
            self._root()._ranges = []

    def rows(self, a, b):
        """Row slice [a:b) of a packed parameter (e.g. the q / k / v blocks of in_proj_weight)."""
        return ParamView(None if self.w is None else self.w[a:b], None if self.g is None else self.g[a:b],
                         None if self.f32 is None else self.f32[a:b], fresh=self.fresh, parent=self, lo=a, hi=b)

    def mark_written(self):
        root = self._root()
        lo, hi = self._span()
        root._ranges.append((None, None) if hi is None else (lo, hi))

    def unwritten_rows(self):
        """root view: list of row ranges no launch has stored (the whole slot when nothing was written)."""
        if any(a is None for a, _ in self._ranges):
            return []
        n = self.g.shape[0] if self.g.dim() > 0 else 1
        gaps, pos = [], 0
        for a, b in sorted(self._ranges):
            if a > pos:
                gaps.append((pos, a))
            pos = max(pos, b)
        if pos < n:
            gaps.append((pos, n))
        return gaps


# Opt-in for training loops that never keep a parameter gradient beyond optimizer.zero_grad(): the flat gradient buffer of
# a program (and its ~600 per-parameter views) is then reused from step to step instead of being re-sliced every forward.
# Off by default: with it, a gradient tensor stashed by the caller would be overwritten by the next forward pass.
REUSE_GRAD_BUFFERS = False
STORE_ONCE = knob("TOIST_STORE_ONCE", True)      # weight gradients of nn.Linear / nn.Conv2d overwrite their (un-zeroed) slots instead of accumulating into zeroed ones
POISON_FRESH = False   # tests: fill the store-once slots with NaN before every backward


class ParamSet:
    """Per-call view of a module's parameters: bf16 weight copies and one flat fp32 gradient buffer.

    `named` is an ordered {name: tensor(fp32 master)}; weights (dim >= 2) get a bf16 copy, vectors
    are used in fp32 directly.  grads() returns gradients in the same order (None for frozen)."""

    def __init__(self, named, trainable, need_grads, bf16_cache=None, transforms=None, store_once=None):
        """store_once(name, tensor) -> True marks a parameter whose gradient is produced by exactly one weight-gradient GEMM per
        backward (every nn.Linear / nn.Conv2d weight of the hot path): its slot lives in the tail of the flat buffer, is NOT zeroed
        -- the GEMM overwrites it (Tape.linear_wgrad / conv_wgrad, accumulate=False) -- which removes 0.74 GB of zero-fill writes and
        as many read-modify-write reads per step.  Everything else (biases, LayerNorm and embedding gradients: atomics, scatter-adds)
        sits in the zeroed head of the buffer.  A store-once slot no launch reached is zeroed by Tape.backward."""
        self.names = list(named.keys())
        self.views = {}
        self.zero_elems = 0
        # The slicing of the flat gradient buffer into ~600 per-parameter views costs ~4 ms of host time per step; with
        # REUSE_GRAD_BUFFERS the layout is memoised in the program's cache and reused (buffer zeroed in place) as long as the
        # parameters are the same objects and no parameter's .grad still aliases the buffer (gradient accumulation gets a
        # fresh buffer, as without the option).
        memo = bf16_cache.get("__layout__") if bf16_cache is not None else None
        params = list(named.values())
        sig = (need_grads, len(params), tuple(trainable.get(n, False) for n in self.names), id(params[0]) if params else 0,
               params[0].data_ptr() if params else 0, id(params[-1]) if params else 0, params[-1].data_ptr() if params else 0,
               STORE_ONCE and store_once is not None)
        if REUSE_GRAD_BUFFERS and memo is not None and memo["sig"] == sig and need_grads and memo["flat"] is not None:
            lo, hi = memo["flat"].data_ptr(), memo["flat"].data_ptr() + memo["flat"].numel() * 4
            if not any(p.grad is not None and lo <= p.grad.data_ptr() < hi for p in params):
                self.flat = memo["flat"]
                self.zero_elems = memo["zero_elems"]
                self._clear()
                for n, t in named.items():
                    pv = memo["views"][n]
                    pv.written = False
                    if t.dim() >= 2:
                        pv.w = compute_copy(t, (transforms or {}).get(n) or _cast_bf16, bf16_cache, n)
                    self.views[n] = pv
                return
        fresh_names = set()
        if need_grads and STORE_ONCE and store_once is not None:
            fresh_names = {n for n in self.names if trainable.get(n, False) and store_once(n, named[n])}
        pad = lambda t: (t.numel() + 63) // 64 * 64
        total = 0
        if need_grads:
            self.zero_elems = sum(pad(named[n]) for n in self.names if trainable.get(n, False) and n not in fresh_names)
            total = self.zero_elems + sum(pad(named[n]) for n in fresh_names)
        dev = next(iter(named.values())).device if named else None
        self.flat = torch.empty(total, dtype=torch.float32, device=dev) if total else None
        self._clear()
        off_zero, off_fresh = 0, self.zero_elems
        for n in self.names:
            t = named[n]
            g = None
            if need_grads and trainable.get(n, False):
                off = off_fresh if n in fresh_names else off_zero
                g = self.flat[off:off + t.numel()].view(t.shape)
                if t.dim() == 4 and not t.is_contiguous() and t.is_contiguous(memory_format=torch.channels_last):
                    # channels_last conv weights (physically KRSC): the gradient keeps the parameter's layout
                    O, I, R, S = t.shape
                    g = self.flat[off:off + t.numel()].view(O, R, S, I).permute(0, 3, 1, 2)
                if n in fresh_names:
                    off_fresh += pad(t)
                else:
                    off_zero += pad(t)
            wb = None
            if t.dim() >= 2:
                make = (transforms or {}).get(n) or _cast_bf16
                wb = compute_copy(t, make, bf16_cache, n)
            self.views[n] = ParamView(wb, g, t.detach(), fresh=n in fresh_names)
        if REUSE_GRAD_BUFFERS and bf16_cache is not None and need_grads and (memo is None or memo["sig"] != sig):
            bf16_cache["__layout__"] = {"sig": sig, "flat": self.flat, "views": dict(self.views), "zero_elems": self.zero_elems}

    def _clear(self):
        """zero the accumulating head of the flat buffer; the store-once tail is left as it is (POISON_FRESH: NaN, so that a slot
        nothing wrote shows up in the tests)"""
        if self.flat is None:
            return
        if self.zero_elems:
            self.flat[:self.zero_elems].zero_()
        if POISON_FRESH and self.zero_elems < self.flat.numel():
            self.flat[self.zero_elems:].fill_(float("nan"))

    def fresh_views(self):
        return [v for v in self.views.values() if v.fresh and v.g is not None]

    def __getitem__(self, name):
        return self.views[name]

    def grads(self):
        return [self.views[n].g for n in self.names]


def krsc(w):
    """[Co,Cin,R,S] channels_last tensor -> its physical [Co,R,S,Cin] contiguous view."""
    v = w.permute(0, 2, 3, 1)
    assert v.is_contiguous(), "conv weights must be channels_last"
    return v


def accumulate(var, g):
    """var.grad += g (g becomes the slot when empty)."""
    if var.grad is None:
        var.grad = g
    else:
        k.add(var.grad, g, var.grad)


# ------------------------------------------------------------------------------------------ dense ops
def linear_chain(tape, x, layers, res=None, final_drop=False, out_dtype=BF16, last_act_external=False, in_relu_mask=False):
    """y = L_n(...L_1(x)), layer = (W, b, act[, dropout_after_act]); optional `+ res` after an
    optional dropout on the last layer's output (the transformer's `x + dropout(sublayer(x))`).
    Returns Var.  Activation backward of layer i is fused into the dgrad GEMM of layer i+1."""
    acts = []
    cur = x.data
    p = tape.drop_p
    seeds = []
    n = len(layers)
    for i, (W, b, act, drop_after) in enumerate(layers):
        last = i == n - 1
        kw = {}
        seed = 0
        if drop_after and p > 0:
            seed = tape.next_seed()
            kw = dict(drop_where=2, drop_p=p, drop_seed=seed)
        if last and final_drop and p > 0:
            seed = tape.next_seed()
            kw = dict(drop_where=1, drop_p=p, drop_seed=seed)
        pre = None
        if act == k.ACT_GELU:
            pre = torch.empty(cur.shape[0], W.w.shape[0], dtype=BF16, device=cur.device)
        y = ops.linear(cur, W.w, b.f32 if b is not None else None, act=act, res=res.data if (last and res is not None) else None,
                       pre_out=pre, out_dtype=out_dtype if last else BF16, **kw)
        acts.append((cur, y, pre))
        seeds.append(seed)
        cur = y
    out = Var(cur)
    if final_drop and p > 0:
        out.drop = (p, seeds[-1])

    def bwd():
        g = out.take_grad()
        if g is None:
            return
        if res is not None and res.needs_grad:
            accumulate(res, g)
            if res.grad is g:  # keep our own copy if we are about to modify g
                pass
        for i in range(n - 1, -1, -1):
            W, b, act, drop_after = layers[i]
            xin, y, pre = acts[i]
            last = i == n - 1
            if last:
                # g is the gradient w.r.t. the layer output after residual; undo dropout / activation
                if final_drop and p > 0:
                    g = out.take_branch_grad(g)
                if act != k.ACT_NONE and not last_act_external:
                    raise NotImplementedError("activation on the last layer of a chain needs last_act_external")
            # g is now the gradient w.r.t. the pre-activation output of layer i
            if W.g is not None:
                # the last layer's g doubles as the residual's gradient accumulator unless dropout made a fresh tensor
                tape.linear_wgrad(g, xin, W, b, owned=not (last and res is not None and res.needs_grad and not (final_drop and p > 0)))
            if i == 0:
                if x.needs_grad:
                    if in_relu_mask:  # x is a ReLU output whose producer wants d/d(pre-ReLU): mask in the epilogue
                        x.grad = ops.linear_dgrad(g, W.w, res=x.grad, act=k.ACT_MASK_POS, aux=x.data)
                    else:
                        x.grad = ops.linear_dgrad(g, W.w, res=x.grad)
            else:
                pW, pb, pact, pdrop = layers[i - 1]
                _, py, ppre = acts[i - 1]
                alpha = 1.0 / (1.0 - p) if (pdrop and p > 0) else 1.0
                if pact == k.ACT_RELU:
                    g = ops.linear_dgrad(g, W.w, act=k.ACT_MASK_POS, aux=py, alpha=alpha)
                elif pact == k.ACT_GELU:
                    if pdrop and p > 0:
                        raise NotImplementedError("dropout after GELU")
                    g = ops.linear_dgrad(g, W.w, act=k.ACT_GELU_BWD, aux=ppre)
                elif pact == k.ACT_NONE:
                    g = ops.linear_dgrad(g, W.w)
                else:
                    raise NotImplementedError

    tape.record(bwd)
    return out


def layernorm(tape, x, gamma, beta, eps, y=None, add=None):
    """LayerNorm; with `add` (bf16 [rows, D], a constant of the backward pass) the same launch also writes out.plus = y + add."""
    rows, D = x.data.shape
    if y is None:
        y = torch.empty_like(x.data)
    mean = torch.empty(rows, dtype=torch.float32, device=y.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=y.device)
    y2 = torch.empty_like(x.data) if add is not None else None
    k.layernorm_fwd(x.data, gamma.f32, beta.f32, eps, y, mean, rstd, add=add, y2=y2)
    out = Var(y)
    out.plus = y2

    def bwd():
        g = out.take_grad()
        if g is None:
            return
        dx = torch.empty_like(g)
        if x.drop is not None and x.grad is None and x.needs_grad:
            # x = residual + dropout(branch): emit the branch's masked gradient from the same pass over dx
            x.gdrop = torch.empty_like(g)
            k.layernorm_bwd(g, x.data, mean, rstd, gamma.f32, dx, gamma.g, beta.g if gamma.g is not None else None, dx_drop=x.gdrop,
                            drop_p=x.drop[0], seed=x.drop[1], defer=True)
        else:
            k.layernorm_bwd(g, x.data, mean, rstd, gamma.f32, dx, gamma.g, beta.g if gamma.g is not None else None, defer=True)
        if x.needs_grad:
            accumulate(x, dx)

    tape.record(bwd)
    return out


def add_const(tape, x, c):
    """x + c with c a constant bf16 tensor (positional encoding); c may be broadcast over rows."""
    y = torch.empty_like(x.data)
    k.add(x.data, c, y, b_period=c.numel())
    out = Var(y)

    def bwd():
        g = out.take_grad()
        if g is not None and x.needs_grad:
            accumulate(x, g)

    tape.record(bwd)
    return out


def add_vars(tape, a, b):
    y = torch.empty_like(a.data)
    k.add(a.data, b.data, y, b_period=b.data.numel())
    out = Var(y)
    rep = a.data.numel() // b.data.numel()

    def bwd():
        g = out.take_grad()
        if g is None:
            return
        if b.needs_grad:
            if rep == 1:
                accumulate(b, g)
            else:
                # b was broadcast over `rep` leading blocks: reduce (rare: query_pos over the batch)
                gb = g.view(rep, -1).float().sum(0).to(BF16).view(b.data.shape)
                accumulate(b, gb)
        if a.needs_grad:
            accumulate(a, g)

    tape.record(bwd)
    return out


def dropout(tape, x):
    p = tape.drop_p
    if p <= 0:
        return x
    seed = tape.next_seed()
    y = torch.empty_like(x.data)
    k.dropout(x.data, p, seed, y)
    out = Var(y)

    def bwd():
        g = out.take_grad()
        if g is None or not x.needs_grad:
            return
        gd = torch.empty_like(g)
        k.dropout(g, p, seed, gd)
        accumulate(x, gd)

    tape.record(bwd)
    return out


def attention(tape, q_in, k_in, v_in, Pq, Pk, Pv, Wo, bo, resid, key_pad, B, Sq, Sk, H, packed_qk=None):
    """resid + dropout(out_proj(softmax(q k^T / sqrt(dh) + mask) v)) -- nn.MultiheadAttention plus the
    residual/dropout that follows it (transformer.py:297-299, 370-400), also HF RobertaSelfAttention +
    RobertaSelfOutput.dense.  Pq/Pk/Pv = (weight ParamView, bias ParamView); packed_qk = the same for
    the stacked [2d, d] q/k block when q_in is k_in (one GEMM).  Returns Var [B*Sq, d]."""
    d = Wo.w.shape[0]
    dh = d // H
    scale = 1.0 / math.sqrt(dh)
    dev = q_in.data.device
    p = tape.drop_p
    fused = packed_qk is not None and q_in is k_in
    if fused:
        qk = ops.linear(q_in.data, packed_qk[0].w, packed_qk[1].f32)
        qb, kb = qk[:, :d], qk[:, d:]
    else:
        qb = ops.linear(q_in.data, Pq[0].w, Pq[1].f32)
        kb = ops.linear(k_in.data, Pk[0].w, Pk[1].f32)
    vb = ops.linear(v_in.data, Pv[0].w, Pv[1].f32)
    seed_p = tape.next_seed() if p > 0 else 0
    ctx = torch.empty(B * Sq, d, dtype=BF16, device=dev)
    fused_core = FUSED_ATTENTION and dh == 32
    if fused_core:
        # scores -> mask -> softmax -> dropout -> P V in one flash-style launch (csrc/attn2.hip): nothing score-shaped is written (22 MB of P and
        # 22 MB of dropout(P) per encoder layer at B = 8 otherwise), only (row maximum, 1 / row sum) per score row; any key count
        prob = prob_used = None
        ld = ops.round8(Sk)
        lse = torch.empty(B * H, Sq, 2, dtype=torch.float32, device=dev)
        k.attn2_fwd(qb, kb, vb, key_pad, B, H, Sq, Sk, dh, scale, p, seed_p, ctx, lse)
    else:
        s = ops.attn_scores(qb, kb, B, H, Sq, Sk, dh, scale)
        ld = s.shape[-1]
        prob = torch.empty_like(s)
        prob_used = torch.empty_like(s) if p > 0 else None
        k.softmax_fwd(s, key_pad, B, H, Sq, Sk, ld, prob, prob_used, p, seed_p)
        del s
        ops.attn_context(prob_used if prob_used is not None else prob, vb, B, H, Sq, Sk, dh, ctx)
    if prob_used is None:
        prob_used = prob
    seed_o = tape.next_seed() if p > 0 else 0
    z = ops.linear(ctx, Wo.w, bo.f32, res=resid.data, drop_where=1 if p > 0 else 0, drop_p=p, drop_seed=seed_o)
    out = Var(z)
    if p > 0:
        out.drop = (p, seed_o)

    def bwd():
        g = out.take_grad()
        if g is None:
            return
        if resid.needs_grad:
            accumulate(resid, g)
        go = out.take_branch_grad(g) if p > 0 else g
        if Wo.g is not None:
            tape.linear_wgrad(go, ctx, Wo, bo, owned=(go is not g or not resid.needs_grad))
        dctx = ops.linear_dgrad(go, Wo.w)
        if fused:
            dqk = torch.empty(B * Sq, 2 * d, dtype=BF16, device=dev)
            dq, dk = dqk[:, :d], dqk[:, d:]
        else:
            dq = torch.empty(B * Sq, d, dtype=BF16, device=dev)
            dk = torch.empty(B * Sk, d, dtype=BF16, device=dev)
        dv = torch.empty(B * Sk, d, dtype=BF16, device=dev)

        def sm_bwd(dp):
            ds = torch.empty_like(dp)
            k.softmax_bwd(prob, dp, B * H * Sq, Sk, ld, ds, p, seed_p)
            return ds

        if fused_core:
            # key-owning backward (csrc/attn2.hip): dK / dV written once, dQ directly or as one bf16 share per 128-key split
            splits = k.attn2_splits(Sk)
            part = torch.empty(splits, B * Sq, d, dtype=BF16, device=dev) if splits > 1 else None
            k.attn2_bwd(qb, kb, vb, ctx, dctx, lse, key_pad, B, H, Sq, Sk, dh, scale, p, seed_p, dq if splits == 1 else None, dk, dv, dq_part=part)
            if part is not None:
                dq.copy_(part.float().sum(0))
        else:
            ops.attn_backward(prob_used, scale, qb, kb, vb, dctx, B, H, Sq, Sk, dh, dq, dk, dv, sm_bwd)
        if fused:
            if packed_qk[0].g is not None:
                tape.linear_wgrad(dqk, q_in.data, packed_qk[0], packed_qk[1])
            if q_in.needs_grad:
                q_in.grad = ops.linear_dgrad(dqk, packed_qk[0].w, res=q_in.grad)
        else:
            if Pq[0].g is not None:
                tape.linear_wgrad(dq, q_in.data, Pq[0], Pq[1])
            if Pk[0].g is not None:
                tape.linear_wgrad(dk, k_in.data, Pk[0], Pk[1])
            if q_in.needs_grad:
                q_in.grad = ops.linear_dgrad(dq, Pq[0].w, res=q_in.grad)
            if k_in.needs_grad:
                k_in.grad = ops.linear_dgrad(dk, Pk[0].w, res=k_in.grad)
        if Pv[0].g is not None:
            tape.linear_wgrad(dv, v_in.data, Pv[0], Pv[1])
        if v_in.needs_grad:
            v_in.grad = ops.linear_dgrad(dv, Pv[0].w, res=v_in.grad)

    tape.record(bwd)
    return out


# ---- attention blocks with the packed in_proj applied in one launch (FUSED_BLOCKS) -------------------------------------
# nn.MultiheadAttention(q = k = x + e, v = x) (transformer.py:293-297, 366-372): ONE GEMM forms [q | k | v] -- output columns
# < 2d read their A rows from xe = x + e, the others from x (toist_gemm.a2) -- and one GEMM carries d[q | k | v] back to x.
# The gradient w.r.t. a trainable e (the decoder's query embedding) is not formed per layer: every block leaves its dq / dk in a
# column slice of one [rows, n] buffer (`e_sink`) and the caller multiplies that buffer once by the stacked projection weights.
def _attn_core(tape, qb, kb, vb, key_pad, B, Sq, Sk, H, ctx, p, seed_p):
    """attention core of the fused blocks (csrc/attn2.hip, head dim 32); returns core_bwd(dctx, dq, dk, dv)"""
    dh = qb.shape[1] // H
    scale = 1.0 / math.sqrt(dh)
    if not (FUSED_ATTENTION and dh == 32):
        raise NotImplementedError("fused attention blocks need head dim 32")
    lse = torch.empty(B * H, Sq, 2, dtype=torch.float32, device=qb.device)
    k.attn2_fwd(qb, kb, vb, key_pad, B, H, Sq, Sk, dh, scale, p, seed_p, ctx, lse)
    splits = k.attn2_splits(Sk)

    def core_bwd(dctx, dq, dk, dv):
        part = torch.empty(splits, B * Sq, H * dh, dtype=BF16, device=qb.device) if splits > 1 else None
        k.attn2_bwd(qb, kb, vb, ctx, dctx, lse, key_pad, B, H, Sq, Sk, dh, scale, p, seed_p, dq if splits == 1 else None, dk, dv, dq_part=part)
        if part is not None:
            dq.copy_(part.float().sum(0))

    return core_bwd


def _out_proj(tape, ctx, Wo, bo, resid, p):
    """resid + dropout(ctx Wo^T + bo); returns (Var, bwd_head) where bwd_head() -> gradient w.r.t. ctx (or None)."""
    seed_o = tape.next_seed() if p > 0 else 0
    z = ops.linear(ctx, Wo.w, bo.f32, res=resid.data, drop_where=1 if p > 0 else 0, drop_p=p, drop_seed=seed_o)
    out = Var(z)
    if p > 0:
        out.drop = (p, seed_o)

    def head():
        g = out.take_grad()
        if g is None:
            return None
        if resid.needs_grad:
            accumulate(resid, g)
        go = out.take_branch_grad(g) if p > 0 else g
        if Wo.g is not None:
            tape.linear_wgrad(go, ctx, Wo, bo, owned=(go is not g or not resid.needs_grad))
        return ops.linear_dgrad(go, Wo.w)

    return out, head


def self_attention_block(tape, x, xe, Win, bin_, Wo, bo, key_pad, B, S, H, e_sink=None):
    """x + dropout(out_proj(MHA(q = k = xe, v = x)));  xe = x + e as a bf16 tensor.  e_sink = (buffer [B*S, n] bf16, column):
    d q and d k are written at buffer[:, column : column + 2d] (and d v right after, buffer[:, column + 2d : column + 3d])."""
    d = Wo.w.shape[0]
    M = B * S
    dev = x.data.device
    p = tape.drop_p
    qkv = torch.empty(M, 3 * d, dtype=BF16, device=dev)
    k.gemm(M, 3 * d, d, k.A_ROWK, k.operand(xe, xe.stride(0)), k.B_ROWK, k.operand(Win.w, Win.w.stride(0)), qkv, 3 * d, shift=bin_.f32,
           a2=x.data, a2_from=2 * d, flops=2 * M * 3 * d * d)
    seed_p = tape.next_seed() if p > 0 else 0
    ctx = torch.empty(M, d, dtype=BF16, device=dev)
    core_bwd = _attn_core(tape, qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], key_pad, B, S, S, H, ctx, p, seed_p)
    out, head = _out_proj(tape, ctx, Wo, bo, x, p)

    def bwd():
        dctx = head()
        if dctx is None:
            return
        if e_sink is not None:
            buf, col = e_sink
            dqkv = buf[:, col:col + 3 * d]
        else:
            dqkv = torch.empty(M, 3 * d, dtype=BF16, device=dev)
        core_bwd(dctx, dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:])
        if Win.g is not None:
            tape.linear_wgrad(dqkv[:, :2 * d], xe, Win.rows(0, 2 * d), bin_.rows(0, 2 * d))
            tape.linear_wgrad(dqkv[:, 2 * d:], x.data, Win.rows(2 * d, 3 * d), bin_.rows(2 * d, 3 * d))
        if x.needs_grad:
            x.grad = ops.linear_dgrad(dqkv, Win.w, res=x.grad)      # d x = [dq | dk | dv] W_in  (+ residual gradient)

    tape.record(bwd)
    return out


def W0_has_grad(layers):
    return layers[0][0].g is not None


def cross_kv_projections(tape, mem, mem_e, layers):
    """k_i = mem_e Wk_i^T + bk_i, v_i = mem Wv_i^T + bv_i for every decoder layer i in ONE grouped launch (the inputs are the same
    for all layers, transformer.py:386-391): layers = [(Win_i, bin_i)] with the packed [3d, d] in_proj of cross_attn_image.
    Returns (kv [rows, L*2d] bf16 with [k_i | v_i] at columns 2d*i, dkv of the same shape for the blocks' backward to fill)."""
    L = len(layers)
    d = layers[0][0].w.shape[1]
    M = mem.data.shape[0]
    dev = mem.data.device
    kv = torch.empty(M, L * 2 * d, dtype=BF16, device=dev)
    dkv = torch.empty(M, L * 2 * d, dtype=BF16, device=dev) if (mem.needs_grad or W0_has_grad(layers)) else None
    W0, b0 = layers[0]
    rows = []
    for i, (W, b) in enumerate(layers):
        wkv = W.w[d:]
        boff = b.f32.data_ptr() - b0.f32.data_ptr()
        assert boff % 4 == 0
        rows.append([mem_e.data_ptr(), wkv.data_ptr(), i * 2 * d, 0, 0, boff // 4 + d])
    table = k.group_table(rows, dev)
    k.gemm(M, 2 * d, d, k.A_ROWK, k.operand(mem_e, mem_e.stride(0)), k.B_ROWK, k.operand(W0.w[d:], W0.w.stride(0)), kv, L * 2 * d, shift=b0.f32,
           batch=L, group=table, a2=mem.data, a2_from=d, flops=2 * M * 2 * d * d * L)

    def bwd():
        if dkv is None:
            return
        for i, (W, b) in enumerate(layers):
            if W.g is not None:
                tape.linear_wgrad(dkv[:, i * 2 * d:i * 2 * d + d], mem_e, W.rows(d, 2 * d), b.rows(d, 2 * d))
                tape.linear_wgrad(dkv[:, i * 2 * d + d:(i + 1) * 2 * d], mem.data, W.rows(2 * d, 3 * d), b.rows(2 * d, 3 * d))
        if mem.needs_grad:
            wstack = torch.cat([W.w[d:] for W, _ in layers], dim=0)             # [L*2d, d]: d mem = [dk_0 | dv_0 | dk_1 | ...] W_stack
            mem.grad = ops.linear_dgrad(dkv, wstack, res=mem.grad)

    tape.record(bwd)     # recorded before the decoder layers: runs after all of them have written their slices of dkv
    return kv, dkv


def cross_attention_block(tape, t, te, Wq, bq, kv, dkv, col, Wo, bo, key_pad, B, Sq, Sk, H, e_sink=None):
    """t + dropout(out_proj(MHA(q = te, k, v precomputed)));  te = t + e (bf16 tensor); k, v = kv[:, col : col + d], kv[:, col + d : col + 2d]
    from cross_kv_projections, whose backward consumes the dk / dv this block writes into the same columns of dkv."""
    d = Wo.w.shape[0]
    M = B * Sq
    dev = t.data.device
    p = tape.drop_p
    qb = ops.linear(te, Wq.w, bq.f32)
    seed_p = tape.next_seed() if p > 0 else 0
    ctx = torch.empty(M, d, dtype=BF16, device=dev)
    core_bwd = _attn_core(tape, qb, kv[:, col:col + d], kv[:, col + d:col + 2 * d], key_pad, B, Sq, Sk, H, ctx, p, seed_p)
    out, head = _out_proj(tape, ctx, Wo, bo, t, p)

    def bwd():
        dctx = head()
        if dctx is None:
            return
        if e_sink is not None:
            buf, c2 = e_sink
            dq = buf[:, c2:c2 + d]
        else:
            dq = torch.empty(M, d, dtype=BF16, device=dev)
        core_bwd(dctx, dq, dkv[:, col:col + d], dkv[:, col + d:col + 2 * d])
        if Wq.g is not None:
            tape.linear_wgrad(dq, te, Wq, bq)
        if t.needs_grad:
            t.grad = ops.linear_dgrad(dq, Wq.w, res=t.grad)

    tape.record(bwd)
    return out


def packed_cast(view):
    """compute_copy transform that keeps a parameter's bf16 copy INSIDE a larger buffer (`view` = its rows there): several parameters
    -- RoBERTa's separate query / key / value matrices -- then form one contiguous GEMM operand, and the optimizer tail keeps
    refreshing each copy in place through its own pointer."""
    def make(m):
        view.copy_(m.detach())
        return view
    make.elementwise = True
    make.pinned = True
    return make


def text_attention_block(tape, x, proj, Wpacked, Wo, bo, key_pad, B, L, H):
    """x + dropout(out_proj(MHA(q = k = v = x))) for the text encoder (HF RobertaSelfAttention + RobertaSelfOutput.dense,
    transformer.py:129-130): the three projections are ONE GEMM on the packed [3D, D] weight copy (their biases are added when the
    attention kernel loads q / k / v), the whole head runs in one launch each way (csrc/attn_small.hip), one GEMM carries
    d[q | k | v] back to x.  proj = [(Wq, bq), (Wk, bk), (Wv, bv)] ParamViews; Wpacked = bf16 [3D, D] holding their copies."""
    D = Wo.w.shape[0]
    dh = D // H
    M = B * L
    dev = x.data.device
    p = tape.drop_p
    scale = 1.0 / math.sqrt(dh)
    qkv = ops.linear(x.data, Wpacked)
    seed_p = tape.next_seed() if p > 0 else 0
    ctx = torch.empty(M, D, dtype=BF16, device=dev)
    stats = torch.empty(B * H, L, 2, dtype=torch.float32, device=dev)
    bias = [b.f32 for _, b in proj]
    k.attn_small_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], key_pad, B, H, L, dh, scale, p, seed_p, ctx, stats, *bias)
    out, head = _out_proj(tape, ctx, Wo, bo, x, p)

    def bwd():
        dctx = head()
        if dctx is None:
            return
        dqkv = torch.empty(M, 3 * D, dtype=BF16, device=dev)
        k.attn_small_bwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], key_pad, B, H, L, dh, scale, p, seed_p, stats, dctx,
                         dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:], *bias)
        for i, (W, b) in enumerate(proj):
            if W.g is not None:
                tape.linear_wgrad(dqkv[:, i * D:(i + 1) * D], x.data, W, b)
        if x.needs_grad:
            x.grad = ops.linear_dgrad(dqkv, Wpacked, res=x.grad)

    tape.record(bwd)
    return out


# ------------------------------------------------------------------------------------------ ResNet blocks
def bottleneck(tape, x, W, bn, stride, has_down, train):
    """torchvision Bottleneck (v1.5) on NHWC bf16 with FrozenBatchNorm folded: conv weights `W[name].w`
    are already multiplied by the BN scale, `bn[name]` = (scale, shift) fp32.  The gradient this block
    leaves in x.grad is already masked by (x > 0), i.e. it is the gradient w.r.t. the pre-ReLU sum of
    the producing block (every block input is a ReLU output)."""
    w1, w2, w3 = krsc(W["conv1"].w), krsc(W["conv2"].w), krsc(W["conv3"].w)
    s1, t1 = bn["bn1"]
    s2, t2 = bn["bn2"]
    s3, t3 = bn["bn3"]
    a1 = ops.conv2d(x.data, w1, shift=t1, act=k.ACT_RELU)
    a2 = ops.conv2d(a1, w2, stride=stride, pad=1, shift=t2, act=k.ACT_RELU)
    if has_down:
        wd = krsc(W["down"].w)
        sd, td = bn["down"]
        idn = ops.conv2d(x.data, wd, stride=stride, shift=td)
    else:
        idn = x.data
    y = ops.conv2d(a2, w3, shift=t3, res=idn, act=k.ACT_RELU)
    if has_down:
        del idn
    out = Var(y)
    if not train:
        return out
    H, Wd = x.data.shape[1], x.data.shape[2]

    def bwd():
        g3 = out.take_grad()  # w.r.t. the pre-ReLU sum (masked by the consumer)
        if g3 is None:
            return
        tape.conv_wgrad(g3, a2, w3.shape, krsc(W["conv3"].g), s3, view=W["conv3"])
        g2 = ops.conv2d_dgrad(g3, w3, a2.shape[1:3], act=k.ACT_MASK_POS, aux=a2)
        tape.conv_wgrad(g2, a1, w2.shape, krsc(W["conv2"].g), s2, stride=stride, pad=1, view=W["conv2"])
        g1 = ops.conv2d_dgrad(g2, w2, (H, Wd), stride=stride, pad=1, act=k.ACT_MASK_POS, aux=a1)
        tape.conv_wgrad(g1, x.data, w1.shape, krsc(W["conv1"].g), s1, view=W["conv1"])
        if has_down:
            tape.conv_wgrad(g3, x.data, wd.shape, krsc(W["down"].g), sd, stride=stride, view=W["down"])
        if not x.needs_grad:
            return
        prev = x.take_grad()
        if has_down:
            if stride == 1:
                gx = ops.conv2d_dgrad(g3, wd, (H, Wd), res=prev)
            else:
                gx = prev if prev is not None else torch.zeros_like(x.data)
                ops.conv2d_dgrad(g3, wd, (H, Wd), stride=stride, out=gx, res=gx if prev is not None else None)
        else:
            gx = g3
            if prev is not None:
                k.add(prev, g3, prev)
                gx = prev
        x.grad = ops.conv2d_dgrad(g1, w1, (H, Wd), res=gx, act=k.ACT_MASK_POS, aux=x.data)

    tape.record(bwd)
    return out

