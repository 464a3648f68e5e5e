"""Optimizer tail of the TOIST training step on MI355X.

The reference does three library sweeps over ~185 M parameters every step (engine.py:87-101):
`torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm)`, `optimizer.step()` with
`torch.optim.AdamW` over the three parameter groups of main.py:351-392, then `update_ema`
(util/optim.py:9-26).  `FusedClipAdamWEMA` does the same arithmetic in three HIP launches
(csrc/optim.hip) that read every gradient twice and every other tensor once, and it rewrites the bf16
compute copies of the weights in the same pass (engine.compute_copy), so no cast kernels run in the
next forward.  All state that changes per step (step count, clip coefficient, bias corrections) lives
on the device: the tail can be captured in a hipGraph; learning rates are re-read from a device table.

There is no CPU path: tensors must live on the GPU and the HIP library must be loadable.
"""
import numpy as np
import torch

from . import engine
from . import kernels as k
from .knobs import knob

LATE_BLOCKS = knob("TOIST_LATE_BLOCKS", 0)     # workgroups of a late group's update launch (0 = one per 8192-element chunk)

_TENSOR_DT = np.dtype([("p", "<i8"), ("m", "<i8"), ("v", "<i8"), ("ema", "<i8"), ("w", "<i8"), ("row_scale", "<i8"),
                       ("numel", "<i8"), ("row_len", "<i4"), ("group", "<i4")])
assert _TENSOR_DT.itemsize == 64  # toist_opt_tensor


def ema_pairs(model, model_ema):
    """(source, ema) tensor pairs for every floating-point state_dict entry, as update_ema visits them
    (util/optim.py:23-26; integer buffers carry no average)."""
    if hasattr(model, "module"):
        model = model.module
    msd = model.state_dict()
    return [(msd[name], ev) for name, ev in model_ema.state_dict().items() if ev.is_floating_point()]


def _dense(t):
    return t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last))


import weakref as _weakref

# Live tails of this process.  A step bumps the GLOBAL weight epoch (engine.WEIGHT_EPOCH: the masters changed behind torch's version counters),
# which also makes the compute copies of every OTHER model look stale -- with two models and two tails (the distillation step: teacher +
# student) each step() invalidated the other model's ~260 copies, and the next forward re-made them one torch launch at a time (524 of the
# 883 torch operators of an eager configs[4] step).  A tail therefore carries the copies of the other tails over its own bump when they were
# valid before it and none of their masters is a parameter of this tail.
_LIVE_TAILS = _weakref.WeakSet()
CARRY_OTHER_TAILS = True      # False: A/B switch (every step() invalidates the other tails' copies, as before)


class FusedClipAdamWEMA:
    """clip_grad_norm_ + AdamW + EMA (+ bf16 compute-copy refresh) in one multi-tensor pass.

    param_groups: like torch.optim.AdamW -- list of {"params": [...], "lr": ..., "weight_decay": ...}; the
    group dicts are kept in `self.param_groups` so `adjust_learning_rate` (util/optim.py:29-90) can assign
    `group["lr"]` as it does for a torch optimizer.  ema: list of (source, ema) pairs or None.
    max_norm <= 0 disables clipping (engine.py:89 `if max_norm > 0`).

    A parameter group with "late": True (the text encoder: 125 M of the 185 M parameters, 67 % of the tail's bytes) is updated LATE:
    step() still sums its gradients into the clipping norm, but its AdamW + EMA + bf16-refresh launch is not issued; the next forward pass
    issues it at the head of the text branch (engine.TEXT_PRELUDE, on the text stream), where it runs beside the ResNet forward, which
    neither reads nor writes those tensors -- the only consumer of the late parameters is the text encoder, which follows on the same
    stream.  The arithmetic is unchanged (same gradients, same clip coefficient, same step count: the device state of step N is not
    touched before step N + 1's finish_norm): parameters are bit-identical to the serial tail (tests/test_gpu_optim.py).  Anything that
    reads the late parameters outside a forward pass (state_dict, evaluation with a different module, the end of training) calls
    finish() first.

    A parameter group with "early_norm": True (again the text encoder) has the squares of its gradients summed as soon as the program that owns
    them has finished its backward pass (functions.AFTER_BACKWARD hook, on that program's stream -- the text branch, beside the ResNet
    backward): step() then reads only the other gradients (a third of the bytes) before finish_norm.  Same per-chunk partial sums, same
    total: bit-identical.  Used only while the gradient addresses are the ones of the previous step (REUSE_GRAD_BUFFERS / hipGraph replay);
    otherwise step() sums everything as before.

    defer_ema=True takes the moving average out of step(): it only needs the updated parameters, so the loop may run it as
    `ema_update()` on a side stream beside the next forward pass (12 of the tail's 38 bytes per parameter leave the critical path).
    The average is the same sequence of values; step() applies a still-pending update itself before it changes the parameters
    again, and `ema_update()` after the last step completes it."""

    def __init__(self, param_groups, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4, max_norm=0.1, ema=None, ema_decay=0.9998,
                 defer_ema=False):
        if isinstance(param_groups, (list, tuple)) and param_groups and torch.is_tensor(param_groups[0]):
            param_groups = [{"params": list(param_groups)}]
        self.param_groups = []
        for g in param_groups:
            g = dict(g)
            g["params"] = [p for p in g["params"]]
            g.setdefault("lr", lr)
            g.setdefault("weight_decay", weight_decay)
            self.param_groups.append(g)
        self.betas, self.eps, self.max_norm, self.ema_decay = (float(betas[0]), float(betas[1])), float(eps), float(max_norm), float(ema_decay)
        params = [p for g in self.param_groups for p in g["params"]]
        if not params:
            raise ValueError("FusedClipAdamWEMA: no parameters")
        self.device = params[0].device
        if self.device.type != "cuda":
            raise RuntimeError("FusedClipAdamWEMA needs GPU tensors: toist_amd has no CPU fallback")
        self.params = params
        self.exp_avg = [torch.zeros_like(p) for p in params]      # preserve_format: same physical layout as p
        self.exp_avg_sq = [torch.zeros_like(p) for p in params]
        ema = list(ema or [])
        by_ptr = {src.data_ptr(): e for src, e in ema}
        self._ema_of = [by_ptr.pop(p.data_ptr(), None) for p in params]
        # sources that are averaged but not optimised: frozen parameters and floating-point buffers
        self._ema_only = [(src, e) for src, e in ema if src.data_ptr() in by_ptr]
        self._group_of = [gi for gi, g in enumerate(self.param_groups) for _ in g["params"]]
        self._late = [bool(g.get("late", False)) for g in self.param_groups for _ in g["params"]]
        self._early = [bool(g.get("early_norm", False)) and not bool(g.get("late", False)) for g in self.param_groups for _ in g["params"]]
        self._param_ids = frozenset(id(p) for p in params)
        _LIVE_TAILS.add(self)
        self._early_ids = frozenset(id(p) for p, e in zip(params, self._early) if e)
        self._early_idx = [i for i, e in enumerate(self._early) if e]
        self._early_done = False
        self._e_lo = self._e_hi = 0
        if self._early_ids:
            import weakref
            from . import functions
            ref = weakref.ref(self)

            def hook(prog_params, _ref=ref):
                o = _ref()
                if o is not None:
                    o._maybe_norm_early(prog_params)

            functions.AFTER_BACKWARD[:] = [h for h in functions.AFTER_BACKWARD if getattr(h, "_owner", lambda: None)() is not None]
            hook._owner = ref
            functions.AFTER_BACKWARD.append(hook)
        self._late_pending = False
        self._late_captured = False
        self._late_grads = None
        self._n_now = self._n_late = 0
        if any(self._late):
            if defer_ema:
                raise ValueError("FusedClipAdamWEMA: late groups and defer_ema are alternatives")
            engine.register_text_prelude(self, [p for p, l in zip(params, self._late) if l])
        for t in params + self.exp_avg + [e for e in self._ema_of if e is not None] + [x for pr in self._ema_only for x in pr]:
            if t.dtype != torch.float32 or not _dense(t) or t.device != self.device:
                raise TypeError("FusedClipAdamWEMA: tensors must be dense fp32 tensors on one GPU")
        for p, e in zip(params, self._ema_of):
            if e is not None and (e.shape != p.shape or e.stride() != p.stride()):
                raise ValueError("FusedClipAdamWEMA: an EMA tensor must share its source's shape and strides")
        self.state = torch.zeros(32, dtype=torch.uint8, device=self.device)          # toist_opt_state
        self._groups_host = None
        self._groups_dev = torch.zeros(len(self.param_groups), 2, dtype=torch.float32, device=self.device)
        self._chunk = k.opt_chunk_elems()
        n_t = len(params) + len(self._ema_only)
        # gradient-pointer table: two pinned host copies used alternately, each guarded by an event recorded behind its upload, so a host
        # that runs ahead of the GPU never rewrites a table whose asynchronous copy has not been consumed; the upload is skipped
        # altogether when no pointer changed (REUSE_GRAD_BUFFERS, hipGraph replay)
        self._grads_host = [torch.zeros(n_t, dtype=torch.int64).pin_memory() for _ in range(2)]
        self._grads_event = [None, None]
        self._grads_turn = 0
        self._grads_last = None
        self._grads_dev = torch.zeros(n_t, dtype=torch.int64, device=self.device)
        self._table = None
        self._chunks = self._partial = self._ema_table = self._ema_chunks = self._ema_grads = None
        self._retired = []
        self.defer_ema = bool(defer_ema) and (any(e is not None for e in self._ema_of) or bool(self._ema_only))
        self._ema_pending = False
        self._copy_gen = -1
        self._copies = []
        self.sync_hyperparams()

    # ---- host-side tables -----------------------------------------------------------------------------------
    def sync_hyperparams(self):
        """Upload lr / weight_decay of self.param_groups (call after changing them when the step is replayed
        from a hipGraph; step() does it by itself otherwise)."""
        cur = [(float(g["lr"]), float(g["weight_decay"])) for g in self.param_groups]
        if cur != self._groups_host:
            self._groups_host = cur
            self._groups_dev.copy_(torch.tensor(cur, dtype=torch.float32), non_blocking=False)

    def _build_table(self):
        rows = np.zeros(len(self.params) + len(self._ema_only), dtype=_TENSOR_DT)
        engine.prune_copies()
        self._copies = []
        for i, p in enumerate(self.params):
            r = rows[i]
            r["p"], r["m"], r["v"] = p.data_ptr(), self.exp_avg[i].data_ptr(), self.exp_avg_sq[i].data_ptr()
            r["ema"] = self._ema_of[i].data_ptr() if self._ema_of[i] is not None else 0
            r["numel"], r["group"], r["row_len"] = p.numel(), self._group_of[i], 1
            ent = engine.copy_of(p)
            if ent is not None and ent.elementwise and ent.w.numel() == p.numel():
                r["w"] = ent.w.data_ptr()
                if ent.row_scale is not None:
                    r["row_scale"] = ent.row_scale.data_ptr()
                    r["row_len"] = p.numel() // p.shape[0]
                self._copies.append((ent, p))
        for j, (src, e) in enumerate(self._ema_only):
            r = rows[len(self.params) + j]
            r["p"], r["ema"], r["numel"], r["row_len"] = src.data_ptr(), e.data_ptr(), src.numel(), 1

        def chunked(rr):
            numels = rr["numel"]
            nch = (numels + self._chunk - 1) // self._chunk
            tens = np.repeat(np.arange(len(rr), dtype=np.int32), nch)
            first = np.repeat(np.cumsum(nch) - nch, nch)
            idx = (np.arange(int(nch.sum()), dtype=np.int64) - first).astype(np.int32)
            ch = np.ascontiguousarray(np.stack([tens, idx], axis=1).astype(np.int32))
            return torch.from_numpy(rr.view(np.uint8).copy()).to(self.device), torch.from_numpy(ch).to(self.device), int(ch.shape[0])

        def keep(name, new):
            """A rebuilt table takes the ADDRESS of the one it replaces when the shape allows: captured hipGraphs hold these pointers (a
            rebuild is triggered by any new compute copy in the process, e.g. a second model's first forward pass; replacing the tensors
            freed memory that a graph captured earlier still read -- a write through a garbage row faulted in
            tests/test_gpu_captured_step.py).  A table of another shape retires the old one instead of freeing it."""
            cur = getattr(self, name, None)
            if cur is not None and cur.shape == new.shape and cur.dtype == new.dtype:
                cur.copy_(new)
                return
            if cur is not None:
                self._retired.append(cur)
            setattr(self, name, new)

        if self.defer_ema:
            # the average gets its own table (source, average, size only): no gradient, no moments, no compute copy
            erows = rows[rows["ema"] != 0].copy()
            erows["m"], erows["v"], erows["w"], erows["row_scale"] = 0, 0, 0, 0
            et, ec, self._n_ema_chunks = chunked(erows)
            keep("_ema_table", et)
            keep("_ema_chunks", ec)
            keep("_ema_grads", torch.zeros(len(erows), dtype=torch.int64, device=self.device))
            rows = rows[:len(self.params)].copy()
            rows["ema"] = 0
        table, chunks, self._n_chunks = chunked(rows)
        keep("_table", table)
        # launch order of the chunks: everything updated inside step() first, the late groups' chunks behind them -- sqnorm walks all,
        # adamw_ema is launched on the two runs separately
        late_rows = torch.tensor(self._late + [False] * len(self._ema_only), dtype=torch.bool, device=self.device)
        early_rows = torch.tensor(self._early + [False] * len(self._ema_only), dtype=torch.bool, device=self.device)
        is_late = late_rows[chunks[:, 0].long()]
        is_early = early_rows[chunks[:, 0].long()]
        # [updated in step(), norm in step()] [updated in step(), norm taken early] [late]
        keep("_chunks", torch.cat([chunks[~is_late & ~is_early], chunks[is_early], chunks[is_late]], dim=0).contiguous())
        self._n_late = int(is_late.sum())
        self._n_now = self._n_chunks - self._n_late
        self._e_hi = self._n_now
        self._e_lo = self._n_now - int(is_early.sum())
        self._late_copies = [(ent, p_) for ent, p_ in self._copies if any(p_ is q for q, l in zip(self.params, self._late) if l)]
        keep("_partial", torch.empty(self._n_chunks, dtype=torch.float32, device=self.device))
        self._copy_gen = engine.COPY_GEN

    # ---- torch.optim-like surface ---------------------------------------------------------------------------
    def zero_grad(self, set_to_none=True):
        # a pending late-group launch reads the gradient VALUES through the pointers kept in _late_grads: it must run before they are zeroed
        # in place (set_to_none=False) or their storage is handed back to the allocator
        if not set_to_none:
            self.flush_late()           # (set_to_none=True keeps the tensors alive through _late_grads: the launch may still wait for the next text branch)
        self._early_done = False        # partial sums of gradients that are going away
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    @torch.no_grad()
    def step(self):
        capturing = torch.cuda.is_current_stream_capturing()
        if self._late_pending and not self._late_captured:
            self.flush_late()       # nobody ran it at the head of a text branch: the late groups must be updated before their gradients are re-read
        self._late_captured = False
        if self._table is None or (self._copy_gen != engine.COPY_GEN and not capturing):
            self._build_table()
        if not capturing:
            self.sync_hyperparams()
        if self._ema_pending:
            self.ema_update()       # nobody ran it beside the forward pass: it must see the parameters before they change again
        turn = self._grads_turn
        if self._grads_event[turn] is not None and not capturing:
            self._grads_event[turn].synchronize()      # the copy that last read this host table has completed
        gh = self._grads_host[turn].numpy()
        for i, p in enumerate(self.params):
            g = p.grad
            if g is None:
                gh[i] = 0
                continue
            if g.dtype != torch.float32 or g.shape != p.shape or g.stride() != p.stride():
                raise TypeError("FusedClipAdamWEMA: every gradient must be fp32 with its parameter's shape and strides")
            gh[i] = g.data_ptr()
        if self._grads_last is None or capturing or not np.array_equal(gh, self._grads_last):
            self._grads_dev.copy_(self._grads_host[turn], non_blocking=True)
            if not capturing:
                ev = self._grads_event[turn] or torch.cuda.Event()
                ev.record()
                self._grads_event[turn] = ev
            self._grads_last = gh.copy()
            self._grads_turn = turn ^ 1
        if self._early_done and self._e_hi > self._e_lo:
            # partial[e_lo:e_hi] were written by norm_early() of this step (the text branch, joined before the tail)
            if self._e_lo:
                k.opt_sqnorm(self._table, self._grads_dev, self._chunks, self._e_lo, self._partial)
            if self._n_chunks > self._e_hi:
                k.opt_sqnorm(self._table, self._grads_dev, self._chunks[self._e_hi:], self._n_chunks - self._e_hi, self._partial[self._e_hi:])
        else:
            k.opt_sqnorm(self._table, self._grads_dev, self._chunks, self._n_chunks, self._partial)
        self._early_done = False
        k.opt_finish_norm(self._partial, self._n_chunks, self.max_norm, self.betas[0], self.betas[1], self.state)
        if self._n_now:
            k.opt_adamw_ema(self._table, self._grads_dev, self._chunks, self._n_now, self._groups_dev, self.state, self.betas[0],
                            self.betas[1], self.eps, self.ema_decay)
        self._ema_pending = self.defer_ema
        # the masters changed behind torch's version counters: every compute copy of THESE parameters is stale except the ones just rewritten
        self._bump_epoch()
        if self._n_late:
            self._late_pending = True
            self._late_grads = [p.grad for p, l in zip(self.params, self._late) if l]     # the pointers in the device table must stay valid

    def note_replayed_step(self):
        """Host-side bookkeeping of ONE replay of a hipGraph that holds this optimizer's step(): the replay rewrote the masters and the bf16 compute
        copies in the tail's table, but no Python ran -- every OTHER cached copy of a parameter (another program cache of the same module, an
        evaluation-time transform) must be seen as stale, exactly as after an eager step()."""
        self._bump_epoch()

    def _bump_epoch(self):
        old = engine.WEIGHT_EPOCH
        engine.bump_weight_epoch()
        late_ids = {id(p_) for _, p_ in self._late_copies} if self._n_late else ()
        for ent, p in self._copies:
            if p.grad is not None and id(p) not in late_ids:
                ent.epoch = engine.WEIGHT_EPOCH
        for other in list(_LIVE_TAILS) if CARRY_OTHER_TAILS else ():          # the other tails' copies: their masters did not move in this step
            if other is self:
                continue
            for ent, p in getattr(other, "_copies", ()):
                if ent.epoch == old and id(p) not in self._param_ids:
                    ent.epoch = engine.WEIGHT_EPOCH

    @torch.no_grad()
    def _maybe_norm_early(self, prog_params):
        """AFTER_BACKWARD hook: a program has attached its gradients.  If it owns the early-norm parameters and their gradients sit at the
        addresses the device table already holds (those of the previous step), sum their squares now, on the current stream."""
        # (a second backward pass before step() -- gradient accumulation -- simply sums again: the launch reads the accumulated gradients)
        if not self._early_ids or self._table is None or self._grads_last is None or self._e_hi <= self._e_lo:
            return
        if not any(id(p) in self._early_ids for p in prog_params):
            return
        # from here on the early parameters' gradients have just changed: sums taken by an earlier backward pass (a step() that was skipped)
        # are stale unless this call replaces them
        self._early_done = False
        if self._copy_gen != engine.COPY_GEN and not torch.cuda.is_current_stream_capturing():
            return                          # step() is about to rebuild the tables
        last = self._grads_last
        for i in self._early_idx:
            g = self.params[i].grad
            if (0 if g is None else g.data_ptr()) != int(last[i]):
                return                      # fresh gradient buffers: the table is uploaded in step(), which then sums everything
        k.opt_sqnorm(self._table, self._grads_dev, self._chunks[self._e_lo:], self._e_hi - self._e_lo, self._partial[self._e_lo:])
        self._early_done = True

    @torch.no_grad()
    def flush_late(self):
        """Issue the AdamW + EMA + bf16-refresh launch of the "late" groups for the last step() on the current stream (no-op when
        nothing is pending; inside a hipGraph capture it is ALWAYS recorded, since the captured step leaves one pending for every
        replay -- the update of the step before the capture is then applied by the first replay)."""
        capturing = torch.cuda.is_current_stream_capturing()
        if not self._n_late or not (self._late_pending or capturing) or (capturing and self._late_captured):
            return
        k.opt_adamw_ema(self._table, self._grads_dev, self._chunks[self._n_now:], self._n_late, self._groups_dev, self.state, self.betas[0],
                        self.betas[1], self.eps, self.ema_decay, max_blocks=LATE_BLOCKS)
        for ent, p in self._late_copies:
            ent.epoch = engine.WEIGHT_EPOCH
        if not capturing:
            self._late_pending = False
            self._late_grads = None
        else:
            self._late_captured = True      # recorded at the head of this capture's text branch: the captured step() must not record it again

    def finish(self):
        """complete every deferred piece of the last step (late groups, deferred EMA): call before reading parameters outside a forward pass"""
        self.flush_late()
        if self._ema_pending:
            self.ema_update()

    @torch.no_grad()
    def ema_update(self):
        """defer_ema: fold the parameters of the last step() into the moving average (on the current stream; the caller joins that
        stream before the next step()).  No-op when nothing is pending."""
        if not self._ema_pending:
            return
        if self._table is None:
            self._build_table()
        k.opt_adamw_ema(self._ema_table, self._ema_grads, self._ema_chunks, self._n_ema_chunks, self._groups_dev, self.state, self.betas[0],
                        self.betas[1], self.eps, self.ema_decay)
        self._ema_pending = False

    # ---- introspection / checkpointing ----------------------------------------------------------------------
    def device_state(self):
        """{'clip_coef', 'grad_norm', 'bias1', 'bias2_sqrt', 'step'} read back from the device (synchronises)."""
        raw = self.state.cpu().numpy()
        f = raw[:16].view(np.float32)
        return {"clip_coef": float(f[0]), "grad_norm": float(f[1]), "bias1": float(f[2]), "bias2_sqrt": float(f[3]),
                "step": int(raw[16:20].view(np.int32)[0])}

    def state_dict(self):
        """torch.optim.AdamW's layout ({"state": {i: {"step", "exp_avg", "exp_avg_sq"}}, "param_groups": [{..., "params": [i, ...]}]}),
        so the 'optimizer' entry of a reference checkpoint (main.py:511-513, 645) resumes here and vice versa.  The step count is one
        device counter shared by every parameter (all parameters of the hot path receive a gradient every step)."""
        self.finish()
        step = self.device_state()["step"]
        state = {i: {"step": torch.tensor(float(step)), "exp_avg": self.exp_avg[i].clone(), "exp_avg_sq": self.exp_avg_sq[i].clone()}
                 for i in range(len(self.params))} if step > 0 else {}
        groups, base = [], 0
        for g in self.param_groups:
            d = {kk: vv for kk, vv in g.items() if kk != "params"}
            d.update(betas=self.betas, eps=self.eps, amsgrad=False, maximize=False)
            d["params"] = list(range(base, base + len(g["params"])))
            base += len(g["params"])
            groups.append(d)
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        self.finish()       # a pending late update belongs to the OLD moments
        if "state" not in sd and "exp_avg" in sd:      # round-1 layout of this class
            sd = {"state": {i: {"step": sd["step"], "exp_avg": a, "exp_avg_sq": b} for i, (a, b) in enumerate(zip(sd["exp_avg"], sd["exp_avg_sq"]))},
                  "param_groups": sd.get("param_groups", [])}
        steps = []
        for i, st in sd["state"].items():
            i = int(i)
            self.exp_avg[i].copy_(st["exp_avg"].reshape(self.exp_avg[i].shape) if st["exp_avg"].shape != self.exp_avg[i].shape else st["exp_avg"])
            self.exp_avg_sq[i].copy_(st["exp_avg_sq"].reshape(self.exp_avg_sq[i].shape) if st["exp_avg_sq"].shape != self.exp_avg_sq[i].shape else st["exp_avg_sq"])
            steps.append(int(float(st["step"])))
        if steps and min(steps) != max(steps):
            raise ValueError("FusedClipAdamWEMA.load_state_dict: parameters with different step counts (the fused tail keeps one counter)")
        raw = np.zeros(32, dtype=np.uint8)
        raw[16:20] = np.array([steps[0] if steps else 0], dtype=np.int32).view(np.uint8)
        self.state.copy_(torch.from_numpy(raw))
        for g, src in zip(self.param_groups, sd.get("param_groups", [])):
            g.update({kk: vv for kk, vv in src.items() if kk in ("lr", "weight_decay", "initial_lr")})
        self.sync_hyperparams()
