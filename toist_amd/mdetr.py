"""MDETR model, set criterion and builder on the MI355X kernels.

Mirrors /root/reference/models/mdetr.py: MDETR (:315-462), SetCriterion (:465-1021; the non-list
branch used by configs 1-4), MLP (:1024-1036), build (:1039-1141) -- same constructor arguments,
forward signatures, output / loss dict keys and state_dict names, so engine.py / main.py style
drivers run unchanged.
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F
from torch import nn

from . import dist, engine, functions
from . import kernels as k
from .backbone import build_backbone, nearest_mask
from .matcher import StaticTargets, build_matcher
from .misc import NestedTensor
from .transformer import build_transformer
from .distill import ClusterCriterion, char_span_to_tokens, noun_token_features
from .matcher import TOKEN_MASK_WORDS

BF16 = torch.bfloat16


class MLP(nn.Module):
    """Parameter holder of the 3-layer box head (mdetr.py:1024-1036)."""

    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        h = [hidden_dim] * (num_layers - 1)
        self.layers = nn.ModuleList(nn.Linear(n, o) for n, o in zip([input_dim] + h, h + [output_dim]))


class MDETR(nn.Module):
    def __init__(self, backbone, transformer, num_classes, num_queries, aux_loss=False, contrastive_hdim=64,
                 contrastive_align_loss=False, cluster_num=16, args=None):
        super().__init__()
        self.args = args
        self.num_queries = num_queries
        self.transformer = transformer
        hidden_dim = transformer.d_model
        self.class_embed = nn.Linear(hidden_dim, num_classes + 1)
        self.bbox_embed = MLP(hidden_dim, hidden_dim, 4, 3)
        self.query_embed = nn.Embedding(num_queries, hidden_dim)
        self.input_proj = nn.Conv2d(backbone.num_channels, hidden_dim, kernel_size=1)
        self.backbone = backbone
        self.aux_loss = aux_loss
        self.contrastive_align_loss = contrastive_align_loss
        if contrastive_align_loss:
            self.contrastive_align_projection_image = nn.Linear(hidden_dim, contrastive_hdim)
            self.contrastive_align_projection_text = nn.Linear(hidden_dim, contrastive_hdim)
        self._cache_proj, self._cache_heads = {}, {}

    # ---- native pieces ------------------------------------------------------------------------------
    def _project(self, c5):
        """input_proj (1x1 conv 2048 -> d, mdetr.py:351,383) on NHWC bf16 C5 -> tokens [B*hw, d] bf16."""
        B, h, w, C = c5.shape
        named = OrderedDict(self.input_proj.named_parameters())

        def prog(tape, ps, x):
            W, b = ps["weight"], ps["bias"]
            d = W.w.shape[0]
            Wv = engine.ParamView(W.w.reshape(d, C), None if W.g is None else engine.krsc(W.g).reshape(d, C), None, fresh=W.fresh, parent=W)
            xin = engine.Var(x.data.view(B * h * w, C), needs_grad=x.needs_grad)

            def bwd():  # recorded first -> runs after the chain's backward
                g = xin.take_grad()
                if g is not None:
                    x.grad = g.view(B, h, w, C)

            tape.record(bwd)
            y = engine.linear_chain(tape, xin, [(Wv, b, k.ACT_NONE, False)], in_relu_mask=True)
            return [y], None

        (tok,) = functions.run_program(prog, named, [c5], cache=self._cache_proj, training=self.training, store_once=lambda n, t: t.dim() == 4)
        return tok

    def _text_tokens(self, memory_cache, B):
        """bf16 [B*Lt, d] text rows of the encoder output (memory_cache["text_memory"], batch-major)."""
        native = memory_cache.get("_native")
        lazy = getattr(memory_cache, "is_lazy", None)
        if native is not None and ((lazy is not None and lazy("img_memory") and lazy("text_memory")) or
                                   (native.get("img_memory_ref") is not None and native.get("img_memory_ref") is memory_cache.get("img_memory"))):
            mem, Lt = native["memory"], native["L"]     # nobody replaced (or even read) the fp32 copies: the encoder's own bf16 rows
            return mem.view(B, native["S"], -1)[:, native["S"] - Lt:, :].reshape(B * Lt, -1)
        tm = memory_cache["text_memory"]
        Lt = tm.shape[0]
        return tm.permute(1, 0, 2).to(BF16).reshape(B * Lt, -1)

    def _heads(self, stack, B, text_tok=None):
        """class_embed / bbox_embed (+ contrastive image and text projections) on all decoder layers at once
        (mdetr.py:420-433).  stack: [L, B*Q, d] bf16 -> logits [L,B,Q,K+1] f32, boxes [L,B,Q,4] f32
        (+ raw projections [L,B,Q,h] and [B*Lt,h] f32)."""
        L, BQ, d = stack.shape
        Q = BQ // B
        named = OrderedDict()
        named.update(("class_embed." + n, p) for n, p in self.class_embed.named_parameters())
        named.update(("bbox_embed." + n, p) for n, p in self.bbox_embed.named_parameters())
        if self.contrastive_align_loss:
            named.update(("cimg." + n, p) for n, p in self.contrastive_align_projection_image.named_parameters())
            named.update(("ctxt." + n, p) for n, p in self.contrastive_align_projection_text.named_parameters())
        want_proj = self.contrastive_align_loss

        def prog(tape, ps, hs, txt=None):
            x = engine.Var(hs.data.view(L * BQ, d), needs_grad=hs.needs_grad)

            def x_bwd():
                g = x.take_grad()
                if g is not None:
                    hs.grad = g.view(L, BQ, d)

            tape.record(x_bwd)
            outs = []
            # --- class logits (f32 out) ---
            logits = engine.linear_chain(tape, x, [(ps["class_embed.weight"], ps["class_embed.bias"], k.ACT_NONE, False)],
                                         out_dtype=torch.float32)

            def logits_bwd():
                if logits.grad is not None:
                    logits.grad = logits.grad.to(BF16)

            tape.record(logits_bwd)
            outs.append(logits)
            # --- box head: last layer padded from 4 to 8 outputs (16-byte rows), sigmoid in the epilogue ---
            W2, b2 = ps["bbox_embed.layers.2.weight"], ps["bbox_embed.layers.2.bias"]
            w2p = box_pad_w                   # [8, d] bf16, rows 4-7 zero; rows 0-3 ARE W2's compute copy (engine.packed_cast: refreshed in place)
            assert W2.w.data_ptr() == w2p.data_ptr()
            b2p = box_pad_b
            b2p[:4].copy_(b2.f32)
            need = W2.g is not None
            gpad = torch.zeros(8 * d + 8, dtype=torch.float32, device=hs.data.device) if need else None
            g2w = gpad[:8 * d].view(8, d) if need else None
            g2b = gpad[8 * d:] if need else None

            def pad_bwd():
                if need:
                    k.flush_reductions()  # g2w comes from a split-K weight gradient whose fold may still be queued
                    W2.g.add_(g2w[:4])
                    b2.g.add_(g2b[:4])

            tape.record(pad_bwd)
            boxes = engine.linear_chain(
                tape, x, [(ps["bbox_embed.layers.0.weight"], ps["bbox_embed.layers.0.bias"], k.ACT_RELU, False),
                          (ps["bbox_embed.layers.1.weight"], ps["bbox_embed.layers.1.bias"], k.ACT_RELU, False),
                          (engine.ParamView(w2p, g2w), engine.ParamView(None, g2b, b2p), k.ACT_SIGMOID, False)],
                out_dtype=torch.float32, last_act_external=True)

            def boxes_bwd():
                if boxes.grad is not None:
                    y = boxes.data
                    gb = torch.empty(y.shape, dtype=BF16, device=y.device)
                    torch.mul(boxes.grad, torch.addcmul(y, y, y, value=-1.0), out=gb)      # sigmoid': y (1 - y) = y - y^2; two launches
                    boxes.grad = gb

            tape.record(boxes_bwd)
            outs.append(boxes)
            if want_proj:
                proj = engine.linear_chain(tape, x, [(ps["cimg.weight"], ps["cimg.bias"], k.ACT_NONE, False)], out_dtype=torch.float32)

                def proj_bwd():
                    if proj.grad is not None:
                        proj.grad = proj.grad.to(BF16)

                tape.record(proj_bwd)
                outs.append(proj)
                ptxt = engine.linear_chain(tape, txt, [(ps["ctxt.weight"], ps["ctxt.bias"], k.ACT_NONE, False)], out_dtype=torch.float32)

                def ptxt_bwd():
                    if ptxt.grad is not None:
                        ptxt.grad = ptxt.grad.to(BF16)

                tape.record(ptxt_bwd)
                outs.append(ptxt)
            return outs, None

        # the box head's last layer runs padded from 4 to 8 outputs: its bf16 compute copy lives in rows 0-3 of a zero [8, d] buffer
        pads = self.__dict__.setdefault("_box_pad", {})
        key_dev = str(stack.device)
        if key_dev not in pads or pads[key_dev][2] != id(self):      # id: a deepcopy of the module gets its own buffers
            buf = torch.zeros(8, d, dtype=BF16, device=stack.device)
            pads[key_dev] = (buf, torch.zeros(8, dtype=torch.float32, device=stack.device), id(self), {"bbox_embed.layers.2.weight": engine.packed_cast(buf[:4])})
        box_pad_w, box_pad_b, _, transforms = pads[key_dev]
        once = {"class_embed.weight", "bbox_embed.layers.0.weight", "bbox_embed.layers.1.weight", "cimg.weight", "ctxt.weight"}   # not the padded last box layer
        res = functions.run_program(prog, named, [stack] + ([text_tok] if want_proj else []), cache=self._cache_heads, training=self.training,
                                    store_once=lambda n, t: n in once, transforms=transforms)
        K = res[0].shape[-1]
        logits = res[0].view(L, B, Q, K)
        boxes = res[1][:, :4].reshape(L, B, Q, 4)
        proj = res[2].view(L, B, Q, -1) if want_proj else None
        return logits, boxes, proj, (res[3] if want_proj else None)

    # ---- reference-compatible forward -------------------------------------------------------------------
    def forward(self, samples: NestedTensor, captions, encode_and_save=True, memory_cache=None):
        """Two-phase call of the reference (mdetr.py:359-462): encode -> memory_cache dict; decode -> outputs."""
        if encode_and_save:
            assert memory_cache is None
            if not isinstance(samples, NestedTensor):
                samples = NestedTensor.from_tensor_list(samples)
            return self.encode(samples, captions)
        assert memory_cache is not None
        return self.decode(memory_cache)

    def encode(self, samples, captions, levels=(4,)):
        body = self.backbone[0]
        # The text branch (RoBERTa + resizer) and the image branch (ResNet) are independent until the
        # cross-modal encoder: fork the text branch onto its own HIP stream so its small GEMMs fill the
        # tails of the backbone kernels (a parallel branch of the captured hipGraph).  torch.autograd
        # replays each node's backward on the stream of its forward, so the backward pass forks the same way.
        side = None
        cuts = {}
        cut_on = getattr(self, "split_backward", False) and torch.is_grad_enabled()
        pre_encoded = isinstance(captions, tuple) and len(captions) == 3 and torch.is_tensor(captions[0])
        fork = not pre_encoded and samples.tensors.is_cuda and engine.overlap_enabled()
        tail = captions[1].shape[0] if pre_encoded else 0           # caption tokens behind the image tokens of the cross-modal sequence
        if not pre_encoded and samples.tensors.is_cuda:
            from .transformer import EncodedText
            if fork:
                main = torch.cuda.current_stream()
                side = engine.side_stream(samples.tensors.device, "text")
                k.stamp("fwd.fork")
                side.wait_stream(main)
                functions.REJOIN = main
            try:
                with torch.cuda.stream(side if fork else torch.cuda.current_stream()):
                    k.stamp("fwd.text.start")
                    tokenized = self.transformer._tokenize(captions, samples.tensors.device)
                    flat, key_pad_text = self.transformer.encode_text(tokenized)
                    k.stamp("fwd.text.end")
            finally:
                functions.REJOIN = None
            if cut_on and flat.requires_grad:
                leaf = flat.detach().requires_grad_(True)
                cuts["text"] = ([flat], [leaf])
                flat = leaf
            captions = EncodedText(tokenized, flat, key_pad_text)
            tail = tokenized["input_ids"].shape[1]
        stage_cuts = [] if cut_on else None          # data-parallel: the ResNet body runs as three programs with cuts between them
        k.stamp("fwd.backbone.start")
        feats = body.forward_native(samples.tensors, levels, premasked=(len(levels) - 1,), stage_cuts=stage_cuts)
        k.stamp("fwd.backbone.end")
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
        k.stamp("fwd.join")
        # Data-parallel jobs may cut the autograd graph at the outputs of the backbone and of the text encoder
        # (toist_amd.parallel.enable_backward_cuts): loss.backward() then stops there and the two long leaf programs are
        # run by backward_cut(memory_cache, name), so each gradient all-reduce can start as soon as its segment is done
        # and run underneath the remaining backward work.
        if cut_on and any(f.requires_grad for f in feats):
            leaves = [f.detach().requires_grad_(f.requires_grad) for f in feats]
            cuts["backbone"] = (list(feats), leaves)
            feats = leaves
            # backward order: "backbone" (layer4, or the whole body when it ran as one program), then layer3, then stem .. layer2
            for name, y, leaf in stage_cuts or ():      # "backbone.layer2", "backbone.layer3": only the stages that produced a differentiable output
                cuts[name] = ([y], [leaf])
        c5 = feats[-1]
        B, h, w, _ = c5.shape
        mask = nearest_mask(samples.mask, (h, w))
        pos_tok = self.backbone[1].tokens(mask, tail=tail) if hasattr(self.backbone[1], "tokens") else \
            self.backbone[1](NestedTensor(c5.permute(0, 3, 1, 2), mask)).flatten(2).permute(0, 2, 1).to(BF16).contiguous()
        tok = self._project(c5).view(B, h * w, -1)
        mc = self.transformer.encode_native(tok, pos_tok, mask.flatten(1), self.query_embed.weight, captions)
        mc["_native"]["features"] = feats
        mc["_native"]["cuts"] = cuts
        mc["_native"]["feat_mask"] = mask
        mc["_native"]["src_proj"] = tok
        return mc

    def decode(self, memory_cache):
        native = memory_cache.get("_native")
        key = "img_memory_mod" if (self.args is not None and getattr(self.args, "cluster", False)) else "img_memory"
        lazy = getattr(memory_cache, "is_lazy", None)
        if native is not None and lazy is not None and lazy(key):
            # the fp32 API copies were never read, let alone replaced: decode straight from the encoder's bf16 buffers
            stack = self.transformer.decode_native(None, None, None, native["query_embed"], native=native)
        else:
            stack = self.transformer.decode_native(memory_cache[key], memory_cache["pos_embed"], memory_cache["mask"],
                                                   memory_cache["query_embed"], native=native)
        B = memory_cache["mask"].shape[0]
        text_tok = self._text_tokens(memory_cache, B) if self.contrastive_align_loss else None
        logits, boxes, proj, proj_text = self._heads(stack, B, text_tok)
        out = {"pred_logits": logits[-1], "pred_boxes": boxes[-1]}
        proj_tokens = None
        if self.contrastive_align_loss:
            # F.normalize(Linear(hs)) / F.normalize(Linear(text_memory).transpose(0, 1)) of mdetr.py:429-433: both projections are
            # GEMMs of the heads program, the normalisation is the l2norm kernel of csrc/contrastive.hip (fwd + bwd)
            proj_q = l2_normalize(proj)
            proj_tokens = l2_normalize(proj_text.view(B, -1, proj_text.shape[-1]))
            out.update({"proj_queries": proj_q[-1], "proj_tokens": proj_tokens, "tokenized": memory_cache["tokenized"]})
        if self.aux_loss:
            aux = []
            for i in range(logits.shape[0] - 1):
                a = {"pred_logits": logits[i], "pred_boxes": boxes[i]}
                if self.contrastive_align_loss:
                    a.update({"proj_queries": proj_q[i], "proj_tokens": proj_tokens, "tokenized": memory_cache["tokenized"]})
                aux.append(a)
            out["aux_outputs"] = aux
        out["_stacked"] = {"pred_logits": logits, "pred_boxes": boxes, "hs": stack}
        if self.contrastive_align_loss:
            out["_stacked"]["proj_queries"] = proj_q
        return out


class SetCriterion(nn.Module):
    """Set-prediction loss (mdetr.py:465-1021, non-list branch): Hungarian matching of every decoder
    layer in one device launch, then labels / boxes / cardinality (+ contrastive_align, + masks)."""

    def __init__(self, args, num_classes, matcher, eos_coef, losses, temperature, contrastive_hdim=64, task_count=14):
        super().__init__()
        self.args = args
        self.num_classes = num_classes
        self.matcher = matcher
        self.eos_coef = eos_coef
        self.losses = losses
        self.temperature = temperature
        self.last_match = None
        self._maps, self._nb, self._nb_reduced = {}, {}, {}
        self._tokmask = None       # (member mask tensors, concatenated [sum T, 4] int64): spans of the last batch, uploaded once
        self._pending_status = []  # (pinned host copy, event) of matcher status words not yet looked at

    # -- helpers ------------------------------------------------------------------------------------------
    def _slot_maps(self, match, device):
        """Per matched slot: image index and offset of the image's first target, plus the per-image target
        counts -- host arithmetic on shapes, cached per batch signature (no H2D copy in steady state)."""
        key = (tuple(match.sizes), tuple(match.counts), str(device))
        ent = self._maps.get(key)
        if ent is None:
            if len(self._maps) > 64:
                self._maps.clear()
            b_idx, t_base, acc = [], [], 0
            for b, (cnt, size) in enumerate(zip(match.counts, match.sizes)):
                b_idx += [b] * cnt
                t_base += [acc] * cnt
                acc += size
            ent = (torch.tensor(b_idx, dtype=torch.int64, device=device), torch.tensor(t_base, dtype=torch.int64, device=device),
                   torch.tensor(match.sizes, dtype=torch.float32, device=device))
            self._maps[key] = ent
        return ent

    def _num_boxes(self, targets, device):
        total = float(sum(len(t["labels"]) for t in targets))
        key = (total, str(device))
        n = self._nb.get(key)
        if n is None:
            n = torch.as_tensor([total], dtype=torch.float, device=device)
            self._nb = {key: n}
        if not dist.is_dist_avail_and_initialized():
            return torch.clamp(n, min=1)[0]
        if n.is_cuda and torch.cuda.is_current_stream_capturing():
            # no collective inside a captured hipGraph: the graph's inputs are static, so the world sum of the
            # preceding eager (warm-up) step for the same local count is the value to bake in
            red = self._nb_reduced.get(key)
            if red is None:
                raise RuntimeError("SetCriterion: run one eager step before capturing a hipGraph in a distributed job")
            return red
        red = n.clone()
        torch.distributed.all_reduce(red)   # mdetr.py:997-1001 of the reference
        red = torch.clamp(red / dist.get_world_size(), min=1)[0]
        self._nb_reduced = {key: red}
        return red

    def _stack(self, outputs):
        """[L,B,Q,K] logits and [L,B,Q,4] boxes of the layers the reference would visit: the main layer, preceded by the
        auxiliary ones only when the model emitted 'aux_outputs' (mdetr.py:1009; --no_aux_loss recipes match one layer)."""
        st = outputs.get("_stacked")
        if st is not None:
            lg, bx = st["pred_logits"], st["pred_boxes"]
            return (lg, bx) if "aux_outputs" in outputs else (lg[-1:], bx[-1:])
        layers = list(outputs.get("aux_outputs", [])) + [outputs]
        return torch.stack([o["pred_logits"] for o in layers]), torch.stack([o["pred_boxes"] for o in layers])

    def _stack_proj(self, outputs, L):
        st = outputs.get("_stacked")
        if st is not None and "proj_queries" in st:
            return st["proj_queries"][-L:]
        layers = list(outputs.get("aux_outputs", [])) + [outputs]
        return torch.stack([o["proj_queries"] for o in layers])[-L:]

    # -- matcher status (SciPy raises ValueError on NaN / -inf costs at matcher.py:85) ----------------------------------
    def _note_status(self, match):
        """The losses of a layer whose cost block was invalid are NaN already (criterion kernel); additionally stage the status
        words in pinned host memory without waiting, so the NEXT call (or check_status()) can raise the reference's
        ValueError -- no host sync on the step's critical path."""
        st = match.status
        if not st.is_cuda or torch.cuda.is_current_stream_capturing():
            return
        host = torch.empty(st.shape, dtype=st.dtype, pin_memory=True)
        host.copy_(st, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._pending_status.append((host, ev))

    def check_status(self, wait=True):
        """Raise ValueError (as scipy.optimize.linear_sum_assignment does inside the reference matcher) if a cost block of an
        earlier call held NaN / -inf.  wait=False only looks at copies that have already arrived."""
        keep = []
        for host, ev in self._pending_status:
            if not wait and not ev.query():
                keep.append((host, ev))
                continue
            ev.synchronize()
            if bool((host == 1).any()):
                self._pending_status = []
                raise ValueError("matrix contains invalid numeric entries")
            if bool((host == 2).any()):
                self._pending_status = []
                raise ValueError("cost matrix is infeasible")
        self._pending_status = keep

    def token_masks_host(self, targets, tokenized):
        """int64 [sum T, TOKEN_MASK_WORDS] host tensor of the targets' token-span bit masks (input of StaticTargets.load)."""
        rows = []
        for i, tgt in enumerate(targets):
            rows += self._span_bits(tgt, i, tokenized)
        return torch.tensor(rows, dtype=torch.int64).reshape(len(rows), TOKEN_MASK_WORDS)

    @staticmethod
    def _span_bits(tgt, i, tokenized):
        rows = []
        for t in range(int(tgt["boxes"].shape[0])):
            if "token_spans" in tgt:
                spans = tgt["token_spans"][t]
            else:
                spans = []
                for beg, end in tgt["tokens_positive" if "tokens_positive" in tgt else "tokens"][t]:
                    ft = char_span_to_tokens(tokenized, i, beg, end)
                    if ft is not None:
                        spans.append(ft)
            bits = 0
            for bp, ep in spans:
                if ep >= 64 * TOKEN_MASK_WORDS:
                    raise ValueError("contrastive_align: token position %d is beyond max_text_len = %d (mdetr.py:601-666)" % (ep, 64 * TOKEN_MASK_WORDS))
                for tkn in range(bp, ep + 1):
                    bits |= 1 << tkn
            words = [(bits >> (64 * w)) & ((1 << 64) - 1) for w in range(TOKEN_MASK_WORDS)]
            rows.append([w - (1 << 64) if w >= (1 << 63) else w for w in words])         # two's complement: the tensor is int64
        return rows

    def _token_masks(self, targets, tokenized, device):
        """int64 [sum T, TOKEN_MASK_WORDS] token bit masks of every target's positive spans (the host part of mdetr.py:614-643, done once per
        batch instead of once per layer and call: the spans do not depend on the assignment)."""
        parts = []
        for i, tgt in enumerate(targets):
            m = tgt.get("_tok_mask")
            if m is None or m.device != device:
                n = int(tgt["boxes"].shape[0])
                rows = self._span_bits(tgt, i, tokenized)
                m = torch.tensor(rows, dtype=torch.int64).reshape(n, TOKEN_MASK_WORDS).to(device)
                tgt["_tok_mask"] = m
            parts.append(m)
        ent = self._tokmask
        if ent is not None and len(ent[0]) == len(parts) and all(a is b for a, b in zip(ent[0], parts)):
            return ent[1]
        cat = torch.cat(parts) if parts else torch.zeros(0, TOKEN_MASK_WORDS, dtype=torch.int64, device=device)
        self._tokmask = (parts, cat)
        return cat

    # -- losses over all layers at once -------------------------------------------------------------------
    def _detection_losses(self, logits, boxes, match, targets, positive_map, num_boxes):
        """labels / boxes / cardinality for every decoder layer: one forward launch of the fused HIP
        criterion kernel (and one backward launch), see csrc/criterion.hip."""
        L = logits.shape[0]
        vals = _SetLossFn.apply(logits.float().contiguous(), boxes.float().contiguous(), match, positive_map.float().contiguous(),
                                num_boxes.reshape(1).float().contiguous(), float(self.eos_coef))
        self._note_status(match)
        out, index = LossDict(), {}
        for l in range(L):
            sfx = "" if l == L - 1 else f"_{l}"
            if "labels" in self.losses:
                out["loss_ce" + sfx], index["loss_ce" + sfx] = vals[l, 0], 4 * l
            if "boxes" in self.losses:
                out["loss_bbox" + sfx], index["loss_bbox" + sfx] = vals[l, 1], 4 * l + 1
                out["loss_giou" + sfx], index["loss_giou" + sfx] = vals[l, 2], 4 * l + 2
            if "cardinality" in self.losses:
                out["cardinality_error" + sfx] = vals[l, 3].detach()
        out.groups.append((vals, index))
        return out

    def _contrastive_align(self, outputs, match, targets, num_boxes, L):
        """loss_contrastive_align of the L visited layers (mdetr.py:601-666) -> LossDict: one launch of the device kernel
        (csrc/contrastive.hip) each way; the token spans of the targets are uploaded once per batch."""
        pq = self._stack_proj(outputs, L)
        pt = outputs["proj_tokens"]
        out, index = LossDict(), {}
        if match.tgt_boxes is None:      # no target in the batch: every term is masked out (mdetr.py:655,662)
            vals = (pq.sum() * 0 + pt.sum() * 0).expand(L)
        else:
            masks = self._token_masks(targets, outputs.get("tokenized"), pq.device)
            vals = _ContrastiveFn.apply(pq.float().contiguous(), pt.float().contiguous(), match, masks,
                                        num_boxes.reshape(1).float().contiguous(), float(self.temperature))
        for l in range(L):
            key = "loss_contrastive_align" + ("" if l == L - 1 else f"_{l}")
            out[key], index[key] = vals[l], l
        if match.tgt_boxes is not None:
            out.groups.append((vals, index))
        return out

    def forward(self, memory_cache, outputs, targets, positive_map, example_rel=None):
        if self._pending_status and not torch.cuda.is_current_stream_capturing():
            self.check_status(wait=False)   # an invalid cost block of an earlier call raises here, like SciPy inside the reference matcher
        if isinstance(outputs, list):
            if isinstance(targets[0], StaticTargets):
                return self._forward_pair_static(memory_cache, outputs, targets)
            return self._forward_pair(memory_cache, outputs, targets, positive_map)
        logits, boxes = self._stack(outputs)
        L = logits.shape[0]
        if isinstance(targets, StaticTargets):
            return self._forward_static(outputs, logits, boxes, targets, L)
        match = self.matcher.match_layers(logits.detach(), boxes.detach(), targets, positive_map)
        self.last_match = match
        num_boxes = self._num_boxes(targets, logits.device)
        losses = self._detection_losses(logits, boxes, match, targets, positive_map, num_boxes)
        if "contrastive_align" in self.losses:
            losses.merge(self._contrastive_align(outputs, match, targets, num_boxes, L))
        if "masks" in self.losses:
            from .segmentation import mask_losses
            losses.update(mask_losses(outputs, targets, match, L - 1, num_boxes))
        return losses


    def _forward_static(self, outputs, logits, boxes, st, L):
        """Detection losses on a StaticTargets image (matcher.StaticTargets): no shape, pointer or launch parameter depends on the
        batch's target counts -- the captured step is replayed for any batch after st.load(...).  labels / boxes / cardinality /
        contrastive_align / masks (StaticTargets(mask_hw=...) carries the ground-truth masks)."""
        match = self.matcher.match_layers_static(logits.detach(), boxes.detach(), st)
        self.last_match = match
        losses = self._detection_losses(logits, boxes, match, None, st.positive_map, st.num_boxes)
        if "contrastive_align" in self.losses:
            pq, pt = self._stack_proj(outputs, L), outputs["proj_tokens"]
            vals = _ContrastiveFn.apply(pq.float().contiguous(), pt.float().contiguous(), match, st.tok_mask, st.num_boxes, float(self.temperature))
            out, index = LossDict(), {}
            for l in range(L):
                key = "loss_contrastive_align" + ("" if l == L - 1 else f"_{l}")
                out[key], index[key] = vals[l], l
            out.groups.append((vals, index))
            losses.merge(out)
        if "masks" in self.losses:
            from .segmentation import mask_losses_static
            losses.update(mask_losses_static(outputs, st, match, L - 1, L))
        return losses

    # ---- distillation: [teacher (noun), student (pronoun)] -------------------------------------------------------
    def _forward_pair(self, memory_cache, outputs, targets, positive_map):
        """List branch of the reference (mdetr.py:887-987): every detection loss for both models under the prefixes
        noun_ / sth_, then the cross losses nsthl2 (main layer) and softkd (every layer)."""
        losses, sides = LossDict(), []
        for prefix, out, tgt, pm in zip(("noun", "sth"), outputs, targets, positive_map):
            logits, boxes = self._stack(out)
            L = logits.shape[0]
            match = self.matcher.match_layers(logits.detach(), boxes.detach(), tgt, pm)
            num_boxes = self._num_boxes(tgt, logits.device)
            side = self._detection_losses(logits, boxes, match, tgt, pm, num_boxes)
            if "contrastive_align" in self.losses:
                side.merge(self._contrastive_align(out, match, tgt, num_boxes, L))
            if "masks" in self.losses:
                from .segmentation import mask_losses
                side.update(mask_losses(out, tgt, match, L - 1, num_boxes))
            losses.merge(side, prefix + "_")
            sides.append((logits, boxes, match, L))
        self.last_match = sides[1][2]
        if getattr(self.args, "nsthl2_loss", False):
            losses["loss_nsthl2"] = self._loss_nsthl2(memory_cache, outputs, targets, sides[1][2])
        if getattr(self.args, "softkd_loss", False):
            L = sides[0][3]
            per_layer = self._loss_softkd(sides[0], sides[1])
            for l in range(L):
                losses["loss_softkd" + ("" if l == L - 1 else f"_{l}")] = per_layer[l]
            losses.groups.append((per_layer, {"loss_softkd" + ("" if l == L - 1 else f"_{l}"): l for l in range(L)}))
        return losses

    # ---- the same on StaticTargets (+ .distill tables): nothing depends on the batch's target counts, captions or tasks -> one hipGraph for any batch ----
    def _forward_pair_static(self, memory_cache, outputs, sts):
        """_forward_pair on two matcher.StaticTargets images (teacher, student), each carrying its distill.DistillTables: the step captured by
        harness.CapturedDistillStep.  The two sides must hold the same number of targets per image (softkd pairs the matched queries through
        their target, mdetr.py:543-599) -- as the reference's (noun, pronoun) pairs do."""
        losses, sides = LossDict(), []
        for prefix, out, st in zip(("noun", "sth"), outputs, sts):
            logits, boxes = self._stack(out)
            L = logits.shape[0]
            side = self._forward_static(out, logits, boxes, st, L)
            losses.merge(side, prefix + "_")
            sides.append((logits, boxes, self.last_match, L))
        self.last_match = sides[1][2]
        if getattr(self.args, "nsthl2_loss", False):
            feats = []
            for mc, st in zip(memory_cache, sts):
                text = mc["text_memory"].permute(1, 0, 2)
                feats.append((st.distill.W_span.to(text.dtype).unsqueeze(-1) * text).sum(1))
            mo = sts[1].match_off
            keep = (mo[1:] - mo[:-1]) > 0                                          # images in which the student's main layer matched a box
            per_image = ((feats[1] - feats[0].detach()) ** 2).mean(1)
            losses["loss_nsthl2"] = torch.where(keep, per_image, torch.zeros_like(per_image)).sum() / keep.sum().clamp(min=1)
        if getattr(self.args, "softkd_loss", False):
            L = sides[0][3]
            if engine.overlap_enabled() and getattr(self, "defer_pair_join", False):
                # (opt-in: the CALLER must wait for `losses.join` before it reads a softkd value -- harness.CapturedDistillStep sets defer_pair_join around its own call; any
                # other caller gets the whole dict on its own stream)
                # inside a captured step the softkd block -- cost matrices, 24 LSAP problems on 24 CUs for ~3.6 ms, the KL terms -- goes to a side stream: the
                # caller (harness.CapturedDistillStep) starts the TEACHER's backward pass beside it (the teacher is detached in softkd / nsthl2: its gradients come
                # from the noun_ losses alone) and joins `losses.join` before it forms the student's total
                main = torch.cuda.current_stream()
                side = engine.side_stream(sides[0][0].device, "softkd")
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    per_layer = self._loss_softkd_static(sides[0], sides[1], sts[0], sts[1])
                losses.join = side
            else:
                per_layer = self._loss_softkd_static(sides[0], sides[1], sts[0], sts[1])
            for l in range(L):
                losses["loss_softkd" + ("" if l == L - 1 else f"_{l}")] = per_layer[l]
            losses.groups.append((per_layer, {"loss_softkd" + ("" if l == L - 1 else f"_{l}"): l for l in range(L)}))
        return losses

    def _loss_softkd_static(self, noun, sth, st_n, st_s):
        """_loss_softkd with every index built on the device from the StaticTargets offsets: the pair tables have the fixed capacity st.cap (matched pairs)
        and Q per (layer, image) LSAP problem (unmatched pairs); the problem sizes Q - c_i are device vectors that toist_lsap reads; dead slots are
        masked.  Every image's row count c_i + (Q - c_i) is Q, so the batchmean / image-mean weights are the constant 1 / (Q B)."""
        from . import kernels as k
        from .matcher import check_lsap_status
        (lg_n, bx_n, m_n, _), (lg_s, bx_s, m_s, _) = noun, sth
        L, B, Q = lg_n.shape[0], lg_n.shape[1], lg_n.shape[2]
        dev, cap = lg_n.device, st_n.cap

        def make():
            p = torch.arange(L * cap, dtype=torch.int64, device=dev)
            s = torch.arange(L * B * Q, dtype=torch.int64, device=dev)
            prob = s // Q
            return (p, s % Q, prob // B, prob % B, torch.arange(L * B, dtype=torch.int64, device=dev) * (Q * Q), torch.arange(L * B, dtype=torch.int64, device=dev) * Q)
        p, slot_s, slot_l, slot_b, offsets, out_off = _cached(("softkd_static", L, B, Q, cap, str(dev)), make)
        mo = st_n.match_off.to(torch.int64)                                       # [B + 1] (the student's must be equal: same targets per image)
        mtot = mo[B]
        live = p < L * mtot
        l_of = torch.div(p, mtot.clamp(min=1), rounding_mode="floor").clamp(max=L - 1)
        j = p - l_of * mtot
        img = torch.bucketize(j, mo[1:], right=True).clamp(max=B - 1)

        def binarise(lg):
            pr = lg.float().softmax(-1)
            return torch.cat([pr[..., :-1].sum(-1, keepdim=True), pr[..., -1:]], dim=-1)                  # [L,B,Q,2]

        def unmatched_first(m):
            """[L,B,Q] query order with the unmatched queries first (ascending), the matched ones after."""
            free = torch.ones(L * B * Q + 1, dtype=torch.int8, device=dev)
            free.scatter_(0, torch.where(live, (l_of * B + img) * Q + m.src.reshape(-1)[:L * cap], torch.full_like(p, L * B * Q)), 0)
            return torch.sort(free[:L * B * Q].view(L, B, Q), dim=-1, descending=True, stable=True).indices

        p_n, p_s = binarise(lg_n).detach(), binarise(lg_s)
        ord_n, ord_s = unmatched_first(m_n), unmatched_first(m_s)
        take = lambda t, order: torch.gather(t, 2, order[..., None].expand(-1, -1, -1, t.shape[-1]))
        fp_n, fp_s = take(p_n, ord_n), take(p_s, ord_s)                                                 # unmatched first
        with torch.no_grad():
            fb_n, fb_s = take(bx_n.float(), ord_n), take(bx_s.float(), ord_s)
            cost_class = (fp_n[:, :, None, :, :] * (fp_n.log()[:, :, None, :, :] - fp_s.log()[:, :, :, None, :])).sum(-1)   # [L,B,S,T]
            cost = (torch.cdist(fb_s, fb_n, p=1) + cost_class - _paired_giou_matrix(fb_s, fb_n)).contiguous()
            n_b = (Q - (mo[1:] - mo[:-1])).to(torch.int32).repeat(L).contiguous()                       # [L*B] problem sizes, layer-major
            rows = torch.zeros(L * B * Q, dtype=torch.int64, device=dev)
            cols = torch.zeros(L * B * Q, dtype=torch.int64, device=dev)
            status = torch.zeros(L * B, dtype=torch.int32, device=dev)
            k.lsap(cost, offsets, n_b, n_b, L * B, Q, Q, Q * Q, out_off, rows, cols, status, ld=Q)
            valid = slot_s < n_b.to(torch.int64)[slot_l * B + slot_b]
        kl_fp = _kl_rows(fp_n[slot_l, slot_b, cols], fp_s[slot_l, slot_b, rows])                       # [L*B*Q]; dead slots read row 0 and are masked
        kl_fp = torch.where(valid, kl_fp, torch.zeros_like(kl_fp))

        def by_target(pr, m):
            """[L, cap, 2]: the probabilities of the query matched to target t of image i at position match_off[i] + t; dead rows (0.5, 0.5) on both sides: KL 0"""
            vals = pr.reshape(L * B * Q, 2)[torch.where(live, (l_of * B + img) * Q + m.src.reshape(-1)[:L * cap], torch.zeros_like(p))]
            dst = torch.where(live, l_of * cap + mo[img] + m.tgt.reshape(-1)[:L * cap], torch.full_like(p, L * cap))
            out = torch.full((L * cap + 1, 2), 0.5, device=dev, dtype=pr.dtype)
            return out.index_copy(0, dst, vals)[:L * cap].view(L, cap, 2)
        kl_tp = _kl_rows(by_target(p_n, m_n), by_target(p_s, m_s))                                      # [L, cap]
        per_layer = (kl_tp.sum(1) + kl_fp.view(L, B * Q).sum(1)) / float(Q * B)
        # the list path raises when the two sides of a pair hold different numbers of targets; here the counts live on the device: poison instead (NaN losses trip
        # the loop's finite-loss guard, engine.py:212-215) -- harness.CapturedDistillStep.pack() already refuses such a batch on the host
        same = (st_n.match_off == st_s.match_off).all()
        per_layer = torch.where(same, per_layer, torch.full_like(per_layer, float("nan")))
        check_lsap_status(status, defer=True)
        return per_layer

    def _loss_nsthl2(self, memory_cache, outputs, targets, match_sth):
        """mdetr.py:668-781: MSE between the student's and the (detached) teacher's mean noun-token text feature, over
        the images in which the student's main layer matched at least one box."""
        feats = []
        for mc, out, tgt in zip(memory_cache, outputs, targets):
            feats.append(noun_token_features(mc["text_memory"].permute(1, 0, 2), out.get("tokenized", mc.get("tokenized")), tgt))
        keep = [i for i, c in enumerate(match_sth.counts) if c > 0]
        if not keep:
            return torch.zeros((), device=feats[0].device)
        idx = _cached(("keep", tuple(keep), str(feats[0].device)), lambda: torch.as_tensor(keep, device=feats[0].device))
        per_image = ((feats[1][idx] - feats[0][idx].detach()) ** 2).mean(1)     # F.mse_loss per image
        return per_image.sum() / len(keep)

    def _loss_softkd(self, noun, sth):
        """mdetr.py:543-599 for every decoder layer at once -> [L]: KL(student || teacher) on the binarised (object,
        no-object) probabilities -- matched queries paired through their target, unmatched ones through an LSAP on a
        KL + L1 + GIoU cost (mdetr.py:520-541).  All layers x images are one launch of the device LSAP kernel on blocks
        of one [L,B,Q,Q] cost tensor; no per-image Python work, no host sync except the final status check."""
        from .matcher import check_lsap_status, lsap_blocks
        (lg_n, bx_n, m_n, _), (lg_s, bx_s, m_s, _) = noun, sth
        L, B, Q = lg_n.shape[0], lg_n.shape[1], lg_n.shape[2]
        dev = lg_n.device
        if m_n.counts != m_s.counts:
            raise ValueError("softkd needs the same number of targets on the noun and the pronoun side of every pair")

        def binarise(lg):
            p = lg.float().softmax(-1)
            return torch.cat([p[..., :-1].sum(-1, keepdim=True), p[..., -1:]], dim=-1)                  # [L,B,Q,2]

        def unmatched_first(m):
            """[L,B,Q] query order with the unmatched queries first (ascending), the matched ones after."""
            free = torch.ones(L, B, Q, dtype=torch.int8, device=dev)
            if m.src.shape[1]:
                b_of = _pair_image_index(tuple(m.counts), dev)                                         # [Mtot]
                free.view(L, B * Q).scatter_(1, b_of[None, :] * Q + m.src, 0)        # (scatter: advanced-index assignment synchronises the host)
            return torch.sort(free, dim=-1, descending=True, stable=True).indices

        p_n, p_s = binarise(lg_n).detach(), binarise(lg_s)
        ord_n, ord_s = unmatched_first(m_n), unmatched_first(m_s)
        take = lambda t, order: torch.gather(t, 2, order[..., None].expand(-1, -1, -1, t.shape[-1]))
        fp_n, fp_s = take(p_n, ord_n), take(p_s, ord_s)                                                 # unmatched first
        with torch.no_grad():
            fb_n, fb_s = take(bx_n.float(), ord_n), take(bx_s.float(), ord_s)
            cost_class = (fp_n[:, :, None, :, :] * (fp_n.log()[:, :, None, :, :] - fp_s.log()[:, :, :, None, :])).sum(-1)   # [L,B,S,T]
            cost = (torch.cdist(fb_s, fb_n, p=1) + cost_class - _paired_giou_matrix(fb_s, fb_n)).contiguous()
        shapes = [(Q - c, Q - c) for _ in range(L) for c in m_n.counts]
        offsets = [(l * B + i) * Q * Q for l in range(L) for i in range(B)]
        rows, cols, out_off, pairs, status = lsap_blocks(cost, shapes, offsets, Q)
        # unmatched pairs: student row `rows`, teacher row `cols` of problem (l, i)
        pl, pb = _problem_index(L, tuple(pairs), dev)
        kl_fp = _kl_rows(fp_n[pl, pb, cols], fp_s[pl, pb, rows])                                        # [sum pairs]
        # matched pairs: both sides indexed by target -> position match_off[i] + tgt inside the image's run
        def by_target(p, m):
            out = torch.zeros(L, max(m.src.shape[1], 1), 2, device=dev, dtype=p.dtype)
            if m.src.shape[1]:
                b_of = _pair_image_index(tuple(m.counts), dev)
                base = _pair_image_offset(tuple(m.counts), dev)
                lidx = torch.arange(L, device=dev)[:, None].expand_as(m.src)
                out = out.index_put((lidx, base[None, :] + m.tgt), p[lidx, b_of[None, :].expand_as(m.src), m.src])
            return out
        kl_tp = _kl_rows(by_target(p_n, m_n), by_target(p_s, m_s))                                      # [L, Mtot]
        # per image: batchmean over its (matched + assigned unmatched) rows; then mean over images
        n_rows = [c + (Q - c) for c in m_n.counts]
        counts = tuple(m_n.counts)
        w_tp = _cached(("w_tp", counts, Q, B, str(dev)),
                       lambda: torch.tensor([1.0 / (n_rows[i] * B) for i, c in enumerate(counts) for _ in range(c)] or [0.0], device=dev))
        w_fp = _cached(("w_fp", counts, L, Q, B, str(dev)),
                       lambda: torch.tensor([1.0 / (n_rows[i] * B) for _ in range(L) for i, c in enumerate(counts) for _ in range(Q - c)] or [0.0], device=dev))
        per_layer = (kl_tp * w_tp[None, :kl_tp.shape[1]]).sum(1) if m_n.src.shape[1] else torch.zeros(L, device=dev)
        per_layer = per_layer + (kl_fp * w_fp[:kl_fp.shape[0]]).view(L, -1).sum(1)
        check_lsap_status(status, defer=True)      # an invalid cost block raises at the next call (no host sync inside the step)
        return per_layer


def _kl_rows(teacher, student):
    """sum_c t * (log t - log s) per row, with F.kl_div's 0 * log 0 = 0 convention."""
    return (torch.xlogy(teacher, teacher) - teacher * student.log()).sum(-1)


def _paired_giou_matrix(a, b):
    """generalized_box_iou(box_cxcywh_to_xyxy(a), box_cxcywh_to_xyxy(b)) for batched cxcywh boxes [..., S, 4] x [..., T, 4] -> [..., S, T]
    (util/box_ops.py:11-61, same operation order)."""
    def xyxy(x):
        cx, cy, w, h = x.unbind(-1)
        return torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], -1)
    a, b = xyxy(a), xyxy(b)
    area_a = (a[..., 2] - a[..., 0]) * (a[..., 3] - a[..., 1])
    area_b = (b[..., 2] - b[..., 0]) * (b[..., 3] - b[..., 1])
    lt = torch.max(a[..., :, None, :2], b[..., None, :, :2])
    rb = torch.min(a[..., :, None, 2:], b[..., None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    union = area_a[..., :, None] + area_b[..., None, :] - inter
    iou = inter / union
    lt = torch.min(a[..., :, None, :2], b[..., None, :, :2])
    rb = torch.max(a[..., :, None, 2:], b[..., None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    area = wh[..., 0] * wh[..., 1]
    return iou - (area - union) / area


_INDEX_CACHE = {}


def _cached(key, make):
    ent = _INDEX_CACHE.get(key)
    if ent is None:
        if len(_INDEX_CACHE) > 128:
            _INDEX_CACHE.clear()
        ent = _INDEX_CACHE[key] = make()
    return ent


def _pair_image_index(counts, dev):
    """[Mtot] image index of every matched pair."""
    return _cached(("img", counts, str(dev)), lambda: torch.tensor([i for i, c in enumerate(counts) for _ in range(c)], dtype=torch.int64, device=dev))


def _pair_image_offset(counts, dev):
    """[Mtot] first position of the pair's image inside the [Mtot] run."""
    offs = [sum(counts[:i]) for i in range(len(counts))]
    return _cached(("off", counts, str(dev)), lambda: torch.tensor([offs[i] for i, c in enumerate(counts) for _ in range(c)], dtype=torch.int64, device=dev))


def _problem_index(L, pairs, dev):
    """(layer, image) of every LSAP output pair; problems are ordered layer-major."""
    B = len(pairs) // L
    def make():
        pl = [l for l in range(L) for i in range(B) for _ in range(pairs[l * B + i])]
        pb = [i for l in range(L) for i in range(B) for _ in range(pairs[l * B + i])]
        return torch.tensor(pl, dtype=torch.int64, device=dev), torch.tensor(pb, dtype=torch.int64, device=dev)
    return _cached(("prob", L, pairs, str(dev)), make)


class LossDict(dict):
    """What the criterion returns: the reference's dict of scalar losses, plus the stacked tensors those scalars are views of
    (`groups`: [(tensor, {key: flat index})]).  Summing the dict key by key, as engine.py:77 does, costs two tiny kernels
    per key forward and three per key backward (select_backward = fill + copy + accumulate): ~1.4 ms per step for the
    30 detection keys.  weighted_total() forms the same sum from the stacked tensors with two kernels each way."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.groups = []
        self.join = None        # a side stream some of the values were produced on: the consumer's stream must wait for it (SetCriterion._forward_pair_static)

    def merge(self, other, prefix=""):
        self.update({prefix + k_: v for k_, v in other.items()})
        for stacked, index in getattr(other, "groups", ()):
            self.groups.append((stacked, {prefix + k_: i for k_, i in index.items()}))
        return self


_WEIGHT_VECTORS = {}


def weighted_total(loss_dict, weight_dict):
    """sum(loss_dict[k] * weight_dict[k] for k in loss_dict if k in weight_dict) -- engine.py:77 / :227."""
    total, covered = None, set()
    for stacked, index in getattr(loss_dict, "groups", ()):
        flat = stacked.reshape(-1)
        key = (str(flat.device), flat.numel(), tuple(sorted((i, float(weight_dict[k_])) for k_, i in index.items() if k_ in weight_dict)))
        covered.update(index)
        if not key[2]:           # none of the group's keys is weighted: no term (and no backward pass through the group's graph for a sum of zeros)
            continue
        w = _WEIGHT_VECTORS.get(key)
        if w is None:
            host = torch.zeros(flat.numel(), dtype=torch.float32)
            for i, v in key[2]:
                host[i] = v
            w = _WEIGHT_VECTORS[key] = host.to(flat.device)
        term = (flat.float() * w).sum()        # (torch.dot would put a rocBLAS launch into the replayed step)
        total = term if total is None else total + term
    for k_, v in loss_dict.items():
        if k_ in weight_dict and k_ not in covered:
            total = v * weight_dict[k_] if total is None else total + v * weight_dict[k_]
    return total


class _SetLossFn(torch.autograd.Function):
    """losses [L,4] = (loss_ce, loss_bbox, loss_giou, cardinality_error) per decoder layer."""

    @staticmethod
    def forward(ctx, logits, boxes, match, positive_map, num_boxes, eos_coef):
        L = logits.shape[0]
        losses = torch.zeros(L, 4, dtype=torch.float32, device=logits.device)
        k.criterion_fwd(logits, boxes, match.tgt_boxes, positive_map, match.tgt_off_dev, match.match_off_dev, match.src, match.tgt,
                        num_boxes, eos_coef, losses, match.status if match.tgt_boxes is not None else None)
        ctx.save_for_backward(logits, boxes, positive_map, num_boxes)
        ctx.match, ctx.eos = match, eos_coef
        return losses

    @staticmethod
    def backward(ctx, g):
        logits, boxes, positive_map, num_boxes = ctx.saved_tensors
        m = ctx.match
        dlogits, dboxes = torch.empty_like(logits), torch.empty_like(boxes)
        k.criterion_bwd(logits, boxes, m.tgt_boxes, positive_map, m.tgt_off_dev, m.match_off_dev, m.src, m.tgt, num_boxes, ctx.eos,
                        g.float().contiguous(), dlogits, dboxes)
        return dlogits, dboxes, None, None, None, None


class _ContrastiveFn(torch.autograd.Function):
    """losses [L] = loss_contrastive_align per visited decoder layer (csrc/contrastive.hip)."""

    @staticmethod
    def forward(ctx, pq, pt, match, tok_mask, num_boxes, temperature):
        L = pq.shape[0]
        losses = torch.zeros(L, dtype=torch.float32, device=pq.device)
        k.contrastive_fwd(pq, pt, tok_mask, match.tgt_off_dev, match.match_off_dev, match.src[-L:], match.tgt[-L:], num_boxes, temperature, losses)
        ctx.save_for_backward(pq, pt, tok_mask, num_boxes)
        ctx.match, ctx.temperature = match, temperature
        return losses

    @staticmethod
    def backward(ctx, g):
        pq, pt, tok_mask, num_boxes = ctx.saved_tensors
        m, L = ctx.match, pq.shape[0]
        dpq, dpt = torch.empty_like(pq), torch.zeros_like(pt)
        k.contrastive_bwd(pq, pt, tok_mask, m.tgt_off_dev, m.match_off_dev, m.src[-L:], m.tgt[-L:], num_boxes, ctx.temperature,
                          g.float().contiguous(), dpq, dpt)
        return dpq, dpt, None, None, None, None


class _L2NormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y = torch.empty_like(x)
        k.l2norm_fwd(x, y)
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        dx = torch.empty_like(x)
        k.l2norm_bwd(x, g.float().contiguous(), dx)
        return dx


def l2_normalize(x):
    """F.normalize(x, p=2, dim=-1) on fp32 device rows (mdetr.py:429-433), forward and backward on csrc/contrastive.hip."""
    return _L2NormFn.apply(x.float().contiguous())


def _paired_giou(a, b):
    """GIoU of matching rows (the diagonal the reference takes at mdetr.py:820-822); cxcywh inputs."""
    ax0, ay0, ax1, ay1 = a[..., 0] - 0.5 * a[..., 2], a[..., 1] - 0.5 * a[..., 3], a[..., 0] + 0.5 * a[..., 2], a[..., 1] + 0.5 * a[..., 3]
    bx0, by0, bx1, by1 = b[..., 0] - 0.5 * b[..., 2], b[..., 1] - 0.5 * b[..., 3], b[..., 0] + 0.5 * b[..., 2], b[..., 1] + 0.5 * b[..., 3]
    area_a, area_b = (ax1 - ax0) * (ay1 - ay0), (bx1 - bx0) * (by1 - by0)
    iw = (torch.min(ax1, bx1) - torch.max(ax0, bx0)).clamp(min=0)
    ih = (torch.min(ay1, by1) - torch.max(ay0, by0)).clamp(min=0)
    inter = iw * ih
    union = area_a + area_b - inter
    ew = (torch.max(ax1, bx1) - torch.min(ax0, bx0)).clamp(min=0)
    eh = (torch.max(ay1, by1) - torch.min(ay0, by0)).clamp(min=0)
    hull = ew * eh
    return inter / union - (hull - union) / hull


def build(args):
    """Reference build(), mdetr.py:1039-1141: (model, criterion, cluster_criterion, weight_dict)."""
    num_classes = 255
    device = torch.device(args.device)
    assert not args.masks or args.mask_model != "none"
    backbone = build_backbone(args)
    transformer = build_transformer(args)
    model = MDETR(backbone, transformer, num_classes=num_classes, num_queries=args.num_queries, aux_loss=args.aux_loss,
                  contrastive_hdim=args.contrastive_loss_hdim, contrastive_align_loss=args.contrastive_align_loss,
                  cluster_num=getattr(args, "cluster_num", 16), args=args)
    if args.mask_model != "none":
        from .segmentation import DETRsegm
        model = DETRsegm(model, mask_head=args.mask_model, freeze_detr=(args.frozen_weights is not None))
    matcher = build_matcher(args)
    weight_dict = {"loss_ce": args.ce_loss_coef, "loss_bbox": args.bbox_loss_coef}
    if args.contrastive_align_loss:
        weight_dict["loss_contrastive_align"] = args.contrastive_align_loss_coef
    nsthl2, softkd = getattr(args, "nsthl2_loss", False), getattr(args, "softkd_loss", False)
    cluster, distillation = getattr(args, "cluster", False), getattr(args, "distillation", False)
    if nsthl2:
        weight_dict["loss_nsthl2"] = args.nsthl2_coef
    if softkd:
        weight_dict["loss_softkd"] = args.softkd_coef
    if cluster and distillation:
        weight_dict["loss_cluster_choice"] = args.cluster_choice_loss
        weight_dict["loss_cluster_feature"] = args.cluster_feature_loss
    weight_dict["loss_giou"] = args.giou_loss_coef
    if args.masks:
        weight_dict["loss_mask"] = args.mask_loss_coef
        weight_dict["loss_dice"] = args.dice_loss_coef
    if distillation:   # per-model losses get the noun_ / sth_ prefixes, the cross losses keep their names (mdetr.py:1083-1097)
        cross = ("loss_nsthl2", "loss_softkd", "loss_cluster_choice", "loss_cluster_feature")
        prefixed = {}
        for kk, v in weight_dict.items():
            if kk in cross:
                prefixed[kk] = v
            else:
                prefixed["noun_" + kk] = v
                prefixed["sth_" + kk] = v
        weight_dict = prefixed
    if args.aux_loss:
        aux = {}
        for i in range(args.dec_layers - 1):
            aux.update({kk + f"_{i}": v for kk, v in weight_dict.items()})
        weight_dict.update(aux)
    losses = ["labels", "boxes", "cardinality"]
    if args.masks:
        losses += ["masks"]
    if args.contrastive_align_loss:
        losses += ["contrastive_align"]
    if nsthl2:
        losses += ["nsthl2"]
    if softkd:
        losses += ["softkd"]
    criterion = SetCriterion(args, num_classes, matcher=matcher, eos_coef=args.eos_coef, losses=losses,
                             temperature=args.temperature_NCE, contrastive_hdim=args.contrastive_loss_hdim)
    criterion.to(device)
    cluster_criterion = None
    if cluster:
        cluster_criterion = ClusterCriterion(feature_dim=args.hidden_dim, memory_size=args.cluster_memory_size, cluster_num=args.cluster_num,
                                             task_count=14, args=args)
    return model, criterion, cluster_criterion, weight_dict
