"""Synthetic-input harness shared by bench.py, __graft_entry__.smoke() and the tests: default args
(the flag defaults of /root/reference/main.py:32-274) and the synthetic batch of SURVEY.md 8(d)."""
from types import SimpleNamespace

import torch

from .transformer import TokenizedText


def default_args(**over):
    a = dict(
        device="cuda", masks=False, mask_model="none", frozen_weights=None, num_queries=100, aux_loss=True,
        contrastive_loss_hdim=64, contrastive_align_loss=False, contrastive_loss=False, cluster_num=3, dec_layers=6, enc_layers=6,
        eos_coef=0.1, temperature_NCE=0.07, hidden_dim=256, nheads=8, dim_feedforward=2048, dropout=0.1, pre_norm=False,
        pass_pos_and_query=True, text_encoder_type="roberta-base", freeze_text_encoder=False, without_pretrain=True,
        ce_loss_coef=1.0, bbox_loss_coef=5.0, giou_loss_coef=2.0, mask_loss_coef=1.0, dice_loss_coef=1.0,
        contrastive_align_loss_coef=1.0, set_loss="hungarian", set_cost_class=1.0, set_cost_bbox=5.0, set_cost_giou=2.0,
        lr_backbone=1e-5, backbone="resnet101", dilation=False, position_embedding="sine", nsthl2_loss=False, softkd_loss=False,
        cluster=False, distillation=False, lr=1e-4, text_encoder_lr=5e-5, weight_decay=1e-4, clip_max_norm=0.1,
        nsthl2_coef=1e4, softkd_coef=1.0, cluster_choice_loss=0.0, cluster_feature_loss=1e4, cluster_memory_size=1024, fifo_memory=False,
        train_batch_size=4,
    )
    a.update(over)
    return SimpleNamespace(**a)


def finite_or_exit(loss_value, loss_dict=None, criterion=None):
    """The NaN / inf guard of the reference loop (engine.py:82-85, 212-215): print the losses and sys.exit(1).  When the criterion is
    given, an invalid matcher cost block is reported as the ValueError SciPy raises in the reference (matcher.py:85)."""
    import math
    import sys
    v = float(loss_value)
    from . import kernels
    kernels.xdec_check()        # (the float() above synchronised) a group of an XCD-resident launch that was not co-resident gave up waiting: results invalid
    if math.isfinite(v):
        return v
    if criterion is not None:
        criterion.check_status()
    print("Loss is {}, stopping training".format(v))
    if loss_dict is not None:
        print({k: float(x) for k, x in loss_dict.items()})
    sys.exit(1)


def synthetic_batch(batch, height=640, width=640, tokens=16, seed=1000, device="cpu", max_targets=10, with_masks=False):
    """Images ~ N(0,1) (already normalised), all-False padding mask; captions = <s> + ids + </s>;
    T_i ~ U{0..max_targets} boxes with cx,cy~U(.2,.8), w,h~U(.05,.4); positive_map rows = 1/(tokens-2)
    on the caption tokens."""
    g = torch.Generator().manual_seed(seed)
    images = torch.randn(batch, 3, height, width, generator=g)
    mask = torch.zeros(batch, height, width, dtype=torch.bool)
    ids = torch.randint(3, 50265, (batch, tokens), generator=g)
    ids[:, 0], ids[:, -1] = 0, 2
    tokenized = TokenizedText({"input_ids": ids, "attention_mask": torch.ones(batch, tokens, dtype=torch.int64)})
    targets, rows = [], []
    for i in range(batch):
        t = int(torch.randint(0, max_targets + 1, (1,), generator=g))
        c = torch.rand(t, 2, generator=g) * 0.6 + 0.2
        s = torch.rand(t, 2, generator=g) * 0.35 + 0.05
        boxes = torch.cat([c, s], -1)
        pm = torch.zeros(t, 256)
        pm[:, 1:tokens - 1] = 1.0 / (tokens - 2)
        tgt = {"boxes": boxes, "labels": torch.ones(t, dtype=torch.int64), "positive_map": pm,
               "token_spans": [[(1, tokens - 2)] for _ in range(t)]}
        if with_masks:
            m = torch.zeros(t, height, width, dtype=torch.bool)
            for j in range(t):
                x0, y0 = int((boxes[j, 0] - boxes[j, 2] / 2) * width), int((boxes[j, 1] - boxes[j, 3] / 2) * height)
                x1, y1 = int((boxes[j, 0] + boxes[j, 2] / 2) * width), int((boxes[j, 1] + boxes[j, 3] / 2) * height)
                m[j, max(y0, 0):max(y1, 1), max(x0, 0):max(x1, 1)] = True
            tgt["masks"] = m
        targets.append(tgt)
        rows.append(pm)
    positive_map = torch.cat(rows) if rows else torch.zeros(0, 256)

    def to(x):
        return x.to(device) if torch.is_tensor(x) else x

    from .misc import NestedTensor
    samples = NestedTensor(images.to(device), mask.to(device))
    targets = [{k: (to(v) if k != "token_spans" else v) for k, v in t.items()} for t in targets]
    return samples, tokenized.to(device), targets, positive_map.to(device)


# ---- distillation (BASELINE config 5): (noun, pronoun) pairs -------------------------------------------------------
class SyntheticCaptions(TokenizedText):
    """Pre-tokenised synthetic captions that also answer char_to_token the way a HF BatchEncoding does, for the
    span lookups of the distillation losses (no tokenizer files exist offline): 4 characters per token, token 0 = <s>."""

    def char_to_token(self, batch_or_char, char=None):
        c = batch_or_char if char is None else char
        t = c // 4 + 1
        return t if 0 <= c and t < self["input_ids"].shape[1] - 1 else None


def synthetic_distill_batch(batch, height=640, width=640, tokens=16, seed=1000, device="cpu", max_targets=10):
    """What collate_fn (util/misc.py:40-91) delivers for `batch` (noun, pronoun) pairs: the two sides share image and
    boxes; the noun caption names the object (characters 8..15 -> tokens 3..4), the pronoun caption says 'something'
    at the same place.  Returns dict(samples, targets, positive_map, captions, tokenized) with two-element lists."""
    samples, tok, targets, pmap = synthetic_batch(batch, height, width, tokens=tokens, seed=seed, device=device, max_targets=max_targets)
    out = {"samples": [samples, samples], "positive_map": [pmap, pmap], "example_rel": list(range(batch)), "targets": [], "captions": [], "tokenized": []}
    for side, word in enumerate(("scissors", "something")):
        caption = ("use the " + word + " to cut the paper up")[:4 * (tokens - 2)]
        out["captions"].append([caption] * batch)
        tg = []
        for i, t in enumerate(targets):
            t = dict(t)
            t["noun_tokens_positive"] = [[(8, 8 + len(word))] for _ in range(len(t["boxes"]))]
            t["dataset_name"] = f"task_{1 + (i + side * 0) % 14}_train.json"
            tg.append(t)
        out["targets"].append(tg)
        out["tokenized"].append(SyntheticCaptions(dict(tok)))
    return out


def distillation_losses(model, model_noun, criterion, cluster_criterion, batch):
    """Forward half of engine.py:152-190 (train_one_epoch_distillation) up to the loss dict: teacher and student encode, memory-bank update and prototype
    substitution, both decodes, the paired criterion (+ the cluster losses)."""
    s_noun, s_sth = batch["samples"]
    t_noun, t_sth = batch["targets"]
    c_noun, c_sth = batch["captions"]
    k_noun, k_sth = batch["tokenized"]
    mc_noun = model_noun(s_noun, k_noun, encode_and_save=True)
    if cluster_criterion is not None:
        mc_noun = cluster_criterion.update_memory(mc_noun, t_noun, c_noun)
    out_noun = model_noun(s_noun, k_noun, encode_and_save=False, memory_cache=mc_noun)
    mc_sth = model(s_sth, k_sth, encode_and_save=True)
    loss_cluster = {}
    if cluster_criterion is not None:
        mc_sth, loss_cluster = cluster_criterion(mc_sth, t_sth, c_sth)
    out_sth = model(s_sth, k_sth, encode_and_save=False, memory_cache=mc_sth)
    losses = criterion([mc_noun, mc_sth], [out_noun, out_sth], [t_noun, t_sth], batch["positive_map"], batch.get("example_rel"))
    losses.update(loss_cluster)
    return losses


def distillation_step(model, model_noun, criterion, cluster_criterion, weight_dict, batch):
    """distillation_losses + the weighted total (engine.py:227).  Returns (total loss, loss dict)."""
    losses = distillation_losses(model, model_noun, criterion, cluster_criterion, batch)
    from .mdetr import weighted_total
    join = getattr(losses, "join", None)
    if join is not None:                       # (only inside a capture: part of the dict was produced on a side stream)
        torch.cuda.current_stream().wait_stream(join)
    total = weighted_total(losses, weight_dict)
    return total, losses


# ---- evaluation (engine.py:253-342) ---------------------------------------------------------------------------------
def synthetic_ground_truth(batches):
    """COCO-format ground truth of synthetic batches (lists of target dicts carrying "image_id", "orig_size", "boxes" in
    normalised cxcywh and optionally "masks" at the original size): what COCO(annFile) holds for the evaluator."""
    images, anns = [], []
    for targets in batches:
        for t in targets:
            h, w = (int(v) for v in t["orig_size"].tolist())
            img = int(t["image_id"])
            images.append({"id": img, "height": h, "width": w})
            boxes = t["boxes"].detach().float().cpu()
            for j in range(boxes.shape[0]):
                cx, cy, bw, bh = (float(v) for v in boxes[j])
                ann = {"id": len(anns) + 1, "image_id": img, "category_id": 1, "iscrowd": 0,
                       "bbox": [(cx - bw / 2) * w, (cy - bh / 2) * h, bw * w, bh * h], "area": bw * w * bh * h}
                if "masks" in t:
                    m = t["masks"][j].cpu().numpy()
                    ann["segmentation"], ann["area"] = m, float(m.sum())
                anns.append(ann)
    return {"images": images, "annotations": anns, "categories": [{"id": 1, "name": "preferred"}]}


@torch.no_grad()
def evaluate(model, criterion, cluster_criterion, postprocessors, weight_dict, batches, evaluator_list, device, args):
    """The reference's evaluation loop: per batch encode -> (prototype choice) -> decode -> losses for logging -> PostProcess
    (-> PostProcessSegm) -> evaluator.update; then gather across ranks, accumulate, summarize.  `batches` yields dicts with
    "samples", "tokenized" (or captions), "targets", "positive_map"; returns {"loss": ..., "coco_eval_bbox": [12 numbers],
    "coco_eval_masks": [...]}."""
    from . import dist as tdist
    from .mdetr import weighted_total
    model.eval()
    if criterion is not None:
        criterion.eval()
    if cluster_criterion is not None:
        cluster_criterion.eval()
    sums, n = {}, 0
    for batch in batches:
        samples, targets = batch["samples"], batch["targets"]
        text = batch["tokenized"] if "tokenized" in batch else [t["caption"] for t in targets]
        memory_cache = model(samples, text, encode_and_save=True)
        if getattr(args, "cluster", False):
            memory_cache = cluster_criterion.infer_choice(memory_cache, [t["dataset_name"] for t in targets], [t["caption"] for t in targets])
        outputs = model(samples, text, encode_and_save=False, memory_cache=memory_cache)
        if criterion is not None:
            loss_dict = tdist.reduce_dict(criterion(memory_cache, outputs, targets, batch.get("positive_map"), batch.get("example_rel")))
            for name, v in loss_dict.items():
                sums[name] = sums.get(name, 0.0) + float(v)
            sums["loss"] = sums.get("loss", 0.0) + float(weighted_total(loss_dict, weight_dict))
        n += 1
        orig = torch.stack([t["orig_size"] for t in targets], dim=0)
        results = postprocessors["bbox"](outputs, orig)
        if "segm" in postprocessors:
            results = postprocessors["segm"](results, outputs, orig, torch.stack([t["size"] for t in targets], dim=0))
        res = {int(t["image_id"]): r for t, r in zip(targets, results)}
        for evaluator in evaluator_list:
            evaluator.update(res)
        from . import kernels
        kernels.xdec_check()      # (the results above reached the host: no extra synchronisation) an XCD-resident launch whose groups were not co-resident
    stats = {name: v / max(n, 1) for name, v in sums.items()}
    for evaluator in evaluator_list:
        evaluator.synchronize_between_processes()
        evaluator.accumulate()
        evaluator.summarize(verbose=tdist.is_main_process() and getattr(args, "verbose_eval", False))
        if "bbox" in evaluator.coco_eval:
            stats["coco_eval_bbox"] = evaluator.coco_eval["bbox"].stats.tolist()
        if "segm" in evaluator.coco_eval:
            stats["coco_eval_masks"] = evaluator.coco_eval["segm"].stats.tolist()
    return stats


# ---- the captured training step as a library feature -------------------------------------------------------------------------
class CapturedTrainStep:
    """The reference's training step (engine.py:54-101: encode -> decode -> SetCriterion -> weighted sum -> backward -> clip + AdamW + EMA)
    replayed from hipGraphs, one per input-shape BUCKET, with real variable-size batches.

    The reference resizes images to 480..800 x <= 1333 (datasets/tdod.py:305-319) and pads a batch to its largest image
    (util/misc.py:185-209); captions are padded to the longest of the batch.  A captured graph has static shapes, so the IMAGES are padded a
    little further -- height / width up to multiples of `pad_hw` -- with the padding masked out exactly as the reference masks its own
    padding (NestedTensor.mask): extra masked pixels change no result.  Caption length is NOT rounded up by default (`pad_tokens=1`):
    padded token positions are masked in every attention, but `loss_contrastive_align` takes its log-sum-exp over EVERY token column
    without an attention mask (/root/reference/models/mdetr.py:646-663), so extra pad tokens would change that loss and its gradients.
    `pad_tokens > 1` is an explicit trade: fewer graphs for captions of mixed lengths, and the contrastive term then equals the
    reference's on a batch whose longest caption has the padded length (the detection losses are unaffected).
    One graph is captured per bucket (Hp, Wp, Lp) on first use and kept in an LRU of `max_graphs`;
    targets travel through matcher.StaticTargets (fixed-address device image, any number of targets per image up to
    `max_targets_per_image`), dropout masks change per replay through the device seed word, learning rates are re-read from the
    optimizer's device table (`optimizer.sync_hyperparams()` after changing them).

    step(samples, tokenized, targets, positive_map) -> total loss (device scalar, valid until the next step of the same bucket).
    The FIRST step of a new bucket runs eagerly (it is a real training step) and the graph is captured right after it, without
    executing anything; later steps of that bucket are one host-to-device copy of the inputs + one graph launch.
    Single process per GPU; with torch.distributed active the step falls back to the eager path (collectives are not captured).
    Forward passes made between steps (validation) must run under torch.no_grad(): a grad-enabled forward that never sees backward()
    leaves AccumulateGrad nodes bound to the stream it ran on, and torch's autograd engine would pull that stream into the next capture."""

    def __init__(self, model, criterion, optimizer, weight_dict, *, batch, max_targets_per_image=16, pad_hw=64, pad_tokens=1, max_graphs=4,
                 contrastive=None, device=None):
        from collections import OrderedDict
        from . import kernels
        self.model, self.criterion, self.optimizer, self.weight_dict = model, criterion, optimizer, weight_dict
        self.batch, self.max_t = int(batch), int(max_targets_per_image)
        self.pad_hw, self.pad_tokens, self.max_graphs = int(pad_hw), int(pad_tokens), int(max_graphs)
        self.device = torch.device(device) if device is not None else next(model.parameters()).device
        det = getattr(model, "detr", model)
        self.num_queries = det.query_embed.weight.shape[0]
        self.contrastive = bool(getattr(det, "contrastive_align_loss", False)) if contrastive is None else bool(contrastive)
        # configs[2]: the ground-truth masks travel in the bucket's StaticTargets, zero-padded to the bucket's (Hp, Wp).  The reference pads them to the
        # batch's largest image (util/misc.py:185-209) and resizes the [ceil(H/4), ceil(W/4)] predictions to that size (mdetr.py:843):
        # StaticTargets.valid_hw carries the batch's own sizes to the mask-loss kernels, which map that corner of the prediction onto that corner of
        # the targets and normalise by H * W -- the same mask losses in any bucket (round 6).
        self.masks = "masks" in getattr(criterion, "losses", ())
        self._buckets = OrderedDict()          # (Hp, Wp, Lp) -> dict(graph, images, mask, ids, att, targets, loss)
        self._side = torch.cuda.Stream(device=self.device)
        if kernels.SEED_DEV is None:
            kernels.SEED_DEV = torch.zeros(1, dtype=torch.int64, device=self.device)
        from . import engine
        engine.REUSE_GRAD_BUFFERS = True       # the loop owns the gradients: one flat buffer per program, shared by every bucket's graph and the eager steps
        self.captures = 0
        self.replays = 0
        self._xdec_off = kernels.XDEC_FAILED

    # -- helpers ------------------------------------------------------------------------------------------------------------------
    def bucket_of(self, samples, tokenized):
        H, W = samples.tensors.shape[-2:]
        L = tokenized["input_ids"].shape[1]
        up = lambda v, m: (int(v) + m - 1) // m * m
        return up(H, self.pad_hw), up(W, self.pad_hw), up(L, self.pad_tokens)

    def _static_inputs(self, key):
        from .matcher import StaticTargets
        from .misc import NestedTensor
        from .transformer import TokenizedText
        Hp, Wp, Lp = key
        dev = self.device
        pad_id = getattr(getattr(self.model, "detr", self.model).transformer.text_encoder.config, "pad_token_id", 1)
        ent = {"samples": NestedTensor(torch.zeros(self.batch, 3, Hp, Wp, device=dev), torch.ones(self.batch, Hp, Wp, dtype=torch.bool, device=dev)),
               "tok": TokenizedText({"input_ids": torch.full((self.batch, Lp), pad_id, dtype=torch.int64, device=dev),
                                     "attention_mask": torch.zeros(self.batch, Lp, dtype=torch.int64, device=dev)}),
               "targets": StaticTargets(self.batch, self.max_t, self.num_queries, 256, dev, mask_hw=(Hp, Wp) if self.masks else None),
               "graph": None, "loss": None, "pad_id": pad_id}
        return ent

    def _fill(self, ent, samples, tokenized, targets, positive_map, packed):
        img, msk = ent["samples"].tensors, ent["samples"].mask
        B, _, H, W = samples.tensors.shape
        if B != self.batch:
            raise ValueError(f"CapturedTrainStep was built for batches of {self.batch} images (got {B})")
        if (H, W) != tuple(img.shape[-2:]):
            img.zero_()
            msk.fill_(True)
        img[:, :, :H, :W].copy_(samples.tensors, non_blocking=True)
        msk[:, :H, :W].copy_(samples.mask, non_blocking=True)
        ids, att = ent["tok"]["input_ids"], ent["tok"]["attention_mask"]
        L = tokenized["input_ids"].shape[1]
        if L != ids.shape[1]:
            ids.fill_(ent["pad_id"])
            att.zero_()
        ids[:, :L].copy_(tokenized["input_ids"], non_blocking=True)
        att[:, :L].copy_(tokenized["attention_mask"], non_blocking=True)
        st = ent["targets"]
        if packed is None:
            host_t = [{k_: (v.cpu() if torch.is_tensor(v) else v) for k_, v in t.items()} for t in targets]
            masks = self.criterion.token_masks_host(host_t, tokenized) if self.contrastive else None
            st.load(host_t, positive_map.cpu() if torch.is_tensor(positive_map) else positive_map, masks)
        else:
            st.load_packed(packed)

    def _fwd_bwd_opt(self, ent):
        from . import kernels
        from .mdetr import weighted_total
        kernels.SEED_DEV.add_(1000003)
        mc = self.model(ent["samples"], ent["tok"], encode_and_save=True)
        out = self.model(ent["samples"], ent["tok"], encode_and_save=False, memory_cache=mc)
        losses = self.criterion(mc, out, ent["targets"], None, None)
        total = weighted_total(losses, self.weight_dict)
        total.backward()
        self.optimizer.step()
        return total

    # -- the step -------------------------------------------------------------------------------------------------------------------
    def step(self, samples, tokenized, targets=None, positive_map=None, packed=None):
        from . import engine
        distributed = torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1
        key = self.bucket_of(samples, tokenized)
        ent = self._buckets.get(key)
        fresh = ent is None
        if fresh:
            ent = self._static_inputs(key)
            self._buckets[key] = ent
            while len(self._buckets) > self.max_graphs:
                self._buckets.popitem(last=False)          # least recently used bucket: its graph and activation pool are released
        from . import kernels as _k
        if _k.XDEC_FAILED and not self._xdec_off:
            # an XCD-resident decoder launch reported an expired spin (kernels.xdec_check, reached through harness.finite_or_exit on the NaN loss of
            # that step): the captured graphs contain those launches -- drop them; the next step of every bucket runs eagerly on the per-op launches
            # and is captured again
            self._xdec_off = True
            for e in self._buckets.values():
                e["graph"] = e["loss"] = None
            ent["graph"] = None
        self._buckets.move_to_end(key)
        self._fill(ent, samples, tokenized, targets, positive_map, packed)
        if ent["graph"] is not None:
            ent["graph"].replay()
            self.replays += 1
            if hasattr(self.optimizer, "note_replayed_step"):
                self.optimizer.note_replayed_step()      # copies of the weights that the captured tail does not rewrite are stale now
            return ent["loss"]
        # first batch of this bucket: a real, eager training step, launched on the object's own stream -- the stream the graph is captured
        # on right afterwards, so that every per-stream cache of the launchers (split-K scratch, reduction arena) exists before the
        # capture begins instead of being allocated inside it ...
        side = self._side
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self.optimizer.zero_grad(set_to_none=True)
            total = self._fwd_bwd_opt(ent)
            if not distributed:
                # ... then the capture of the same call sequence on the static inputs (nothing executes during capture)
                self.optimizer.zero_grad(set_to_none=True)
                from . import kernels
                graph = torch.cuda.CUDAGraph()
                with kernels.tables_beside_graph():     # pointer tables of grouped launches: filled once, not on every replay
                    with torch.cuda.graph(graph, stream=side):
                        ent["loss"] = self._fwd_bwd_opt(ent)
                ent["graph"] = graph
                self.captures += 1
        torch.cuda.current_stream().wait_stream(side)
        return total

    __call__ = step


class CapturedDistillStep:
    """The reference's distillation step (engine.py:152-204: teacher encode -> memory-bank update + prototypes -> teacher decode -> student encode ->
    prototype of 'something' -> student decode -> paired SetCriterion with softkd / nsthl2 + cluster losses -> backward -> clip + AdamW + EMA of both
    models) replayed from ONE hipGraph for ANY batch of a fixed shape (round 6).

    What used to tie a captured graph to its batch -- per-image target counts (LSAP problem sizes (Q - c)^2, pair tables), the grouping of the images by
    task (k-means groups, one memory-bank LSAP per task), the token tables of the captions (noun spans, the word 'something') -- now lives in
    fixed-address device images that `step()` refills before every replay: matcher.StaticTargets (one per side) and distill.DistillTables (one per
    side, attached as StaticTargets.distill); the criterion and the cluster criterion dispatch on them (SetCriterion._forward_pair_static,
    ClusterCriterion.update_memory_static / forward_static).

    step(batch) with batch = what util/misc.collate_fn delivers for (noun, pronoun) pairs (harness.synthetic_distill_batch): dict(samples=[..] * 2,
    tokenized=[..] * 2 (BatchEncodings with char_to_token), targets=[..] * 2, captions=[..] * 2, positive_map=[..] * 2) -> total loss (device scalar).
    The first step runs eagerly (a real training step) and is captured right after; later steps = the H2D copies of the inputs + one graph launch.
    Requirements (checked): one process, every memory bank full, nearest-replacement bank updates, both sides with the same number of targets per
    image (the reference's pairs share their boxes), images / captions of the constructor's shape."""

    def __init__(self, model, model_noun, criterion, cluster_criterion, optimizers, weight_dict, *, batch, image_hw, tokens, max_targets_per_image=16, device=None,
                 stream=None):
        """stream: the HIP stream the steps run (and the graph is captured) on; pass the stream earlier eager steps of the same models ran on, if any -- the programs'
        reused gradient buffers and the launchers' per-stream scratch must not change streams between eager steps and the capture."""
        from . import engine, kernels
        from .distill import DistillTables
        from .matcher import StaticTargets
        from .misc import NestedTensor
        from .transformer import TokenizedText
        self.model, self.model_noun, self.criterion, self.cluster_criterion = model, model_noun, criterion, cluster_criterion
        self.optimizers, self.weight_dict = list(optimizers), weight_dict
        self.B, self.hw, self.L = int(batch), (int(image_hw[0]), int(image_hw[1])), int(tokens)
        dev = self.device = torch.device(device) if device is not None else next(model.parameters()).device
        det = getattr(model, "detr", model)
        Q = det.query_embed.weight.shape[0]
        self.contrastive = bool(getattr(det, "contrastive_align_loss", False))
        pad_id = getattr(det.transformer.text_encoder.config, "pad_token_id", 1)
        H, W = self.hw
        self.sides = []
        for pronoun in (False, True):
            st = StaticTargets(self.B, int(max_targets_per_image), Q, 256, dev)
            st.distill = DistillTables(self.B, self.L, dev, pronoun_side=pronoun)
            self.sides.append({"samples": NestedTensor(torch.zeros(self.B, 3, H, W, device=dev), torch.zeros(self.B, H, W, dtype=torch.bool, device=dev)),
                               "tok": TokenizedText({"input_ids": torch.full((self.B, self.L), pad_id, dtype=torch.int64, device=dev),
                                                     "attention_mask": torch.zeros(self.B, self.L, dtype=torch.int64, device=dev)}),
                               "targets": st})
        self._side = stream if stream is not None else torch.cuda.Stream(device=dev)
        if kernels.SEED_DEV is None:
            kernels.SEED_DEV = torch.zeros(1, dtype=torch.int64, device=dev)
        engine.REUSE_GRAD_BUFFERS = True
        self.graph, self.loss, self._xdec = None, None, False
        self.captures = self.replays = 0

    def pack(self, batch):
        """Host half of a step (no device work, no synchronisation when the batch's targets are host tensors: run it in the loader): the pinned images of both
        sides' StaticTargets and DistillTables, next to the batch's image / token tensors.  step(packed=...) consumes it."""
        if [len(t["boxes"]) for t in batch["targets"][0]] != [len(t["boxes"]) for t in batch["targets"][1]]:
            raise ValueError("softkd needs the same number of targets on the noun and the pronoun side of every pair")
        out = []
        for i, side in enumerate(self.sides):
            samples, tok, targets = batch["samples"][i], batch["tokenized"][i], batch["targets"][i]
            if tuple(samples.tensors.shape) != (self.B, 3, *self.hw) or tuple(tok["input_ids"].shape) != (self.B, self.L):
                raise ValueError(f"CapturedDistillStep was built for {self.B} x 3 x {self.hw[0]} x {self.hw[1]} images and {self.L}-token captions")
            host_t = [{k_: (v.cpu() if torch.is_tensor(v) else v) for k_, v in t.items()} for t in targets]
            pm = batch["positive_map"][i]
            masks = self.criterion.token_masks_host(host_t, tok) if self.contrastive else None
            out.append({"samples": samples, "tok": tok, "targets": side["targets"].pack(host_t, pm.cpu() if torch.is_tensor(pm) else pm, masks),
                        "tables": side["targets"].distill.pack(tok, host_t, batch["captions"][i])})
        return out

    def _fill(self, packed):
        for side, pk in zip(self.sides, packed):
            side["samples"].tensors.copy_(pk["samples"].tensors, non_blocking=True)
            side["samples"].mask.copy_(pk["samples"].mask, non_blocking=True)
            side["tok"]["input_ids"].copy_(pk["tok"]["input_ids"], non_blocking=True)
            side["tok"]["attention_mask"].copy_(pk["tok"]["attention_mask"], non_blocking=True)
            side["targets"].load_packed(pk["targets"])
            side["targets"].distill.load_packed(pk["tables"])

    def _fwd_bwd_opt(self):
        """One step.  The backward pass is issued in TWO calls -- the teacher's total (the noun_ keys) first, then everything else (sth_, softkd, nsthl2, cluster:
        the student's) -- which changes no gradient (the two models' graphs are disjoint: the teacher enters the cross losses detached, mdetr.py:520-599, 668-781)
        but lets a captured step run the teacher's backward beside the softkd assignment problems, which occupy 24 CUs for ~3.6 ms on a side stream
        (SetCriterion._forward_pair_static)."""
        from . import kernels
        from .mdetr import weighted_total
        kernels.SEED_DEV.add_(1000003)
        noun, sth = self.sides
        static = {"samples": [noun["samples"], sth["samples"]], "targets": [noun["targets"], sth["targets"]], "captions": [None, None],
                  "tokenized": [noun["tok"], sth["tok"]], "positive_map": [None, None]}
        self.criterion.defer_pair_join = True          # the softkd block may go to a side stream: this method joins it (below) before the student's total
        try:
            losses = distillation_losses(self.model, self.model_noun, self.criterion, self.cluster_criterion, static)
        finally:
            self.criterion.defer_pair_join = False
        w_noun = {k_: v for k_, v in self.weight_dict.items() if k_.startswith("noun_")}
        w_rest = {k_: v for k_, v in self.weight_dict.items() if not k_.startswith("noun_")}
        total_noun = weighted_total(losses, w_noun)
        total_noun.backward()
        join = getattr(losses, "join", None)
        if join is not None:
            torch.cuda.current_stream().wait_stream(join)
        total_rest = weighted_total(losses, w_rest)
        total_rest.backward()
        for o in self.optimizers:
            o.step()
        return (total_noun + total_rest).detach()

    def step(self, batch=None, packed=None):
        from . import kernels
        batch = packed if packed is not None else self.pack(batch)
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            raise RuntimeError("CapturedDistillStep is single-process (the memory-bank queue all_gathers rows across ranks: collectives are not captured)")
        if kernels.XDEC_FAILED and self.graph is not None and self._xdec:
            self.graph = self.loss = None                  # the captured graph holds XCD-resident launches that have been turned off: capture again
        side = self._side
        if self.graph is not None:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self._fill(batch)
                self.graph.replay()
            torch.cuda.current_stream().wait_stream(side)
            self.replays += 1
            for o in self.optimizers:
                if hasattr(o, "note_replayed_step"):
                    o.note_replayed_step()
            return self.loss
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self._fill(batch)
            for o in self.optimizers:
                o.zero_grad(set_to_none=True)
            total = self._fwd_bwd_opt()
            for o in self.optimizers:
                o.zero_grad(set_to_none=True)
            graph = torch.cuda.CUDAGraph()
            with kernels.tables_beside_graph():
                with torch.cuda.graph(graph, stream=side):
                    self.loss = self._fwd_bwd_opt()
            self.graph, self._xdec = graph, not kernels.XDEC_FAILED
            self.captures += 1
        torch.cuda.current_stream().wait_stream(side)
        return total

    __call__ = step
