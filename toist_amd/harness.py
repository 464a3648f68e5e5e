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


def distillation_step(model, model_noun, criterion, cluster_criterion, weight_dict, batch):
    """Forward half of engine.py:152-190 (train_one_epoch_distillation): teacher and student encode, memory-bank update
    and prototype substitution, both decodes, the paired criterion.  Returns (total loss, loss dict)."""
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
    from .mdetr import weighted_total
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
