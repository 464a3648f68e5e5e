"""Post-processing (reference: /root/reference/models/postprocessors.py:15-117).  Inference-side,
PostProcess works on [B, Q] scalars and stays device torch ops; PostProcessSegm is HBM-bound mask work and
runs in csrc/evalmask.hip."""
from typing import Dict

import torch
import torch.nn.functional as F
from torch import nn

from . import box_ops
from . import kernels as k


class PostProcess(nn.Module):
    @torch.no_grad()
    def forward(self, outputs, target_sizes):
        out_logits, out_bbox = outputs["pred_logits"], outputs["pred_boxes"]
        assert len(out_logits) == len(target_sizes)
        assert target_sizes.shape[1] == 2
        prob = F.softmax(out_logits.float(), -1)
        scores = 1 - prob[:, :, -1]
        labels = torch.ones(prob.shape[:2], dtype=torch.int64, device=prob.device)
        boxes = box_ops.box_cxcywh_to_xyxy(out_bbox.float())
        img_h, img_w = target_sizes.unbind(1)
        boxes = boxes * torch.stack([img_w, img_h, img_w, img_h], dim=1)[:, None, :]
        results = [{"scores": s, "labels": l, "boxes": b} for s, l, b in zip(scores, labels, boxes)]
        final = outputs.get("pred_isfinal")                # MDETR's referring-expression head (postprocessors.py:49-54); no TOIST recipe emits it
        if final is not None:
            refexp = scores * final.float().sigmoid().view_as(scores)
            for r, s in zip(results, refexp):
                r["scores_refexp"] = s
        return results


class PostProcessSegm(nn.Module):
    """postprocessors.py:59-109.  The reference resizes [B, Q, H, W] in fp32 twice (to the padded batch size, then each image's
    un-padded corner to its original size), thresholds the sigmoid and copies Q dense masks per image to the host.  Here
    one kernel (csrc/evalmask.hip: mask_resize_pack) does both bilinear resizes, the sigmoid and the threshold straight from
    the [h0, w0] mask logits and writes column-major bit planes (1/32 of one fp32 mask).  packed=False (default) unpacks
    them into the reference's result format, results[i]["masks"] = bool [Q, 1, H_i, W_i] on the host; packed=True leaves
    results[i]["mask_bits"] (int64 [Q, W_i, ceil(H_i/64)]) and ["mask_size"] on the device for TDODCocoEvaluator."""

    def __init__(self, threshold=0.5, packed=False):
        super().__init__()
        self.threshold, self.packed = threshold, packed

    @torch.no_grad()
    def forward(self, results, outputs, orig_target_sizes, max_target_sizes):
        assert len(orig_target_sizes) == len(max_target_sizes)
        sizes, origs = max_target_sizes.tolist(), orig_target_sizes.tolist()
        max_h, max_w = max(s[0] for s in sizes), max(s[1] for s in sizes)
        logits = outputs["pred_masks"].squeeze(2).float()                     # [B, Q, h0, w0]
        for i, (size, orig) in enumerate(zip(sizes, origs)):
            h, w = int(orig[0]), int(orig[1])
            bits = k.mask_resize_pack(logits[i], (max_h, max_w), (int(size[0]), int(size[1])), (h, w), self.threshold)
            if self.packed:
                results[i]["mask_bits"], results[i]["mask_size"] = bits, (h, w)
            else:
                results[i]["masks"] = k.mask_unpack(bits, h, w).unsqueeze(1).cpu()
        return results


def build_postprocessors(args, dataset_name=None) -> Dict[str, nn.Module]:
    post: Dict[str, nn.Module] = {"bbox": PostProcess()}
    if args.masks:
        post["segm"] = PostProcessSegm()
    return post
