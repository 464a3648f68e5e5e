"""Post-processing (reference: /root/reference/models/postprocessors.py:15-117).  Inference-side,
tiny tensors; kept as device torch ops ("keep in torch unless profiled hot")."""
from typing import Dict

import torch
import torch.nn.functional as F
from torch import nn

from . import box_ops


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
        if "pred_isfinal" in outputs:
            is_final = outputs["pred_isfinal"].sigmoid()
            refexp = scores * is_final.view_as(scores)
            for i in range(len(results)):
                results[i]["scores_refexp"] = refexp[i]
        return results


class PostProcessSegm(nn.Module):
    def __init__(self, threshold=0.5):
        super().__init__()
        self.threshold = threshold

    @torch.no_grad()
    def forward(self, results, outputs, orig_target_sizes, max_target_sizes):
        assert len(orig_target_sizes) == len(max_target_sizes)
        max_h, max_w = max_target_sizes.max(0)[0].tolist()
        masks = outputs["pred_masks"].squeeze(2).float()
        masks = F.interpolate(masks, size=(max_h, max_w), mode="bilinear", align_corners=False)
        min_h, min_w = max_target_sizes.min(0)[0].tolist()
        min_oh, min_ow = orig_target_sizes.min(0)[0].tolist()
        max_oh, max_ow = orig_target_sizes.max(0)[0].tolist()
        if min_h == max_h and min_w == max_w and min_oh == max_oh and min_ow == max_ow:
            masks = (F.interpolate(masks, size=(min_oh, min_ow), mode="bilinear").sigmoid() > self.threshold).cpu()
            for i, m in enumerate(masks):
                results[i]["masks"] = m.unsqueeze(1)
            return results
        for i, (m, t, tt) in enumerate(zip(masks, max_target_sizes, orig_target_sizes)):
            h, w = int(t[0]), int(t[1])
            crop = m[:, :h, :w].unsqueeze(1)
            results[i]["masks"] = (F.interpolate(crop.float(), size=tuple(tt.tolist()), mode="bilinear").sigmoid() > self.threshold).cpu()
        return results


def build_postprocessors(args, dataset_name=None) -> Dict[str, nn.Module]:
    post: Dict[str, nn.Module] = {"bbox": PostProcess()}
    if args.masks:
        post["segm"] = PostProcessSegm()
    return post
