"""Golden vectors for the segmentation branch (config 3) from the REAL reference modules
(models/segmentation.py: MHAttentionMap, MaskHeadSmallConv, dice_loss, sigmoid_focal_loss and
SetCriterion.loss_masks).  Run in the build container: PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_segm.py"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import formula  # noqa: E402
import make_golden as mg  # noqa: E402  (creates the import stubs, puts /root/reference on sys.path)

from models.matcher import HungarianMatcher  # noqa: E402
from models.mdetr import SetCriterion  # noqa: E402
from models.segmentation import MaskHeadSmallConv, MHAttentionMap  # noqa: E402


def main():
    B, Q, d, H, h, w = 2, 5, 256, 8, 3, 4
    att = MHAttentionMap(d, d, H, dropout=0)
    head = MaskHeadSmallConv(d + H, [1024, 512, 256], d)
    att.eval(); head.eval()
    sd = {"bbox_attention." + k: v for k, v in att.state_dict().items()}
    sd.update({"mask_head." + k: v for k, v in head.state_dict().items()})
    filled = formula.fill_state_dict(sd)
    att.load_state_dict({k[len("bbox_attention."):]: v for k, v in filled.items() if k.startswith("bbox_attention.")})
    head.load_state_dict({k[len("mask_head."):]: v for k, v in filled.items() if k.startswith("mask_head.")})
    import json
    with open(os.path.join(HERE, "reference_segm_state_dict_shapes.json"), "w") as f:
        json.dump({k: list(v.shape) for k, v in sd.items()}, f, indent=0, sort_keys=True)
    hs = formula.tensor("sg.hs", (B, Q, d), 2.0)
    memory = formula.tensor("sg.mem", (B, d, h, w), 2.0)
    src_proj = formula.tensor("sg.src", (B, d, h, w), 2.0)
    fmask = torch.zeros(B, h, w, dtype=torch.bool)
    fmask[1, :, 3:] = True
    fpns = [formula.tensor("sg.c4", (B, 1024, 2 * h, 2 * w), 2.0).clamp(min=0), formula.tensor("sg.c3", (B, 512, 4 * h, 4 * w), 2.0).clamp(min=0),
            formula.tensor("sg.c2", (B, 256, 8 * h, 8 * w), 2.0).clamp(min=0)]
    with torch.no_grad():
        bm = att(hs, memory, mask=fmask)
        seg = head(src_proj, bm, fpns)
    pred = seg.view(B, Q, 8 * h, 8 * w)
    # mask losses through the reference criterion (matcher on synthetic logits / boxes)
    logits = formula.tensor("sg.logits", (B, Q, 256), 6.0)
    boxes = torch.cat([formula.tensor("sg.bc", (B, Q, 2), 0.6, 0.5), formula.tensor("sg.bs", (B, Q, 2), 0.3, 0.2)], -1)
    sizes = [2, 3]
    targets = []
    for i, t in enumerate(sizes):
        bx = torch.cat([formula.tensor(f"sg.tc{i}", (t, 2), 0.6, 0.5), formula.tensor(f"sg.ts{i}", (t, 2), 0.3, 0.2)], -1)
        m = formula.tensor(f"sg.m{i}", (t, 32 * h - 8 * i, 32 * w), 1.0) > 0.1   # ragged heights: exercises NestedTensor padding
        targets.append({"boxes": bx, "labels": torch.ones(t, dtype=torch.int64), "masks": m})
    pm = torch.zeros(sum(sizes), 256)
    pm[:, 1:6] = 0.2
    args = types.SimpleNamespace(num_queries=Q, nsthl2_loss=False, softkd_loss=False)
    crit = SetCriterion(args, 255, matcher=HungarianMatcher(1.0, 5.0, 2.0), eos_coef=0.1, losses=["labels", "boxes", "cardinality", "masks"],
                        temperature=0.07, contrastive_hdim=64)
    with torch.no_grad():
        losses = crit(None, {"pred_logits": logits, "pred_boxes": boxes, "pred_masks": pred}, targets, pm, None)
    rec = {"bbox_mask": bm, "pred_masks": pred, "fmask": fmask, "logits": logits, "boxes": boxes, "pm": pm, "sizes": np.array(sizes),
           "names": np.array(sorted(losses)), "values": np.array([float(losses[k]) for k in sorted(losses)], dtype=np.float64)}
    for i, t in enumerate(targets):
        rec[f"tboxes{i}"], rec[f"tmasks{i}"] = t["boxes"], t["masks"]
    mg.save("segm.npz", **rec)


if __name__ == "__main__":
    main()
