"""Oracle restatement of the Hungarian matcher (test infrastructure; see oracle/__init__.py).

Follows /root/reference/models/matcher.py:39-87 and /root/reference/util/box_ops.py:11-61 step by
step in fp32 on the CPU, with the assignment solved by oracle/lsap.c instead of SciPy.
"""
import torch

from . import lsap


def cxcywh_to_xyxy(b):
    """util/box_ops.py:11-14"""
    cx, cy, w, h = b.unbind(-1)
    return torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=-1)


def pairwise_iou(a, b):
    """util/box_ops.py:24-37 -> (iou [N,M], union [N,M])"""
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    top_left = torch.max(a[:, None, :2], b[None, :, :2])
    bot_right = torch.min(a[:, None, 2:], b[None, :, 2:])
    wh = (bot_right - top_left).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    union = area_a[:, None] + area_b[None, :] - inter
    return inter / union, union


def pairwise_giou(a, b):
    """util/box_ops.py:40-61"""
    if not bool((a[:, 2:] >= a[:, :2]).all()) or not bool((b[:, 2:] >= b[:, :2]).all()):
        raise AssertionError("degenerate boxes")
    iou, union = pairwise_iou(a, b)
    top_left = torch.min(a[:, None, :2], b[None, :, :2])
    bot_right = torch.max(a[:, None, 2:], b[None, :, 2:])
    wh = (bot_right - top_left).clamp(min=0)
    hull = wh[..., 0] * wh[..., 1]
    return iou - (hull - union) / hull


def cost_matrix(pred_logits, pred_boxes, tgt_boxes, positive_map, w_class=1.0, w_bbox=5.0, w_giou=2.0):
    """matcher.py:60-82 -> C [B, Q, Ttot] fp32 on the CPU."""
    B, Q = pred_logits.shape[:2]
    prob = torch.softmax(pred_logits.flatten(0, 1).float(), dim=-1)
    boxes = pred_boxes.flatten(0, 1).float()
    tgt_boxes = tgt_boxes.float()
    c_class = -(prob[:, None, :] * positive_map.float()[None, :, :]).sum(-1)
    c_bbox = torch.cdist(boxes, tgt_boxes, p=1)
    c_giou = -pairwise_giou(cxcywh_to_xyxy(boxes), cxcywh_to_xyxy(tgt_boxes))
    C = w_bbox * c_bbox + w_class * c_class + w_giou * c_giou
    return C.view(B, Q, -1)


def hungarian_match(pred_logits, pred_boxes, tgt_boxes_list, positive_map, w_class=1.0, w_bbox=5.0, w_giou=2.0):
    """matcher.py:39-87: list over images of (query_idx int64 ascending, target_idx int64)."""
    sizes = [int(t.shape[0]) for t in tgt_boxes_list]
    tgt = torch.cat(list(tgt_boxes_list)) if len(tgt_boxes_list) else torch.zeros(0, 4)
    assert tgt.shape[0] == positive_map.shape[0]
    with torch.no_grad():  # matcher.py:38 (@torch.no_grad)
        C = cost_matrix(pred_logits.detach().cpu(), pred_boxes.detach().cpu(), tgt.detach().cpu(), positive_map.detach().cpu(),
                        w_class, w_bbox, w_giou)
    out = []
    for i, blk in enumerate(C.split(sizes, -1)):
        r, c = lsap.linear_sum_assignment(blk[i].numpy())
        out.append((torch.as_tensor(r, dtype=torch.int64), torch.as_tensor(c, dtype=torch.int64)))
    return out
