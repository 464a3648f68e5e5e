"""Box utilities with the reference's semantics (/root/reference/util/box_ops.py:11-61).

These operate on a handful of boxes at API edges (post-processing, user code); the matcher and the
set criterion compute the same quantities inside their HIP kernels."""
import torch


def box_cxcywh_to_xyxy(x):
    cx, cy, w, h = x.unbind(-1)
    return torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=-1)


def box_xyxy_to_cxcywh(x):
    x0, y0, x1, y1 = x.unbind(-1)
    return torch.stack([(x0 + x1) / 2, (y0 + y1) / 2, x1 - x0, y1 - y0], dim=-1)


def _pairwise(a, b):
    """For xyxy boxes a [N,4], b [M,4]: (intersection area, union area, area of the smallest enclosing box), each [N,M] -- the three
    quantities IoU and GIoU are made of, from one broadcast of the four corners (the matcher / criterion kernels form the same three per
    (query, target) pair in registers: csrc/matcher.hip pair_costs, csrc/criterion.hip)."""
    ax0, ay0, ax1, ay1 = (a[:, None, i] for i in range(4))
    bx0, by0, bx1, by1 = (b[None, :, i] for i in range(4))
    overlap = (torch.minimum(ax1, bx1) - torch.maximum(ax0, bx0)).clamp(min=0) * (torch.minimum(ay1, by1) - torch.maximum(ay0, by0)).clamp(min=0)
    total = (ax1 - ax0) * (ay1 - ay0) + (bx1 - bx0) * (by1 - by0) - overlap
    hull = (torch.maximum(ax1, bx1) - torch.minimum(ax0, bx0)).clamp(min=0) * (torch.maximum(ay1, by1) - torch.minimum(ay0, by0)).clamp(min=0)
    return overlap, total, hull


def box_area(b):
    """area of xyxy boxes [N,4] (torchvision.ops.boxes.box_area, which the reference imports)"""
    return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])


def box_iou(boxes1, boxes2):
    """-> (iou [N,M], union [N,M]), the reference's return pair (util/box_ops.py:24-36)"""
    overlap, total, _ = _pairwise(boxes1, boxes2)
    return overlap / total, total


def generalized_box_iou(boxes1, boxes2):
    """GIoU [N,M] of xyxy boxes (util/box_ops.py:39-61; degenerate boxes are refused like there)"""
    if not (bool((boxes1[:, 2:] >= boxes1[:, :2]).all()) and bool((boxes2[:, 2:] >= boxes2[:, :2]).all())):
        raise AssertionError("generalized_box_iou: boxes must be xyxy with x1 >= x0 and y1 >= y0")
    overlap, total, hull = _pairwise(boxes1, boxes2)
    return overlap / total - (hull - total) / hull
