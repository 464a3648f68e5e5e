"""Evaluation-path microbenchmark at BASELINE size (B=8 images, Q=100 queries, 160x160 mask logits, 640x640 padded batch,
480x640 originals): the fused PostProcessSegm kernel against the reference's arithmetic as device torch ops (two fp32
bilinear resizes + sigmoid + threshold + dense D2H), device RLE encode, popcount IoU, the batched matching kernel, and one
whole TDODCocoEvaluator.update.  Prints one JSON object; durations are HIP-event averages on the launch stream."""
import json
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from toist_amd import coco_eval as C, kernels as k          # noqa: E402
from toist_amd.postprocessors import PostProcessSegm        # noqa: E402


def timed(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    B, Q, h0, H, W, PAD = 8, 100, 160, 480, 640, 640
    pred = torch.randn(B, Q, 1, h0, h0, device=dev) * 2 - 1
    sizes, origs = torch.tensor([[PAD, PAD]] * B, device=dev), torch.tensor([[H, W]] * B, device=dev)
    out = {}

    def fused():
        return PostProcessSegm(packed=True)([{} for _ in range(B)], {"pred_masks": pred}, origs, sizes)

    def torch_path(to_host):
        m = F.interpolate(pred.squeeze(2), size=(PAD, PAD), mode="bilinear", align_corners=False)
        m = F.interpolate(m, size=(H, W), mode="bilinear").sigmoid() > 0.5
        return m.cpu() if to_host else m
    t_f = timed(fused)
    out["postprocess_fused_ms"] = t_f
    out["postprocess_torch_device_only_ms"] = timed(lambda: torch_path(False), iters=5)
    out["postprocess_torch_with_dense_d2h_ms"] = timed(lambda: torch_path(True), iters=3, warm=1)
    alg = B * Q * (h0 * h0 * 4 + W * ((H + 63) // 64) * 8)
    out["postprocess_fused_algorithmic_bytes"] = alg
    out["postprocess_fused_GBps"] = alg / t_f / 1e6
    out["postprocess_fused_Gpixel_per_s"] = B * Q * H * W / t_f / 1e6
    res = fused()
    bits = torch.cat([r["mask_bits"] for r in res])                      # [800, W, 8]
    plane_bytes = bits.numel() * 8
    t = timed(lambda: k.mask_area(bits, H, W))
    out["area_ms"], out["area_GBps"] = t, plane_bytes / t / 1e6
    t = timed(lambda: k.mask_rle(bits, H, W), iters=5)
    counts, first = k.mask_rle(bits, H, W)
    out["rle_encode_800_masks_ms"], out["rle_runs"] = t, int(counts.numel())
    gt = bits[:10]
    area = k.mask_area(bits, H, W)
    crowd = torch.zeros(10, dtype=torch.uint8, device=dev)
    t = timed(lambda: k.mask_iou(bits[:100], gt, crowd, area[:100], area[:10], H, W))
    out["iou_100x10_ms"], out["iou_100x10_GBps_L2"] = t, 100 * 10 * 2 * (plane_bytes / 800) / t / 1e6
    # whole evaluator step: 8 images x 100 packed detections, 5 ground-truth masks each
    images = [{"id": i + 1, "height": H, "width": W} for i in range(B)]
    anns = []
    for i in range(B):
        dense = k.mask_unpack(res[i]["mask_bits"][:5], H, W).cpu().numpy()
        for q in range(5):
            anns.append({"id": len(anns) + 1, "image_id": i + 1, "category_id": 1, "iscrowd": 0, "area": float(dense[q].sum()),
                         "bbox": [0.0, 0.0, float(W), float(H)], "segmentation": dense[q]})
        res[i]["scores"], res[i]["labels"] = torch.rand(Q, device=dev), torch.ones(Q, dtype=torch.int64, device=dev)
        res[i]["boxes"] = torch.rand(Q, 4, device=dev) * 100
    ev = C.TDODCocoEvaluator({"images": images, "annotations": anns}, ["bbox", "segm"], device=dev)
    ev.update({i + 1: res[i] for i in range(B)})                          # packs the ground truth once
    t0 = time.perf_counter()
    for _ in range(5):
        ev.update({i + 1: res[i] for i in range(B)})
    torch.cuda.synchronize()
    out["evaluator_update_bbox_and_segm_ms"] = (time.perf_counter() - t0) / 5 * 1e3
    # CPU baseline: the oracle's encode + run-merging IoU (pure Python, as a port) on a bounded sample
    from oracle import coco_ref as R
    dense = k.mask_unpack(bits[:4], H, W).cpu().numpy()
    t0 = time.perf_counter()
    rles = [R.rle_encode(m) for m in dense]
    R.rle_iou(rles, rles[:2], [0, 0])
    out["cpu_oracle_encode4_iou4x2_ms"] = (time.perf_counter() - t0) * 1e3
    print(json.dumps({k_: (round(v, 4) if isinstance(v, float) else v) for k_, v in out.items()}))


if __name__ == "__main__":
    main()
