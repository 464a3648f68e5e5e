"""GPU parity of the evaluation path against oracle/coco_ref.py through the C ABI:
bit planes <-> dense masks, run lengths and their text form (bit-exact), areas and IoU (bit-exact doubles), the fused
PostProcessSegm kernel (fp32 bilinear arithmetic: <= 1e-5 of the pixels may fall on the other side of the threshold), the
batched evaluateImg matching and the 12 COCO summary numbers (exact)."""
import numpy as np
import pytest
import torch

from oracle import coco_ref as R

pytestmark = pytest.mark.gpu

SHAPES = [(1, 1), (5, 7), (64, 64), (63, 65), (130, 67), (480, 640)]


def random_masks(rng, n, h, w):
    m = rng.random((n, h, w)) < rng.random((n, 1, 1))
    if h * w >= 64:                                             # blobs: realistic run structure
        for i in range(n // 2):
            y0, x0 = rng.integers(0, h), rng.integers(0, w)
            m[i] = False
            m[i, y0:y0 + rng.integers(1, h + 1), x0:x0 + rng.integers(1, w + 1)] = True
    m[0] = False
    if n > 1:
        m[1] = True
    return m


@pytest.mark.parametrize("h,w", SHAPES)
def test_planes_areas_and_run_lengths_are_bit_exact(dev, h, w):
    from toist_amd import coco_eval as C, kernels as k
    rng = np.random.default_rng(h * 1000 + w)
    masks = random_masks(rng, 7, h, w)
    bits = k.mask_pack(torch.from_numpy(masks).to(dev))
    assert bits.shape == (7, w, (h + 63) // 64)
    assert np.array_equal(k.mask_unpack(bits, h, w).cpu().numpy(), masks)
    assert k.mask_area(bits, h, w).cpu().tolist() == [int(m.sum()) for m in masks]
    counts, first = k.mask_rle(bits, h, w)
    counts, first = counts.cpu().numpy(), first.cpu().numpy()
    for i, m in enumerate(masks):
        want = R.rle_encode(m)
        got = counts[first[i]:first[i + 1]].tolist()
        assert got == want
        assert C.counts_to_string(got) == R.rle_to_string(want)
        assert np.array_equal(C.counts_to_mask(got, h, w), m)                  # encode -> decode round trip


@pytest.mark.parametrize("h,w", [(5, 7), (63, 65), (200, 333)])
def test_mask_iou_is_bit_exact(dev, h, w):
    from toist_amd import kernels as k
    rng = np.random.default_rng(h + w)
    dt, gt = random_masks(rng, 9, h, w), random_masks(rng, 5, h, w)
    crowd = [0, 1, 0, 1, 0]
    db, gb = k.mask_pack(torch.from_numpy(dt).to(dev)), k.mask_pack(torch.from_numpy(gt).to(dev))
    iou = k.mask_iou(db, gb, torch.tensor(crowd, dtype=torch.uint8, device=dev), k.mask_area(db, h, w), k.mask_area(gb, h, w), h, w)
    want = R.rle_iou([R.rle_encode(m) for m in dt], [R.rle_encode(m) for m in gt], crowd)
    assert np.array_equal(iou.cpu().numpy(), want)


@pytest.mark.parametrize("ragged", [False, True])
def test_postprocess_segm_matches_reference_arithmetic(dev, ragged):
    from toist_amd.postprocessors import PostProcess, PostProcessSegm
    torch.manual_seed(3)
    B, Q = 3, 11
    pred = torch.randn(B, Q, 1, 40, 52) * 3
    sizes = torch.tensor([[160, 208], [150, 180], [131, 208]] if ragged else [[160, 208]] * 3)
    origs = torch.tensor([[333, 500], [480, 410], [97, 640]] if ragged else [[427, 640]] * 3)
    want = R.postprocess_segm(pred, origs, sizes)
    outputs = {"pred_logits": torch.randn(B, Q, 256).to(dev), "pred_boxes": torch.rand(B, Q, 4).to(dev), "pred_masks": pred.to(dev)}
    results = PostProcess()(outputs, origs.to(dev))
    dense = PostProcessSegm()(results, outputs, origs.to(dev), sizes.to(dev))
    packed = PostProcessSegm(packed=True)([{} for _ in range(B)], outputs, origs.to(dev), sizes.to(dev))
    from toist_amd import kernels as k
    for i in range(B):
        got = dense[i]["masks"]
        assert got.dtype == torch.bool and not got.is_cuda and tuple(got.shape) == (Q, 1, int(origs[i, 0]), int(origs[i, 1]))
        flipped = int((got != want[i]).sum())
        assert flipped <= 1e-5 * got.numel() + 1, (i, flipped, got.numel())
        assert 0.2 < float(got.float().mean()) < 0.8
        h, w = packed[i]["mask_size"]
        assert torch.equal(k.mask_unpack(packed[i]["mask_bits"], h, w).cpu(), got[:, 0])


def synthetic_eval_set(seed, n_img=6, with_masks=True):
    """COCO-format ground truth + predictions in the reference's result format: jittered copies of the objects, duplicates,
    random false positives, tied scores, a crowd region, an image without ground truth and one without detections."""
    rng = np.random.default_rng(seed)
    images, anns, res = [], [], {}
    for i in range(n_img):
        img = 10 + 3 * i
        h, w = int(rng.integers(90, 200)), int(rng.integers(90, 260))
        images.append({"id": img, "height": h, "width": w})
        n_gt = 0 if i == 1 else int(rng.integers(1, 6))
        boxes, masks, scores = [], [], []
        for j in range(n_gt):
            bw, bh = int(rng.integers(4, w // 2)), int(rng.integers(4, h // 2))
            x, y = int(rng.integers(0, w - bw)), int(rng.integers(0, h - bh))
            m = np.zeros((h, w), dtype=bool)
            m[y:y + bh, x:x + bw] = rng.random((bh, bw)) < 0.9
            seg = {"size": [h, w], "counts": R.rle_encode(m)} if j % 2 else {"size": [h, w], "counts": R.rle_to_string(R.rle_encode(m))}
            anns.append({"id": len(anns) + 1, "image_id": img, "category_id": 1 if rng.random() < 0.9 else 2, "iscrowd": int(j == 2),
                         "bbox": [float(x), float(y), float(bw), float(bh)], "area": float(m.sum()), "segmentation": seg})
            for _ in range(2):                                   # two jittered detections per object
                dx, dy = rng.integers(-6, 7, 2)
                mm = np.roll(np.roll(m, dy, 0), dx, 1)
                boxes.append([x + dx + rng.normal(), y + dy + rng.normal(), x + dx + bw + rng.normal(), y + dy + bh + rng.normal()])
                masks.append(mm)
                scores.append(np.round(rng.random(), 1))         # coarse scores: ties exercise the stable sorts
        for _ in range(int(rng.integers(0, 5))):                 # false positives
            bw, bh = int(rng.integers(3, w // 3)), int(rng.integers(3, h // 3))
            x, y = int(rng.integers(0, w - bw)), int(rng.integers(0, h - bh))
            m = np.zeros((h, w), dtype=bool)
            m[y:y + bh, x:x + bw] = True
            boxes.append([x, y, x + bw, y + bh])
            masks.append(m)
            scores.append(np.round(rng.random(), 1))
        if i == 2:
            boxes, masks, scores = [], [], []
        n = len(boxes)
        res[img] = {"scores": torch.tensor(scores, dtype=torch.float32).view(n), "labels": torch.ones(n, dtype=torch.int64),
                    "boxes": torch.tensor(boxes, dtype=torch.float32).view(n, 4)}
        if with_masks:
            res[img]["masks"] = torch.from_numpy(np.stack(masks)[:, None] if n else np.zeros((0, 1, h, w), dtype=bool))
    return {"images": images, "annotations": anns}, res


def oracle_stats(dataset, res, iou_type):
    from toist_amd import coco_eval as C
    gts = []
    for a in dataset["annotations"]:
        g = dict(a)
        c = a["segmentation"]["counts"]
        g["counts"] = R.rle_from_string(c) if isinstance(c, str) else c
        gts.append(g)
    ev = R.CocoEvalRef(gts, R.detections_from_results({i: {k_: v.numpy() for k_, v in p.items()} for i, p in res.items()}, iou_type),
                       list(res.keys()), iou_type)
    ev.evaluate()
    ev.accumulate()
    return ev, ev.summarize()


@pytest.mark.parametrize("iou_type", ["bbox", "segm"])
def test_evaluator_matches_oracle_exactly(dev, iou_type):
    from toist_amd.coco_eval import TDODCocoEvaluator
    dataset, res = synthetic_eval_set(5)
    ref, want = oracle_stats(dataset, res, iou_type)
    ev = TDODCocoEvaluator(dataset, [iou_type], device=dev)
    ids = list(res.keys())
    ev.update({i: res[i] for i in ids[:4]})                      # two batches, like the evaluation loop
    ev.update({i: res[i] for i in ids[4:]})
    ev.synchronize_between_processes()
    ev.accumulate()
    ev.summarize(verbose=False)
    got = ev.coco_eval[iou_type]
    # per-image tables: same detections matched, same ones ignored, for every area range and threshold
    imgs, I = sorted(ids), len(ids)
    for i, img in enumerate(imgs):
        rec = got.records[img]
        for a in range(4):
            e = ref.eval_imgs[a * I + i]
            if e is None:
                assert rec["scores"].size == 0 and rec["gt_ignore"].shape[1] == 0
                continue
            assert np.array_equal(rec["dt_match"][a] >= 0, e["dtMatches"] > 0), (img, a)
            assert np.array_equal(rec["dt_ignore"][a], e["dtIgnore"].astype(bool)), (img, a)
            assert int((~rec["gt_ignore"][a]).sum()) == int((e["gtIgnore"] == 0).sum())
    assert np.array_equal(got.eval["precision"][:, :, 0], ref.precision[:, :, 0])
    assert np.array_equal(got.stats, want), (got.stats, want)
    assert 0.05 < want[0] < 0.95 and want[1] > want[0]


def test_segmentation_results_carry_pycocotools_text_rle(dev):
    from toist_amd.coco_eval import TDODCocoEvaluator
    dataset, res = synthetic_eval_set(6, n_img=3)
    ev = TDODCocoEvaluator(dataset, ["segm"], device=dev)
    out = ev.segmentation_results(res)
    it = iter(out)
    for img, pred in res.items():
        for q in range(pred["masks"].shape[0]):
            r = next(it)
            m = pred["masks"][q, 0].numpy()
            assert r["image_id"] == img and r["segmentation"]["size"] == list(m.shape)
            assert r["segmentation"]["counts"] == R.rle_to_string(R.rle_encode(m))
    assert next(it, None) is None


def test_full_size_round_trip_and_packed_evaluation(dev):
    """BASELINE-size property checks: 100 queries at 480 x 640, PostProcessSegm(packed) -> evaluator without a dense copy;
    detections that ARE the ground truth score AP = AR = 1, and decode(encode(mask)) is the identity."""
    from toist_amd import coco_eval as C, kernels as k
    from toist_amd.postprocessors import PostProcessSegm
    torch.manual_seed(0)
    B, Q, H, W = 2, 100, 480, 640
    pred = torch.randn(B, Q, 1, 160, 160, device=dev) * 2 - 1
    sizes, origs = torch.tensor([[640, 640]] * B, device=dev), torch.tensor([[H, W]] * B, device=dev)
    outputs = {"pred_masks": pred}
    results = PostProcessSegm(packed=True)([{} for _ in range(B)], outputs, origs, sizes)
    images, anns = [], []
    for i in range(B):
        bits = results[i]["mask_bits"]
        dense = k.mask_unpack(bits, H, W)
        counts, first = k.mask_rle(bits, H, W)
        counts, first = counts.cpu().numpy(), first.cpu().numpy()
        assert int(first[-1]) == counts.size and np.array_equal(np.add.reduceat(counts.astype(np.int64), first[:-1]), np.full(Q, H * W))
        for q in (0, 17, 99):
            assert np.array_equal(C.counts_to_mask(counts[first[q]:first[q + 1]], H, W), dense[q].cpu().numpy())
        assert torch.equal(k.mask_pack(dense), bits)
        images.append({"id": i + 1, "height": H, "width": W})
        area = k.mask_area(bits, H, W).cpu().tolist()
        for q in range(5):                                        # the first five predictions are declared ground truth
            anns.append({"id": len(anns) + 1, "image_id": i + 1, "category_id": 1, "iscrowd": 0, "area": float(area[q]),
                         "bbox": [0.0, 0.0, float(W), float(H)], "segmentation": dense[q].cpu().numpy()})
        results[i]["scores"] = torch.linspace(1, 0, Q, device=dev)
        results[i]["labels"] = torch.ones(Q, dtype=torch.int64, device=dev)
    ev = C.TDODCocoEvaluator({"images": images, "annotations": anns}, ["segm"], device=dev)
    ev.update({i + 1: results[i] for i in range(B)})
    ev.accumulate()
    ev.summarize(verbose=False)
    stats = ev.coco_eval["segm"].stats
    assert stats[0] == pytest.approx(1.0, abs=1e-12) and stats[8] == pytest.approx(1.0, abs=1e-12)


def test_evaluation_loop_end_to_end(dev):
    """engine.evaluate's shape on a random-init segmentation model: two batches through encode / decode / losses / PostProcess /
    PostProcessSegm(packed) / TDODCocoEvaluator; the packed route and the reference-format route (dense host masks) agree."""
    import toist_amd
    from toist_amd import harness
    from toist_amd.postprocessors import PostProcess, PostProcessSegm
    torch.manual_seed(0)
    args = harness.default_args(device="cuda", masks=True, mask_model="smallconv", enc_layers=1, dec_layers=1, num_queries=20)
    model, criterion, _, weight_dict = toist_amd.build_model(args)
    model.to(dev)
    batches = []
    for b in range(2):
        samples, tok, targets, pmap = harness.synthetic_batch(2, 128, 160, tokens=12, seed=20 + b, device=dev, max_targets=4, with_masks=True)
        for i, t in enumerate(targets):
            t["image_id"] = torch.tensor([100 + 2 * b + i], device=dev)
            t["orig_size"], t["size"] = torch.tensor([128, 160], device=dev), torch.tensor([128, 160], device=dev)
        batches.append({"samples": samples, "tokenized": tok, "targets": targets, "positive_map": pmap})
    gt = harness.synthetic_ground_truth([b["targets"] for b in batches])
    stats = {}
    for packed in (True, False):
        post = {"bbox": PostProcess(), "segm": PostProcessSegm(packed=packed)}
        ev = toist_amd.TDODCocoEvaluator(gt, ["bbox", "segm"], device=dev)
        stats[packed] = harness.evaluate(model, criterion, None, post, weight_dict, batches, [ev], dev, args)
        assert sorted(ev.img_ids) == [100, 101, 102, 103]
    for name in ("coco_eval_bbox", "coco_eval_masks"):
        assert len(stats[True][name]) == 12 and stats[True][name] == stats[False][name]
        assert all(-1.0 <= v <= 1.0 for v in stats[True][name])
    assert stats[True]["loss"] > 0 and "loss_dice" in stats[True]


@pytest.mark.parametrize("case", ["equal", "ragged"])
def test_postprocess_segm_vs_reference_fixture(dev, case):
    """The fused PostProcessSegm kernel against masks the REAL reference produced (tests/golden/postprocess_segm.npz, both of its
    branches).  The kernel composes the two bilinear resizes in one pass, so a pixel whose resized logit is within rounding of 0 may
    land on the other side of the threshold: at most 1e-4 of the pixels (and never a different shape / dtype)."""
    import os
    from toist_amd.postprocessors import PostProcessSegm
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "postprocess_segm.npz"))
    pred = torch.from_numpy(d["pred_masks"]).to(dev)
    orig, mx = torch.from_numpy(d[case + "_orig"]).to(dev), torch.from_numpy(d[case + "_max"]).to(dev)
    res = PostProcessSegm()([{} for _ in range(pred.shape[0])], {"pred_masks": pred}, orig, mx)
    for i, r in enumerate(res):
        m = r["masks"]
        want = np.unpackbits(d[f"{case}_bits{i}"])[:m.numel()].reshape(tuple(m.shape)).astype(bool)
        assert m.dtype == torch.bool and tuple(m.shape) == want.shape
        flipped = int((m.numpy() != want).sum())
        assert flipped <= 1e-4 * want.size + 1, (case, i, flipped, want.size)


def test_postprocess_values_vs_reference_fixture(dev):
    """PostProcess (scores = 1 - p(no object), labels = 1, boxes scaled to the image) against the REAL reference's outputs
    (tests/golden/postprocess.npz), computed on the device."""
    import os
    from toist_amd.postprocessors import PostProcess
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "postprocess.npz"))
    out = {"pred_logits": torch.from_numpy(d["logits"]).to(dev), "pred_boxes": torch.from_numpy(d["boxes"]).to(dev)}
    res = PostProcess()(out, torch.from_numpy(d["sizes"]).to(dev))
    for i, r in enumerate(res):
        assert r["scores"].is_cuda and r["labels"].dtype == torch.int64
        assert np.allclose(r["scores"].cpu().numpy(), d["scores"][i], rtol=1e-5, atol=1e-6)
        assert np.array_equal(r["labels"].cpu().numpy(), d["labels"][i])
        assert np.allclose(r["boxes"].cpu().numpy(), d["out_boxes"][i], rtol=1e-5, atol=1e-3)
