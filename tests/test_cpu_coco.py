"""Evaluation path on the CPU: the oracle restatement of pycocotools (oracle/coco_ref.py) against hand-derived known answers
and size-independent properties, and the host-side pieces of toist_amd/coco_eval.py (RLE text form, polygon rasteriser,
accumulate / summarize) against that oracle.  pycocotools itself is absent from the image: parity with it is UNPINNED."""
import numpy as np
import pytest

from oracle import coco_ref as R


def test_rle_known_answers():
    m = np.array([[0, 1], [1, 1]], dtype=np.uint8)                 # column-major pixel order: 0 1 | 1 1
    assert R.rle_encode(m) == [1, 3]
    assert R.rle_to_string([1, 3]) == "13"
    assert R.rle_encode(np.ones((2, 3))) == [0, 6]                    # a mask that starts with foreground: empty first run
    assert R.rle_encode(np.zeros((2, 3))) == [6]
    assert R.rle_area([1, 3]) == 3 and R.rle_area([6]) == 0
    # 5 data bits per character, 0x20 = continuation, +48: 32 -> (0 | 0x20) + 48 = 'P', then 1 + 48 = '1'
    assert R.rle_to_string([32]) == "P1"
    # 4th count onwards is stored as the difference to the count two back: 3 - 5 = -2 -> 0b11110 + 48 = 'N'
    assert R.rle_to_string([7, 5, 2, 3]) == "752N"
    assert R.rle_to_bbox([1, 3], 2, 2) == [0.0, 0.0, 2.0, 2.0]
    assert R.rle_to_bbox(R.rle_encode(np.pad(np.ones((2, 3)), ((1, 4), (2, 5)))), 7, 10) == [2.0, 1.0, 3.0, 2.0]


def test_rle_round_trips_and_iou_against_dense_counting():
    rng = np.random.default_rng(0)
    for h, w in [(1, 1), (5, 7), (64, 3), (65, 130), (37, 200)]:
        masks = rng.random((6, h, w)) < rng.random((6, 1, 1))
        masks[0], masks[1] = False, True
        rles = [R.rle_encode(m) for m in masks]
        for m, c in zip(masks, rles):
            assert sum(c) == h * w
            assert np.array_equal(R.rle_decode(c, h, w).astype(bool), m)
            assert R.rle_from_string(R.rle_to_string(c)) == c
            assert R.rle_area(c) == int(m.sum())
        crowd = [0, 1, 0, 1, 0, 0]
        iou = R.rle_iou(rles[:4], rles, crowd)
        for d in range(4):
            for g in range(6):
                i = int((masks[d] & masks[g]).sum())
                u = int(masks[d].sum()) if crowd[g] else int((masks[d] | masks[g]).sum())
                assert iou[d, g] == (i / u if i else 0.0)


def test_box_iou_known_answers():
    iou = R.bb_iou([[0, 0, 2, 2], [10, 10, 1, 1]], [[1, 1, 2, 2], [0, 0, 4, 4]], [0, 1])
    assert iou[0, 0] == 1 / 7 and iou[0, 1] == 1.0 and iou[1, 0] == 0.0 and iou[1, 1] == 0.0


def _boxes_case():
    # one image, two ground-truth boxes, three detections: hit (0.9), miss (0.8), hit (0.7)
    gts = [{"id": 1, "image_id": 7, "category_id": 1, "iscrowd": 0, "area": 2500.0, "bbox": [0, 0, 50, 50]},
           {"id": 2, "image_id": 7, "category_id": 1, "iscrowd": 0, "area": 2500.0, "bbox": [100, 100, 50, 50]}]
    dts = [{"id": 1, "image_id": 7, "category_id": 1, "score": 0.9, "area": 2500.0, "bbox": [0, 0, 50, 50]},
           {"id": 2, "image_id": 7, "category_id": 1, "score": 0.8, "area": 2500.0, "bbox": [300, 300, 50, 50]},
           {"id": 3, "image_id": 7, "category_id": 1, "score": 0.7, "area": 2500.0, "bbox": [100, 100, 50, 50]}]
    return gts, dts


def test_cocoeval_hand_computed_average_precision():
    gts, dts = _boxes_case()
    ev = R.CocoEvalRef(gts, dts, [7], "bbox")
    ev.evaluate()
    ev.accumulate()
    stats = ev.summarize()
    # recall after each detection: .5 .5 1; precision 1 .5 2/3 -> envelope 1 2/3 2/3; 51 recall thresholds <= .5, 50 above
    want = (51 * 1.0 + 50 * (2 / 3)) / 101
    assert abs(stats[0] - want) < 1e-9 and abs(stats[1] - want) < 1e-9 and abs(stats[2] - want) < 1e-9
    assert stats[3] == -1 and stats[5] == -1 and abs(stats[4] - want) < 1e-9          # both boxes are "medium" (32^2..96^2)
    assert stats[6] == 0.5 and stats[7] == 1.0 and stats[8] == 1.0
    perfect = R.CocoEvalRef(gts, [dts[0], dts[2]], [7], "bbox")
    perfect.evaluate()
    perfect.accumulate()
    assert perfect.summarize()[0] == pytest.approx(1.0, abs=1e-12)


def test_cocoeval_crowd_and_ignore_rules():
    # a crowd region may absorb several detections and none of them counts (neither TP nor FP)
    gts = [{"id": 1, "image_id": 1, "category_id": 1, "iscrowd": 1, "area": 10000.0, "bbox": [0, 0, 100, 100]},
           {"id": 2, "image_id": 1, "category_id": 1, "iscrowd": 0, "area": 400.0, "bbox": [200, 200, 20, 20]}]
    dts = [{"id": 1, "image_id": 1, "category_id": 1, "score": 0.9, "area": 100.0, "bbox": [10, 10, 10, 10]},
           {"id": 2, "image_id": 1, "category_id": 1, "score": 0.8, "area": 100.0, "bbox": [50, 50, 10, 10]},
           {"id": 3, "image_id": 1, "category_id": 1, "score": 0.7, "area": 400.0, "bbox": [200, 200, 20, 20]}]
    ev = R.CocoEvalRef(gts, dts, [1], "bbox")
    ev.evaluate()
    e = ev.eval_imgs[0]
    assert e["dtIgnore"][0].tolist() == [True, True, False] and e["dtMatches"][0].tolist() == [1, 1, 2]
    ev.accumulate()
    assert ev.summarize()[0] == pytest.approx(1.0, abs=1e-12)


# ---- host side of the product against the oracle ------------------------------------------------------------------
def test_rle_text_form_matches_oracle():
    from toist_amd import coco_eval as C
    rng = np.random.default_rng(1)
    for _ in range(50):
        counts = rng.integers(0, 5000, size=rng.integers(1, 40)).tolist()
        s = C.counts_to_string(counts)
        assert s == R.rle_to_string(counts)
        assert C.string_to_counts(s) == counts and C.string_to_counts(s.encode()) == counts
    m = rng.random((33, 70)) < 0.3
    assert np.array_equal(C.counts_to_mask(R.rle_encode(m), 33, 70), m)


def test_polygon_rasteriser_properties():
    from toist_amd import coco_eval as C
    # an axis-aligned polygon on integer corners covers exactly the pixels inside it
    m = C.polygon_to_mask([2, 1, 7, 1, 7, 5, 2, 5], 8, 10)
    want = np.zeros((8, 10), dtype=bool)
    want[1:5, 2:7] = True
    assert np.array_equal(m, want)
    # vertex order and starting vertex do not matter
    assert np.array_equal(C.polygon_to_mask([7, 5, 2, 5, 2, 1, 7, 1], 8, 10), want)
    assert np.array_equal(C.polygon_to_mask([2, 5, 7, 5, 7, 1, 2, 1], 8, 10), want)
    # a triangle: area within a boundary-pixel band of the exact area, clipped polygons stay inside the image
    tri = C.polygon_to_mask([0, 0, 40, 0, 0, 30], 32, 48)
    assert abs(int(tri.sum()) - 600) <= 40
    clipped = C.polygon_to_mask([-5, -5, 20, -5, 20, 50, -5, 50], 16, 12)
    assert clipped.shape == (16, 12) and clipped.all()
    # the dataset-side helper: one mask per object = union of its polygons; empty input keeps the reference's shape
    both = C.convert_coco_poly_to_mask([[[2, 1, 7, 1, 7, 5, 2, 5], [0, 6, 3, 6, 3, 8, 0, 8]], [[2, 1, 7, 1, 7, 5, 2, 5]]], 8, 10)
    want2 = want.copy()
    want2[6:8, 0:3] = True
    assert both.shape == (2, 8, 10) and np.array_equal(both[0].numpy(), want2) and np.array_equal(both[1].numpy(), want)
    assert tuple(C.convert_coco_poly_to_mask([], 8, 10).shape) == (0, 8, 10)


def test_accumulate_and_summarize_match_oracle():
    from toist_amd import coco_eval as C
    rng = np.random.default_rng(2)
    gts, dts, img_ids = [], [], list(range(1, 9))
    for img in img_ids:
        for _ in range(rng.integers(0, 5)):
            x, y, w, h = rng.uniform(0, 300), rng.uniform(0, 300), rng.uniform(5, 200), rng.uniform(5, 200)
            gts.append({"id": len(gts) + 1, "image_id": img, "category_id": 1, "iscrowd": int(rng.random() < 0.2), "area": w * h * rng.uniform(0.5, 1),
                        "bbox": [x, y, w, h]})
        for g in [g for g in gts if g["image_id"] == img] * 2:
            jit = rng.normal(0, 8, 4)
            b = [g["bbox"][0] + jit[0], g["bbox"][1] + jit[1], max(g["bbox"][2] + jit[2], 1), max(g["bbox"][3] + jit[3], 1)]
            dts.append({"id": len(dts) + 1, "image_id": img, "category_id": 1, "score": float(np.round(rng.random(), 2)), "area": b[2] * b[3], "bbox": b})
        for _ in range(rng.integers(0, 4)):
            b = [rng.uniform(0, 300), rng.uniform(0, 300), rng.uniform(5, 100), rng.uniform(5, 100)]
            dts.append({"id": len(dts) + 1, "image_id": img, "category_id": 1, "score": float(np.round(rng.random(), 2)), "area": b[2] * b[3], "bbox": b})
    ref = R.CocoEvalRef(gts, dts, img_ids, "bbox")
    ref.evaluate()
    ref.accumulate()
    want = ref.summarize()
    # feed the product's accumulate with the oracle's per-image match tables (the matching itself is a GPU kernel)
    ev = C.IouTypeEval("bbox")
    A, I = len(R.AREA_RNG), len(img_ids)
    for i, img in enumerate(img_ids):
        per_range = [ref.eval_imgs[a * I + i] for a in range(A)]
        if per_range[0] is None:
            ev.records[img] = {"scores": np.zeros(0), "dt_match": np.zeros((A, 10, 0), np.int32), "dt_ignore": np.zeros((A, 10, 0), bool),
                               "gt_ignore": np.zeros((A, 0), bool)}
            continue
        ev.records[img] = {"scores": np.array(per_range[0]["dtScores"]),
                           "dt_match": np.stack([np.where(e["dtMatches"] > 0, 0, -1) for e in per_range]).astype(np.int32),
                           "dt_ignore": np.stack([e["dtIgnore"].astype(bool) for e in per_range]),
                           "gt_ignore": np.stack([e["gtIgnore"].astype(bool) for e in per_range])}
    ev.accumulate()
    got = ev.summarize(verbose=False)
    assert np.array_equal(got, want), (got, want)
    assert np.array_equal(ev.eval["precision"][:, :, 0], ref.precision[:, :, 0]) and np.array_equal(ev.eval["recall"][:, 0], ref.recall[:, 0])
    assert want[0] > 0.05
