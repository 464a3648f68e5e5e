"""Oracle restatement of the evaluation path (test infrastructure; see oracle/__init__.py).

What it follows:
  * /root/reference/models/postprocessors.py:59-109 (PostProcessSegm: two bilinear resizes, sigmoid, threshold);
  * /root/reference/datasets/coco_eval.py:167-404 (TDODCocoEvaluator: per-batch evaluate, merge, accumulate, summarize);
  * the third-party arithmetic those call, which is NOT under /root/reference: pycocotools, pinned by
    requirements.txt:56 to cocoapi @ 8c9bcc3cf640524c4c20a9c40e89cb6a2f2fa0e9 (PythonAPI/pycocotools/cocoeval.py and
    common/maskApi.c).  Restated here from its published algorithm: column-major run-length masks (rleEncode, rleArea,
    rleIou with the crowd rule, rleToBbox, the LEB128-like rleToString/rleFrString), bbIou, COCO.loadRes' detection
    fields, COCOeval.evaluateImg's greedy matching, accumulate's monotone precision envelope sampled at 101 recall
    thresholds, and the 12 summary numbers.

PARITY UNPINNED against pycocotools itself: the package is absent from this image (no network), and the reference holds
no golden vectors for this path.  The restatement is pinned only by hand-derived known answers and size-independent
properties (tests/test_cpu_coco.py): encode/decode round trips, string round trips, IoU against dense counting, AP = 1 for
perfect detections, a hand-computed precision/recall case.
"""
import numpy as np


# ---- maskApi.c -------------------------------------------------------------------------------------------------
def rle_encode(mask):
    """rleEncode: mask [h, w] (any integer/bool type) -> counts (zeros first), pixels taken column by column."""
    flat = np.asarray(mask).astype(np.uint8).T.reshape(-1)          # column-major order
    counts, run, prev = [], 0, 0
    for v in flat.tolist():
        if v != prev:
            counts.append(run)
            run, prev = 0, v
        run += 1
    counts.append(run)
    return counts


def rle_decode(counts, h, w):
    flat = np.zeros(h * w, dtype=np.uint8)
    pos, val = 0, 0
    for c in counts:
        flat[pos:pos + c] = val
        pos += c
        val ^= 1
    return flat.reshape(w, h).T.copy()


def rle_area(counts):
    return int(sum(counts[1::2]))


def rle_to_bbox(counts, h, w):
    """rleToBbox: [x, y, w, h] of the foreground (zeros when empty)."""
    m = len(counts) // 2 * 2
    if m == 0:
        return [0.0, 0.0, 0.0, 0.0]
    xs, ys, xe, ye, cc = w, h, 0, 0, 0
    xp = 0
    for j in range(m):
        cc += counts[j]
        t = cc - j % 2
        y, x = t % h, (t - t % h) // h
        if j % 2 == 0:
            xp = x
        elif xp < x:
            ys, ye = 0, h - 1
        xs, xe, ys, ye = min(xs, x), max(xe, x), min(ys, y), max(ye, y)
    return [float(xs), float(ys), float(xe - xs + 1), float(ye - ys + 1)]


def rle_to_string(counts):
    """rleToString: difference against the count two back (from the 4th on), 5 bits per character, continuation bit
    0x20, sign carried by bit 0x10 of the last group, characters offset by 48."""
    out = []
    for i, c in enumerate(counts):
        x = int(c)
        if i > 2:
            x -= int(counts[i - 2])
        more = True
        while more:
            ch = x & 0x1F
            x >>= 5
            more = (x != -1) if (ch & 0x10) else (x != 0)
            if more:
                ch |= 0x20
            out.append(chr(ch + 48))
    return "".join(out)


def rle_from_string(s):
    counts, p = [], 0
    while p < len(s):
        x, k, more = 0, 0, True
        while more:
            ch = ord(s[p]) - 48
            x |= (ch & 0x1F) << (5 * k)
            more = bool(ch & 0x20)
            p += 1
            k += 1
            if not more and (ch & 0x10):
                x |= -1 << (5 * k)
        if len(counts) > 2:
            x += counts[-2]
        counts.append(x)
    return counts


def rle_iou(dt, gt, iscrowd):
    """rleIou by merging runs: dt, gt lists of count lists over the same h*w; -> [len(dt), len(gt)] float64
    (crowd ground truth: union = detection area)."""
    out = np.zeros((len(dt), len(gt)), dtype=np.float64)
    for g, gc in enumerate(gt):
        for d, dc in enumerate(dt):
            ka, kb = len(dc), len(gc)
            ca, cb = dc[0], gc[0]
            va = vb = 0
            a = b = 1
            i = u = 0
            ct = 1
            while ct > 0:
                c = min(ca, cb)
                if va or vb:
                    u += c
                    if va and vb:
                        i += c
                ct = 0
                ca -= c
                if not ca and a < ka:
                    ca = dc[a]
                    a += 1
                    va ^= 1
                ct += ca
                cb -= c
                if not cb and b < kb:
                    cb = gc[b]
                    b += 1
                    vb ^= 1
                ct += cb
            if i == 0:
                u = 1
            elif iscrowd[g]:
                u = rle_area(dc)
            out[d, g] = i / u
    return out


def bb_iou(dt, gt, iscrowd):
    """bbIou: boxes [x, y, w, h] float64 -> [len(dt), len(gt)]."""
    out = np.zeros((len(dt), len(gt)), dtype=np.float64)
    for g, G in enumerate(gt):
        ga = G[2] * G[3]
        for d, D in enumerate(dt):
            da = D[2] * D[3]
            w = min(D[2] + D[0], G[2] + G[0]) - max(D[0], G[0])
            if w <= 0:
                continue
            h = min(D[3] + D[1], G[3] + G[1]) - max(D[1], G[1])
            if h <= 0:
                continue
            i = w * h
            u = da if iscrowd[g] else da + ga - i
            out[d, g] = i / u
    return out


# ---- cocoeval.py -----------------------------------------------------------------------------------------------
IOU_THRS = np.linspace(.5, 0.95, int(np.round((0.95 - .5) / .05)) + 1, endpoint=True)
REC_THRS = np.linspace(.0, 1.00, int(np.round((1.00 - .0) / .01)) + 1, endpoint=True)
MAX_DETS = [1, 10, 100]
AREA_RNG = [[0 ** 2, 1e5 ** 2], [0 ** 2, 32 ** 2], [32 ** 2, 96 ** 2], [96 ** 2, 1e5 ** 2]]
AREA_LBL = ["all", "small", "medium", "large"]


class CocoEvalRef:
    """COCOeval over one category list, fed with plain dicts.

    gts: list of {"id", "image_id", "category_id", "iscrowd", "area", "bbox" [x,y,w,h], "counts" (segm), optional "ignore"}
    dts: list of {"id", "image_id", "category_id", "score", "area", "bbox" or "counts"}
    (COCO.loadRes gives detections id = 1-based position in the results list, area = w*h or the mask area.)"""

    def __init__(self, gts, dts, img_ids, iou_type, cat_ids=(1,), sizes=None):
        self.iou_type, self.img_ids, self.cat_ids, self.sizes = iou_type, sorted(set(img_ids)), sorted(set(cat_ids)), sizes
        self.gts, self.dts = {}, {}
        for g in gts:
            g = dict(g)
            g["ignore"] = bool(g.get("iscrowd", 0))                   # _prepare: ignore := iscrowd
            self.gts.setdefault((g["image_id"], g["category_id"]), []).append(g)
        for d in dts:
            self.dts.setdefault((d["image_id"], d["category_id"]), []).append(dict(d))

    def compute_iou(self, img, cat):
        gt, dt = self.gts.get((img, cat), []), self.dts.get((img, cat), [])
        if not gt and not dt:
            return []
        order = np.argsort([-d["score"] for d in dt], kind="mergesort")
        dt = [dt[i] for i in order][:MAX_DETS[-1]]
        crowd = [int(g["iscrowd"]) for g in gt]
        if self.iou_type == "segm":
            return rle_iou([d["counts"] for d in dt], [g["counts"] for g in gt], crowd)
        return bb_iou([d["bbox"] for d in dt], [g["bbox"] for g in gt], crowd)

    def evaluate_img(self, img, cat, rng, max_det):
        gt, dt = self.gts.get((img, cat), []), self.dts.get((img, cat), [])
        if not gt and not dt:
            return None
        for g in gt:
            g["_ignore"] = 1 if (g["ignore"] or g["area"] < rng[0] or g["area"] > rng[1]) else 0
        gtind = np.argsort([g["_ignore"] for g in gt], kind="mergesort")
        gt = [gt[i] for i in gtind]
        dtind = np.argsort([-d["score"] for d in dt], kind="mergesort")
        dt = [dt[i] for i in dtind[:max_det]]
        crowd = [int(g["iscrowd"]) for g in gt]
        ious = self.ious[img, cat]
        ious = ious[:, gtind] if len(ious) > 0 else ious
        T, G, D = len(IOU_THRS), len(gt), len(dt)
        gtm, dtm = np.zeros((T, G)), np.zeros((T, D))
        gt_ig = np.array([g["_ignore"] for g in gt])
        dt_ig = np.zeros((T, D))
        if len(ious) != 0:
            for ti, t in enumerate(IOU_THRS):
                for di, d in enumerate(dt):
                    best, m = min([t, 1 - 1e-10]), -1
                    for gi in range(G):
                        if gtm[ti, gi] > 0 and not crowd[gi]:
                            continue
                        if m > -1 and gt_ig[m] == 0 and gt_ig[gi] == 1:
                            break
                        if ious[di, gi] < best:
                            continue
                        best, m = ious[di, gi], gi
                    if m == -1:
                        continue
                    dt_ig[ti, di] = gt_ig[m]
                    dtm[ti, di] = gt[m]["id"]
                    gtm[ti, m] = d["id"]
        out_of_range = np.array([d["area"] < rng[0] or d["area"] > rng[1] for d in dt]).reshape((1, D))
        dt_ig = np.logical_or(dt_ig, np.logical_and(dtm == 0, np.repeat(out_of_range, T, 0)))
        return {"dtMatches": dtm, "gtMatches": gtm, "dtScores": [d["score"] for d in dt], "gtIgnore": gt_ig, "dtIgnore": dt_ig}

    def evaluate(self):
        self.ious = {(i, c): self.compute_iou(i, c) for i in self.img_ids for c in self.cat_ids}
        self.eval_imgs = [self.evaluate_img(i, c, r, MAX_DETS[-1]) for c in self.cat_ids for r in AREA_RNG for i in self.img_ids]

    def accumulate(self):
        T, R, K, A, M = len(IOU_THRS), len(REC_THRS), len(self.cat_ids), len(AREA_RNG), len(MAX_DETS)
        I = len(self.img_ids)
        precision, recall = -np.ones((T, R, K, A, M)), -np.ones((T, K, A, M))
        for k in range(K):
            for a in range(A):
                for m, max_det in enumerate(MAX_DETS):
                    E = [e for e in (self.eval_imgs[k * A * I + a * I + i] for i in range(I)) if e is not None]
                    if not E:
                        continue
                    scores = np.concatenate([e["dtScores"][:max_det] for e in E])
                    inds = np.argsort(-scores, kind="mergesort")
                    dtm = np.concatenate([e["dtMatches"][:, :max_det] for e in E], axis=1)[:, inds]
                    dt_ig = np.concatenate([e["dtIgnore"][:, :max_det] for e in E], axis=1)[:, inds]
                    gt_ig = np.concatenate([e["gtIgnore"] for e in E])
                    npig = np.count_nonzero(gt_ig == 0)
                    if npig == 0:
                        continue
                    tps = np.logical_and(dtm, np.logical_not(dt_ig))
                    fps = np.logical_and(np.logical_not(dtm), np.logical_not(dt_ig))
                    tp_sum = np.cumsum(tps, axis=1).astype(dtype=float)
                    fp_sum = np.cumsum(fps, axis=1).astype(dtype=float)
                    for t, (tp, fp) in enumerate(zip(tp_sum, fp_sum)):
                        nd = len(tp)
                        rc = tp / npig
                        pr = (tp / (fp + tp + np.spacing(1))).tolist()
                        q = np.zeros(R).tolist()
                        recall[t, k, a, m] = rc[-1] if nd else 0
                        for i in range(nd - 1, 0, -1):
                            if pr[i] > pr[i - 1]:
                                pr[i - 1] = pr[i]
                        for ri, pi in enumerate(np.searchsorted(rc, REC_THRS, side="left")):
                            if pi >= nd:
                                break
                            q[ri] = pr[pi]
                        precision[t, :, k, a, m] = np.array(q)
        self.precision, self.recall = precision, recall

    def summarize(self):
        def one(ap, iou_thr=None, area="all", max_dets=100):
            a, m = AREA_LBL.index(area), MAX_DETS.index(max_dets)
            s = self.precision[:, :, :, a, m] if ap else self.recall[:, :, a, m]
            if iou_thr is not None:
                s = s[np.where(iou_thr == IOU_THRS)[0]]
            return -1 if len(s[s > -1]) == 0 else float(np.mean(s[s > -1]))
        self.stats = np.array([one(1), one(1, .5), one(1, .75), one(1, area="small"), one(1, area="medium"), one(1, area="large"),
                               one(0, max_dets=1), one(0, max_dets=10), one(0), one(0, area="small"), one(0, area="medium"),
                               one(0, area="large")])
        return self.stats


# ---- loadRes / prepare_for_coco_* -------------------------------------------------------------------------------
def detections_from_results(res, iou_type, threshold=0.5):
    """coco_eval.py:291-345 then COCO.loadRes: res {image_id: {"scores","labels","boxes" xyxy,"masks" [Q,1,H,W]}} ->
    detection dicts, numbered from 1 in the order the reference lists them."""
    out = []
    for img, pred in res.items():
        scores, labels = np.asarray(pred["scores"], dtype=np.float64), np.asarray(pred["labels"])
        if iou_type == "bbox":
            b = np.asarray(pred["boxes"], dtype=np.float64)
            for k in range(len(b)):
                box = [b[k, 0], b[k, 1], b[k, 2] - b[k, 0], b[k, 3] - b[k, 1]]
                out.append({"image_id": img, "category_id": int(labels[k]), "bbox": box, "score": float(scores[k]), "area": box[2] * box[3]})
        else:
            masks = np.asarray(pred["masks"]) > threshold
            for k in range(len(masks)):
                counts = rle_encode(masks[k, 0])
                h, w = masks[k, 0].shape
                out.append({"image_id": img, "category_id": int(labels[k]), "counts": counts, "score": float(scores[k]),
                            "area": rle_area(counts), "bbox": rle_to_bbox(counts, h, w)})
    for i, d in enumerate(out):
        d["id"] = i + 1
    return out


def postprocess_segm(pred_masks, orig_sizes, max_sizes, threshold=0.5):
    """postprocessors.py:73-109 on the CPU in fp32: pred_masks [B,Q,1,h,w] (torch) -> list of bool [Q,1,H_i,W_i]."""
    import torch
    import torch.nn.functional as F
    max_h, max_w = max_sizes.max(0)[0].tolist()
    masks = F.interpolate(pred_masks.squeeze(2).float(), size=(max_h, max_w), mode="bilinear", align_corners=False)
    out = []
    for m, t, tt in zip(masks, max_sizes, orig_sizes):
        crop = m[:, :int(t[0]), :int(t[1])].unsqueeze(1)
        out.append(F.interpolate(crop, size=tuple(tt.tolist()), mode="bilinear").sigmoid() > threshold)
    return out
