"""COCO evaluation of the detection / segmentation outputs (reference: /root/reference/datasets/coco_eval.py:167-404,
TDODCocoEvaluator, driven by engine.py:307-342).

The reference hands every batch to pycocotools on the host: Q dense masks per image are copied off the GPU, run-length
encoded, intersected run by run, then matched in pure-Python loops.  Here the per-batch work stays on the MI355X
(csrc/evalmask.hip): masks are column-major bit planes (from PostProcessSegm(packed=True), or packed from dense masks),
areas and intersections are popcounts, IoUs are formed in double exactly as maskApi's rleIou / bbIou do, and
COCOeval.evaluateImg's greedy matching runs as one thread per (image, area range, IoU threshold).  Only the per-image match
tables travel to the host; accumulate() / summarize() (once per evaluation, a few thousand numbers) are numpy.

pycocotools is a third-party package that is absent from /root/reference and from this image (requirements.txt:56 pins
cocoapi @ 8c9bcc3); its published algorithm is restated in oracle/coco_ref.py, which tests/ hold this module against.
There is no CPU path: without the HIP library, or with host tensors, the kernels raise."""
import numpy as np
import torch

from . import dist as tdist
from . import kernels as k

IOU_THRS = np.linspace(.5, 0.95, int(np.round((0.95 - .5) / .05)) + 1, endpoint=True)
REC_THRS = np.linspace(.0, 1.00, int(np.round((1.00 - .0) / .01)) + 1, endpoint=True)
MAX_DETS = (1, 10, 100)
AREA_RNG = np.array([[0 ** 2, 1e5 ** 2], [0 ** 2, 32 ** 2], [32 ** 2, 96 ** 2], [96 ** 2, 1e5 ** 2]], dtype=np.float64)
AREA_LBL = ("all", "small", "medium", "large")


# ---- RLE text form (maskApi.c rleToString / rleFrString) -------------------------------------------------------
def counts_to_string(counts):
    c = np.asarray(counts, dtype=np.int64)
    x = c.copy()
    x[3:] -= c[1:-2]                                        # from the 4th count on: difference to the count two back
    out = bytearray()
    for v in x.tolist():
        while True:
            ch = v & 0x1F
            v >>= 5
            more = (v != -1) if (ch & 0x10) else (v != 0)
            out.append((ch | 0x20 if more else ch) + 48)
            if not more:
                break
    return out.decode("ascii")


def string_to_counts(s):
    if isinstance(s, bytes):
        s = s.decode("ascii")
    counts, p, n = [], 0, len(s)
    while p < n:
        x, shift = 0, 0
        while True:
            ch = ord(s[p]) - 48
            p += 1
            x |= (ch & 0x1F) << shift
            shift += 5
            if not ch & 0x20:
                if ch & 0x10:
                    x |= -1 << shift
                break
        if len(counts) > 2:
            x += counts[-2]
        counts.append(x)
    return counts


def counts_to_mask(counts, h, w):
    """dense bool [h, w] of a column-major run-length list (zeros first)."""
    c = np.asarray(counts, dtype=np.int64)
    if int(c.sum()) != h * w:
        raise ValueError(f"run lengths sum to {int(c.sum())}, the mask has {h * w} pixels")
    vals = np.zeros(len(c), dtype=bool)
    vals[1::2] = True
    return np.repeat(vals, c).reshape(w, h).T


def polygon_to_mask(xy, h, w):
    """maskApi.c rleFrPoly restated: polygon edges are traced on a 5x finer integer grid, the crossings of pixel-column
    boundaries become run boundaries of the column-major mask.  (Host-side ground-truth preparation, once per annotation.)"""
    scale = 5.0
    pts = np.asarray(xy, dtype=np.float64).reshape(-1, 2)
    x = [int(scale * px + .5) for px in pts[:, 0]]
    y = [int(scale * py + .5) for py in pts[:, 1]]
    x.append(x[0])
    y.append(y[0])
    u, v = [], []
    for j in range(len(pts)):
        xs, xe, ys, ye = x[j], x[j + 1], y[j], y[j + 1]
        dx, dy = abs(xe - xs), abs(ys - ye)
        flip = (dx >= dy and xs > xe) or (dx < dy and ys > ye)
        if flip:
            xs, xe, ys, ye = xe, xs, ye, ys
        if dx >= dy:
            s = (ye - ys) / dx if dx else 0.0
            for d in range(dx + 1):
                t = dx - d if flip else d
                u.append(t + xs)
                v.append(int(ys + s * t + .5))
        else:
            s = (xe - xs) / dy
            for d in range(dy + 1):
                t = dy - d if flip else d
                v.append(t + ys)
                u.append(int(xs + s * t + .5))
    bx, by = [], []
    for j in range(1, len(u)):
        if u[j] == u[j - 1]:
            continue
        xd = float(u[j] if u[j] < u[j - 1] else u[j] - 1)
        xd = (xd + .5) / scale - .5
        if np.floor(xd) != xd or xd < 0 or xd > w - 1:
            continue
        yd = float(v[j] if v[j] < v[j - 1] else v[j - 1])
        yd = (yd + .5) / scale - .5
        yd = min(max(yd, 0.0), float(h))
        bx.append(int(xd))
        by.append(int(np.ceil(yd)))
    a = np.sort(np.array([px * h + py for px, py in zip(bx, by)] + [h * w], dtype=np.int64))
    runs = np.diff(np.concatenate([[0], a]))
    counts, j = [int(runs[0])], 1
    while j < len(runs):                                      # zero-length runs glue their neighbours together
        if runs[j] > 0:
            counts.append(int(runs[j]))
            j += 1
        else:
            j += 1
            if j < len(runs):
                counts[-1] += int(runs[j])
                j += 1
    return counts_to_mask(counts, h, w)


def convert_coco_poly_to_mask(segmentations, height, width):
    """datasets/tdod.py:133-147 (and datasets/coco.py:66-80): one bool mask per object, the union of its polygons
    (coco_mask.frPyObjects + decode + any(dim=2) in the reference)."""
    masks = []
    for polygons in segmentations:
        m = np.zeros((height, width), dtype=bool)
        for poly in polygons:
            m |= polygon_to_mask(poly, height, width)
        masks.append(torch.from_numpy(m))
    return torch.stack(masks, dim=0) if masks else torch.zeros((0, height, width), dtype=torch.uint8)


# ---- ground truth ----------------------------------------------------------------------------------------------
class CocoGroundTruth:
    """The slice of pycocotools.coco.COCO the evaluator reads: a COCO-format dict {"images": [{"id","height","width"}],
    "annotations": [{"id","image_id","category_id","bbox" [x,y,w,h],"area","iscrowd","segmentation"}]}.
    A segmentation is a list of polygons, an RLE dict {"size": [h, w], "counts": list | str} or a dense [h, w] array."""

    def __init__(self, dataset):
        self.dataset = dataset
        self.imgs = {im["id"]: im for im in dataset.get("images", [])}
        self.img_anns = {i: [] for i in self.imgs}
        for ann in dataset.get("annotations", []):
            if ann["id"] == 0:
                raise ValueError("annotation ids must be non-zero: COCOeval stores the matched id and reads 0 as 'unmatched'")
            self.img_anns.setdefault(ann["image_id"], []).append(ann)
        self._planes = {}

    def annotations(self, img_id, cat_ids):
        return [a for a in self.img_anns.get(img_id, []) if cat_ids is None or a["category_id"] in cat_ids]

    def dense_mask(self, ann):
        """COCO.annToMask: polygons are merged by union."""
        im = self.imgs[ann["image_id"]]
        h, w = im["height"], im["width"]
        seg = ann["segmentation"]
        if isinstance(seg, dict):
            counts = string_to_counts(seg["counts"]) if isinstance(seg["counts"], (str, bytes)) else seg["counts"]
            return counts_to_mask(counts, seg["size"][0], seg["size"][1])
        if isinstance(seg, (list, tuple)):
            out = np.zeros((h, w), dtype=bool)
            for poly in seg:
                out |= polygon_to_mask(poly, h, w)
            return out
        return np.asarray(seg).astype(bool)

    def planes(self, img_id, cat_ids, device):
        """Bit planes of an image's ground-truth masks (packed once, kept on the device)."""
        key = (img_id, None if cat_ids is None else tuple(cat_ids))
        if key not in self._planes:
            anns = self.annotations(img_id, cat_ids)
            dense = np.stack([self.dense_mask(a) for a in anns]) if anns else None
            self._planes[key] = None if dense is None else k.mask_pack(torch.from_numpy(dense).to(device))
        return self._planes[key]


# ---- the evaluator -----------------------------------------------------------------------------------------------
class IouTypeEval:
    """What the reference reads from a pycocotools COCOeval: .eval (precision / recall tables) and .stats."""

    def __init__(self, iou_type):
        self.iouType = iou_type
        self.records = {}                 # image id -> per-image match tables
        self.eval, self.stats = None, None

    def accumulate(self):
        T, R, A, M = len(IOU_THRS), len(REC_THRS), len(AREA_RNG), len(MAX_DETS)
        precision, recall, scores = -np.ones((T, R, 1, A, M)), -np.ones((T, 1, A, M)), -np.ones((T, R, 1, A, M))
        recs = [self.records[i] for i in sorted(self.records)]
        recs = [r for r in recs if r["scores"].size or r["gt_ignore"].shape[1]]           # images with neither are skipped
        for a in range(A):
            if not recs:
                continue
            npig = int(sum((~r["gt_ignore"][a]).sum() for r in recs))
            if npig == 0:
                continue
            for m, max_det in enumerate(MAX_DETS):
                sc = np.concatenate([r["scores"][:max_det] for r in recs])
                order = np.argsort(-sc, kind="mergesort")
                sc = sc[order]
                matched = np.concatenate([r["dt_match"][a][:, :max_det] >= 0 for r in recs], axis=1)[:, order]
                ignored = np.concatenate([r["dt_ignore"][a][:, :max_det] for r in recs], axis=1)[:, order]
                tp = np.cumsum(matched & ~ignored, axis=1).astype(np.float64)
                fp = np.cumsum(~matched & ~ignored, axis=1).astype(np.float64)
                nd = tp.shape[1]
                if nd == 0:
                    recall[:, 0, a, m], precision[:, :, 0, a, m], scores[:, :, 0, a, m] = 0, 0, 0
                    continue
                rc = tp / npig
                pr = tp / (fp + tp + np.spacing(1))
                pr = np.maximum.accumulate(pr[:, ::-1], axis=1)[:, ::-1]                 # monotone precision envelope
                recall[:, 0, a, m] = rc[:, -1]
                for t in range(T):
                    at = np.searchsorted(rc[t], REC_THRS, side="left")
                    ok = at < nd
                    precision[t, :, 0, a, m] = np.where(ok, pr[t][np.minimum(at, nd - 1)], 0.0)
                    scores[t, :, 0, a, m] = np.where(ok, sc[np.minimum(at, nd - 1)], 0.0)
        self.eval = {"precision": precision, "recall": recall, "scores": scores, "counts": [T, R, 1, A, M]}

    def summarize(self, verbose=True):
        if self.eval is None:
            raise RuntimeError("Please run accumulate() first")

        def one(ap, iou_thr=None, area="all", max_dets=100):
            a, m = AREA_LBL.index(area), MAX_DETS.index(max_dets)
            s = self.eval["precision"][:, :, :, a, m] if ap else self.eval["recall"][:, :, a, m]
            if iou_thr is not None:
                s = s[np.where(iou_thr == IOU_THRS)[0]]
            val = -1.0 if len(s[s > -1]) == 0 else float(np.mean(s[s > -1]))
            if verbose:
                rng = "{:0.2f}:{:0.2f}".format(IOU_THRS[0], IOU_THRS[-1]) if iou_thr is None else "{:0.2f}".format(iou_thr)
                print(" {:<18} {} @[ IoU={:<9} | area={:>6s} | maxDets={:>3d} ] = {:0.3f}".format(
                    "Average Precision" if ap else "Average Recall", "(AP)" if ap else "(AR)", rng, area, max_dets, val))
            return val
        self.stats = np.array([one(1), one(1, .5), one(1, .75), one(1, area="small"), one(1, area="medium"), one(1, area="large"),
                               one(0, max_dets=1), one(0, max_dets=10), one(0), one(0, area="small"), one(0, area="medium"),
                               one(0, area="large")])
        return self.stats


class TDODCocoEvaluator:
    """Same surface as the reference class (coco_eval.py:167-345): update(res), synchronize_between_processes(), accumulate(),
    summarize(), .coco_eval[iou_type].stats.  `coco_gt` is a CocoGroundTruth (or the COCO-format dict itself).  The reference
    evaluates category 1 only (coco_eval.py:203: params.catIds = 1, and PostProcess labels every query 1)."""

    def __init__(self, coco_gt, iou_types, useCats=True, device="cuda"):
        assert isinstance(iou_types, (list, tuple))
        self.coco_gt = coco_gt if isinstance(coco_gt, CocoGroundTruth) else CocoGroundTruth(coco_gt)
        self.iou_types, self.useCats = list(iou_types), useCats
        self.cat_ids = (1,) if useCats else None
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("TDODCocoEvaluator runs its IoU and matching kernels on the GPU; there is no CPU path")
        self.coco_eval = {t: IouTypeEval(t) for t in iou_types}
        self.img_ids = []
        self._thr = torch.from_numpy(IOU_THRS).to(self.device)
        self._rng = torch.from_numpy(AREA_RNG).to(self.device)

    # -- per batch ---------------------------------------------------------------------------------------------------
    def _detections(self, pred):
        """score-descending (stable) order of the detections of the evaluated category, cut to maxDets[-1]."""
        scores = pred["scores"].to(self.device)
        keep = torch.arange(scores.numel(), device=self.device)
        if self.cat_ids is not None:
            labels = pred["labels"].to(self.device)
            keep = keep[(labels[:, None] == torch.tensor(self.cat_ids, device=self.device)[None]).any(1)]
        order = torch.sort(scores[keep].double(), descending=True, stable=True).indices[:MAX_DETS[-1]]
        return keep[order]

    def update(self, predictions):
        img_ids = sorted(set(predictions.keys()))
        self.img_ids.extend(img_ids)
        for iou_type in self.iou_types:
            if iou_type not in ("bbox", "segm"):
                raise ValueError("Unknown iou type {}".format(iou_type))
            self._evaluate(iou_type, img_ids, predictions)

    def _evaluate(self, iou_type, img_ids, predictions):
        dev = self.device
        ious, dt_area, gt_area, gt_crowd, scores, n_dt, n_gt = [], [], [], [], [], [], []
        for img in img_ids:
            pred = predictions[img]
            anns = self.coco_gt.annotations(img, self.cat_ids)
            crowd = torch.tensor([int(a.get("iscrowd", 0)) for a in anns], dtype=torch.uint8, device=dev)
            sel = self._detections(pred) if len(pred) else torch.zeros(0, dtype=torch.int64, device=dev)
            D, G = int(sel.numel()), len(anns)
            if iou_type == "bbox":
                b = pred["boxes"].to(dev)[sel].float() if D else torch.zeros(0, 4, device=dev)
                xywh = torch.stack((b[:, 0], b[:, 1], b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]), dim=1).double()   # convert_to_xywh in fp32
                area = xywh[:, 2] * xywh[:, 3]
                g = torch.tensor([a["bbox"] for a in anns], dtype=torch.float64, device=dev).view(-1, 4)
                iou = _box_iou(xywh, g, crowd)
            else:
                if D and "mask_bits" in pred:
                    h, w = pred["mask_size"]
                    planes = pred["mask_bits"].to(dev)[sel]
                elif D:
                    dense = pred["masks"].to(dev)
                    h, w = dense.shape[-2:]
                    planes = k.mask_pack((dense[sel, 0] > 0.5) if dense.dtype != torch.bool else dense[sel, 0])
                if D:
                    area_i = k.mask_area(planes, h, w)
                    area = area_i.double()
                else:
                    area = torch.zeros(0, dtype=torch.float64, device=dev)
                iou = torch.zeros(D, G, dtype=torch.float64, device=dev)
                if D and G:
                    gp = self.coco_gt.planes(img, self.cat_ids, dev)
                    im = self.coco_gt.imgs[img]
                    if (im["height"], im["width"]) != (h, w):
                        raise ValueError(f"image {img}: predicted masks are {h}x{w}, the ground truth is {im['height']}x{im['width']}")
                    iou = k.mask_iou(planes, gp, crowd, area_i, k.mask_area(gp, h, w), h, w)
            ious.append(iou.reshape(-1))
            dt_area.append(area)
            gt_area.append(torch.tensor([float(a["area"]) for a in anns], dtype=torch.float64, device=dev))
            gt_crowd.append(crowd)
            scores.append(pred["scores"].to(dev)[sel].double() if D else torch.zeros(0, dtype=torch.float64, device=dev))
            n_dt.append(D)
            n_gt.append(G)
        offs = lambda sizes: torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int64, device=dev)
        A, T = len(AREA_RNG), len(IOU_THRS)
        crowd_all = torch.cat(gt_crowd) if gt_crowd else torch.zeros(0, dtype=torch.uint8, device=dev)
        dt_match, dt_ignore, gt_flag = k.coco_match(torch.cat(ious), offs([d * g for d, g in zip(n_dt, n_gt)]), torch.cat(dt_area), offs(n_dt),
                                                    torch.cat(gt_area), crowd_all, crowd_all, offs(n_gt), self._rng, self._thr)
        dt_match, dt_ignore, gt_flag = dt_match.cpu().numpy(), dt_ignore.cpu().numpy().astype(bool), gt_flag.cpu().numpy().astype(bool)
        scores_h = torch.cat(scores).cpu().numpy()
        d0 = g0 = 0
        for img, D, G in zip(img_ids, n_dt, n_gt):
            rec = {"image_id": img, "scores": scores_h[d0:d0 + D],
                   "dt_match": dt_match[A * T * d0:A * T * (d0 + D)].reshape(A, T, D),
                   "dt_ignore": dt_ignore[A * T * d0:A * T * (d0 + D)].reshape(A, T, D),
                   "gt_ignore": gt_flag[A * g0:A * (g0 + G)].reshape(A, G)}
            self.coco_eval[iou_type].records.setdefault(img, rec)                         # first evaluation of an image wins (merge: np.unique)
            d0, g0 = d0 + D, g0 + G

    # -- whole evaluation --------------------------------------------------------------------------------------------
    def synchronize_between_processes(self):
        """coco_eval.py:345-373: gather every rank's per-image tables, keep one entry per image id."""
        if not tdist.is_dist_avail_and_initialized() or tdist.get_world_size() == 1:
            return
        import torch.distributed as td
        for ev in self.coco_eval.values():
            gathered = [None] * tdist.get_world_size()
            td.all_gather_object(gathered, ev.records)
            merged = {}
            for part in gathered:
                for img, rec in part.items():
                    merged.setdefault(img, rec)
            ev.records = merged
        self.img_ids = sorted(set(i for ev in self.coco_eval.values() for i in ev.records))

    def accumulate(self):
        for ev in self.coco_eval.values():
            ev.accumulate()

    def summarize(self, verbose=True):
        for iou_type, ev in self.coco_eval.items():
            if verbose:
                print("IoU metric: {}".format(iou_type))
            ev.summarize(verbose)

    # -- result export (prepare_for_coco_segmentation, coco_eval.py:307-332) -------------------------------------------
    def segmentation_results(self, predictions):
        """[{"image_id","category_id","segmentation": {"size","counts" (compressed string)},"score"}] with the run lengths
        produced on the device."""
        out = []
        for img, pred in predictions.items():
            if len(pred) == 0:
                continue
            if "mask_bits" in pred:
                (h, w), planes = pred["mask_size"], pred["mask_bits"].to(self.device)
            else:
                dense = pred["masks"].to(self.device)
                h, w = dense.shape[-2:]
                planes = k.mask_pack(dense[:, 0] > 0.5 if dense.dtype != torch.bool else dense[:, 0])
            counts, first = k.mask_rle(planes, h, w)
            counts, first = counts.cpu().numpy(), first.cpu().numpy()
            scores, labels = pred["scores"].tolist(), pred["labels"].tolist()
            for q in range(planes.shape[0]):
                rle = {"size": [h, w], "counts": counts_to_string(counts[first[q]:first[q + 1]])}
                out.append({"image_id": img, "category_id": labels[q], "segmentation": rle, "score": scores[q]})
        return out


def _box_iou(dt, gt, crowd):
    """maskApi.c bbIou on [x, y, w, h] float64 boxes: [D, G]; crowd ground truth: union = detection area."""
    if dt.shape[0] == 0 or gt.shape[0] == 0:
        return torch.zeros(dt.shape[0], gt.shape[0], dtype=torch.float64, device=dt.device)
    da, ga = dt[:, 2] * dt[:, 3], gt[:, 2] * gt[:, 3]
    w = torch.minimum((dt[:, 2] + dt[:, 0])[:, None], (gt[:, 2] + gt[:, 0])[None]) - torch.maximum(dt[:, 0][:, None], gt[:, 0][None])
    h = torch.minimum((dt[:, 3] + dt[:, 1])[:, None], (gt[:, 3] + gt[:, 1])[None]) - torch.maximum(dt[:, 1][:, None], gt[:, 1][None])
    inter = w * h
    union = torch.where(crowd.bool()[None], da[:, None].expand_as(inter), da[:, None] + ga[None] - inter)
    return torch.where((w > 0) & (h > 0), inter / union, torch.zeros_like(inter))
