"""HungarianMatcher on the MI355X (reference: /root/reference/models/matcher.py:16-99).

The cost block and the assignment of every (decoder layer, image) pair are computed by ONE launch of
the wavefront kernel toist_matcher; indices stay on the device for the set criterion.  The public
`forward` keeps the reference contract (list of CPU int64 index pairs) and pays the single host sync
there, on demand.
"""
import torch
from torch import nn

from . import kernels as k

TOKEN_MASK_WORDS = 4      # 64-bit words of a target's token-span mask: 256 tokens = the reference's max_text_len (csrc/contrastive.hip: CA_MW)


class MatchResult:
    """Device-resident assignment for L layers: src/tgt [L, Mtot] int64, per-image slices by `match_off`."""

    def __init__(self, src, tgt, status, sizes, num_queries, tgt_off=None, match_off_dev=None, tgt_boxes=None):
        self.src, self.tgt, self.status = src, tgt, status
        self.tgt_off_dev, self.match_off_dev, self.tgt_boxes = tgt_off, match_off_dev, tgt_boxes
        self.sizes = list(sizes)
        self.counts = [min(num_queries, s) for s in self.sizes]
        self.match_off = [0]
        for c in self.counts:
            self.match_off.append(self.match_off[-1] + c)

    def check(self):
        """Raise like SciPy does for an invalid cost matrix (host sync)."""
        st = self.status.cpu()
        if bool((st == 1).any()):
            raise ValueError("matrix contains invalid numeric entries")
        if bool((st == 2).any()):
            raise ValueError("cost matrix is infeasible")

    def to_list(self, layer=0):
        self.check()
        s, t = self.src[layer].cpu(), self.tgt[layer].cpu()
        return [(s[a:b].clone(), t[a:b].clone()) for a, b in zip(self.match_off[:-1], self.match_off[1:])]


class StaticTargets:
    """Fixed-address device image of a batch's targets, so that ONE captured hipGraph of the training step serves every batch: the
    captured kernels read boxes, positive-map rows, token-span masks, the per-image offsets and num_boxes from these buffers, and
    `load()` refills them (one pinned staging buffer, one asynchronous H2D copy, no host sync) between replays.  Capacity is
    batch * max_per_image targets; the number of targets per image may change from step to step.

    Layout of the arena (bytes): boxes f32 [cap,4] | positive_map f32 [cap,K] | token masks i64 [cap,2] | tgt_off i32 [B+1] |
    match_off i32 [B+1] | num_boxes f32 [1] (local sum of targets; the world mean, clamped to >= 1, is formed on the device) |
    valid_hw i32 [4] (with mask_hw) = {VH, VW, hs, ws}: the largest ground-truth mask of the batch = the size the reference pads the batch's masks to
    and resizes its predictions to (mdetr.py:839-843), and the size of the prediction the reference would have had for that batch (DETRsegm's
    maps have the size of the ResNet C2 feature: ceil(side / 4), `mask_pred_of`); the mask losses map that corner of the bucket's prediction onto that corner of the bucket's targets and
    normalise by VH * VW, so a batch gives the same mask losses in any bucket."""

    def __init__(self, batch, max_per_image, num_queries, K=256, device="cuda", mask_hw=None, mask_pred_of=lambda side: (side + 3) // 4):
        self.B, self.max_per_image, self.Q, self.K = batch, max_per_image, num_queries, K
        # mask_hw = (TH, TW): the ground-truth masks of the targets travel too (uint8 [cap, TH, TW], zero-padded to the padded batch size like
        # NestedTensor.from_tensor_list, util/misc.py:185-209) -- the mask losses of configs[2] then replay from the same graph as the detection losses
        self.mask_hw = None if mask_hw is None else (int(mask_hw[0]), int(mask_hw[1]))
        self.mask_pred_of = mask_pred_of          # image side -> side of pred_masks (the mask head resizes to each FPN level and ends at C2's size: segmentation.py:223-240)
        self.masks = self._mask_host = None
        if self.mask_hw is not None:
            self.masks = torch.zeros(batch * max_per_image, *self.mask_hw, dtype=torch.uint8, device=device)
            self._mask_host = torch.zeros(batch * max_per_image, *self.mask_hw, dtype=torch.uint8).pin_memory()
        self.cap = cap = batch * max_per_image
        self.device = torch.device(device)
        sizes = [cap * 4 * 4, cap * K * 4, cap * TOKEN_MASK_WORDS * 8, (batch + 1) * 4, (batch + 1) * 4, 4, 16]
        offs, total = [], 0
        for n in sizes:
            offs.append(total)
            total += (n + 15) // 16 * 16
        self._host = torch.zeros(total, dtype=torch.uint8).pin_memory()
        self._dev = torch.zeros(total, dtype=torch.uint8, device=self.device)

        def views(buf):
            cut = lambda i, dt, shape: buf[offs[i]:offs[i] + sizes[i]].view(dt).view(shape)
            return (cut(0, torch.float32, (cap, 4)), cut(1, torch.float32, (cap, K)), cut(2, torch.int64, (cap, TOKEN_MASK_WORDS)), cut(3, torch.int32, (batch + 1,)),
                    cut(4, torch.int32, (batch + 1,)), cut(5, torch.float32, (1,)), cut(6, torch.int32, (4,)))

        self._views = views
        self.boxes, self.positive_map, self.tok_mask, self.tgt_off, self.match_off, self._nb_local, self.valid_hw = views(self._dev)
        if self.mask_hw is None:
            self.valid_hw = None
        self.num_boxes = torch.ones(1, dtype=torch.float32, device=self.device)
        self.sizes = [0] * batch
        self._out = {}        # L -> (src [L, cap], tgt [L, cap], status [L*B])
        self._event = None
        self.distill = None   # distill.DistillTables of this side of a (noun, pronoun) pair: the captured distillation step (harness.CapturedDistillStep)

    def pack(self, targets, positive_map, token_masks=None, out=None):
        """Host image of one batch (pinned uint8 tensor in the arena layout, + the per-image target counts): build it ahead of time -- in a
        loader worker -- and hand it to load_packed().  targets: list of dicts with HOST tensors `boxes` [T_i, 4]; positive_map: host
        [sum T_i, K]; token_masks: host int64 [sum T_i, TOKEN_MASK_WORDS] (SetCriterion.token_masks_host) when the contrastive-alignment loss is on."""
        sizes = [int(t["boxes"].shape[0]) for t in targets]
        if len(sizes) != self.B or max(sizes, default=0) > self.max_per_image:
            raise ValueError(f"StaticTargets holds {self.B} images x <= {self.max_per_image} targets; got sizes {sizes}")
        host = out if out is not None else torch.zeros(self._host.numel(), dtype=torch.uint8).pin_memory()
        hb, hp, hm, hto, hmo, hnb, hvalid = self._views(host)
        tot = sum(sizes)
        if tot:
            hb[:tot] = torch.cat([t["boxes"].float().cpu() for t in targets])
            hp[:tot] = positive_map.float().cpu()
            if token_masks is not None:
                hm[:tot] = token_masks
        acc_t, acc_m = 0, 0
        hto[0], hmo[0] = 0, 0
        for i, sz in enumerate(sizes):
            acc_t += sz
            acc_m += min(self.Q, sz)
            hto[i + 1], hmo[i + 1] = acc_t, acc_m
        hnb[0] = float(tot)
        if self.mask_hw is None:
            return host, sizes
        TH, TW = self.mask_hw
        mh = self._mask_host if out is not None else torch.zeros(self.cap, TH, TW, dtype=torch.uint8).pin_memory()
        row, vh, vw = 0, 0, 0
        for t in targets:
            m = t["masks"].to(torch.uint8).cpu()
            n, h, w = m.shape
            if h > TH or w > TW:
                raise ValueError(f"StaticTargets(mask_hw={self.mask_hw}) got a {h} x {w} mask")
            vh, vw = max(vh, h), max(vw, w)         # (an image without targets still carries its [0, h, w] mask tensor)
            mh[row:row + n].zero_()
            mh[row:row + n, :h, :w] = m
            row += n
        vh, vw = (vh or TH), (vw or TW)
        hvalid[0], hvalid[1], hvalid[2], hvalid[3] = vh, vw, self.mask_pred_of(vh), self.mask_pred_of(vw)
        return host, sizes, mh

    def load_packed(self, packed):
        """One asynchronous H2D copy of a pack()ed batch + the device-side num_boxes (mdetr.py:997-1001: all-reduce, / world, clamp >= 1);
        no host sync.  Call between replays of the captured step, on the stream that replays it."""
        host, sizes = packed[0], packed[1]
        self.sizes = list(sizes)
        self._dev.copy_(host, non_blocking=True)
        if self.mask_hw is not None:
            tot = sum(sizes)
            if tot:
                self.masks[:tot].copy_(packed[2][:tot], non_blocking=True)
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            self.num_boxes.copy_(self._nb_local)
            torch.distributed.all_reduce(self.num_boxes)
            self.num_boxes.div_(torch.distributed.get_world_size())
            self.num_boxes.clamp_(min=1)
        else:
            torch.clamp(self._nb_local, min=1, out=self.num_boxes)
        return self

    def load(self, targets, positive_map, token_masks=None):
        """pack() into the internal staging buffer + load_packed()."""
        if self._event is not None:
            self._event.synchronize()          # the previous upload has left the staging buffer
        packed = self.pack(targets, positive_map, token_masks, out=self._host)
        self.load_packed(packed)
        self._event = torch.cuda.Event()
        self._event.record()
        return self

    def outputs(self, L):
        ent = self._out.get(L)
        if ent is None:
            ent = self._out[L] = (torch.zeros(L, self.cap, dtype=torch.int64, device=self.device), torch.zeros(L, self.cap, dtype=torch.int64, device=self.device),
                                  torch.zeros(L * self.B, dtype=torch.int32, device=self.device))
        return ent

    def match_result(self, L):
        """MatchResult of the CURRENT contents (host lists from the last load(), device buffers of the last launch / replay)."""
        src, tgt, status = self.outputs(L)
        mtot = sum(min(self.Q, s_) for s_ in self.sizes)
        flat = lambda t: t.view(-1)[:L * mtot].view(L, mtot)       # the kernels pack the rows with stride match_off[B]
        return MatchResult(flat(src), flat(tgt), status, self.sizes, self.Q, self.tgt_off, self.match_off, self.boxes)


class HungarianMatcher(nn.Module):
    def __init__(self, cost_class: float = 1, cost_bbox: float = 1, cost_giou: float = 1):
        super().__init__()
        self.cost_class, self.cost_bbox, self.cost_giou = cost_class, cost_bbox, cost_giou
        assert cost_class != 0 or cost_bbox != 0 or cost_giou != 0, "all costs cant be 0"
        self._offsets = {}

    @torch.no_grad()
    def match_layers(self, logits, boxes, targets, positive_map):
        """logits [L,B,Q,K], boxes [L,B,Q,4] (any float dtype) -> MatchResult; no host sync."""
        L, B, Q, K = logits.shape
        dev = logits.device
        sizes = [int(t["boxes"].shape[0]) for t in targets]
        assert sum(sizes) == len(positive_map)
        res_counts = [min(Q, s) for s in sizes]
        # prefix sums live on the device; cached per batch signature so a steady-state step issues no
        # host->device copy (and stays hipGraph-capturable)
        key = (tuple(sizes), Q, str(dev))
        ent = self._offsets.get(key)
        if ent is None:
            if len(self._offsets) > 64:
                self._offsets.clear()
            tgt_off = torch.tensor([0] + [sum(sizes[:i + 1]) for i in range(B)], dtype=torch.int32)
            m_off = torch.tensor([0] + [sum(res_counts[:i + 1]) for i in range(B)], dtype=torch.int32)
            ent = (tgt_off.to(dev), m_off.to(dev), int(m_off[-1]), int(tgt_off[-1]))
            self._offsets[key] = ent
        tgt_off, m_off, mtot, ttot = ent
        src = torch.zeros(L, max(mtot, 1), dtype=torch.int64, device=dev)[:, :mtot]
        tgt = torch.zeros(L, max(mtot, 1), dtype=torch.int64, device=dev)[:, :mtot]
        status = torch.zeros(L * B, dtype=torch.int32, device=dev)
        tb = None
        if ttot > 0:
            tb = torch.cat([t["boxes"] for t in targets]).float().contiguous()
            k.matcher(logits.float().contiguous(), boxes.float().contiguous(), tb, positive_map.float().contiguous(),
                      tgt_off, m_off, max(sizes), float(self.cost_class),
                      float(self.cost_bbox), float(self.cost_giou), src if mtot else torch.zeros(L, 1, dtype=torch.int64, device=dev),
                      tgt if mtot else torch.zeros(L, 1, dtype=torch.int64, device=dev), status)
        return MatchResult(src, tgt, status, sizes, Q, tgt_off, m_off, tb)

    @torch.no_grad()
    def match_layers_static(self, logits, boxes, st):
        """match_layers on a StaticTargets image: every shape and pointer is independent of the batch's target counts, so the launch can
        be captured once and replayed for any batch (`st.load(...)` in between)."""
        L, B, Q, K = logits.shape
        assert B == st.B and Q == st.Q and K == st.K
        src, tgt, status = st.outputs(L)
        k.matcher(logits.float().contiguous(), boxes.float().contiguous(), st.boxes, st.positive_map, st.tgt_off, st.match_off, st.max_per_image,
                  float(self.cost_class), float(self.cost_bbox), float(self.cost_giou), src, tgt, status)
        return MatchResult(src, tgt, status, st.sizes, Q, st.tgt_off, st.match_off, st.boxes)

    @torch.no_grad()
    def forward(self, outputs, targets, positive_map):
        """Reference contract (matcher.py:40-87): list over images of (query idx, target idx) CPU int64."""
        res = self.match_layers(outputs["pred_logits"][None], outputs["pred_boxes"][None], targets, positive_map)
        return res.to_list(0)


def build_matcher(args):
    if args.set_loss != "hungarian":
        raise ValueError(f"Only hungarian accepted, got {args.set_loss}")
    return HungarianMatcher(cost_class=args.set_cost_class, cost_bbox=args.set_cost_bbox, cost_giou=args.set_cost_giou)


_LSAP_META = {}


def _lsap_meta(shapes, offsets, dev):
    """Device copies of the per-problem tables, cached per signature (a steady-state step uploads nothing)."""
    key = (tuple(shapes), tuple(offsets), str(dev))
    ent = _LSAP_META.get(key)
    if ent is None:
        if len(_LSAP_META) > 64:
            _LSAP_META.clear()
        pairs = [min(r, c) for r, c in shapes]
        out_off = [sum(pairs[:i]) for i in range(len(pairs))]
        ent = _LSAP_META[key] = (torch.tensor(list(offsets), dtype=torch.int64, device=dev), torch.tensor([r for r, _ in shapes], dtype=torch.int32, device=dev),
                                 torch.tensor([c for _, c in shapes], dtype=torch.int32, device=dev), torch.tensor(out_off, dtype=torch.int64, device=dev), out_off, pairs)
    return ent


_LSAP_PENDING = []      # (pinned host copy of a status vector, event recorded behind the copy)


def _raise_on(st):
    if bool((st == 1).any()):
        raise ValueError("matrix contains invalid numeric entries")
    if bool((st == 2).any()):
        raise ValueError("cost matrix is infeasible")


def check_lsap_status(status, defer=False):
    """Raise like SciPy for invalid / infeasible matrices.  defer=False reads the device now (a host sync).  defer=True stages the status words in
    pinned host memory behind an event and raises for the launches of EARLIER calls whose copies have arrived (the training step's paths: the
    error of a poisoned cost matrix surfaces one call later instead of stalling the launch queue every step, as SetCriterion._note_status
    does for the matcher); check_lsap_pending() waits for everything still in flight."""
    if not defer or not status.is_cuda:
        _raise_on(status.cpu())
        return
    if torch.cuda.is_current_stream_capturing():      # (an event recorded on this stream before the capture may not even be queried now)
        return
    check_lsap_pending(wait=False)
    host = torch.empty(status.shape, dtype=status.dtype, pin_memory=True)
    host.copy_(status, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    _LSAP_PENDING.append((host, ev))


def check_lsap_pending(wait=True):
    keep = []
    try:
        while _LSAP_PENDING:
            host, ev = _LSAP_PENDING.pop(0)
            if not wait and not ev.query():
                keep.append((host, ev))
                continue
            ev.synchronize()
            _raise_on(host)
    finally:
        _LSAP_PENDING[:0] = keep


def lsap_blocks(cost, shapes, offsets, ld):
    """LSAP on blocks of one device buffer: problem p is the [rows, cols] block with row stride `ld` starting at element
    offsets[p] of `cost`.  One launch, no host sync: returns (row_idx, col_idx, out_off, pairs, status) -- the pairs of
    problem p are row_idx[out_off[p] : out_off[p] + pairs[p]]; pass `status` to check_lsap_status when convenient."""
    from . import kernels as k
    dev = cost.device
    off, rows, cols, out_off_dev, out_off, pairs = _lsap_meta(shapes, offsets, dev)
    total = max(sum(pairs), 1)
    ri = torch.zeros(total, dtype=torch.int64, device=dev)
    ci = torch.zeros(total, dtype=torch.int64, device=dev)
    status = torch.zeros(len(shapes), dtype=torch.int32, device=dev)
    k.lsap(cost, off, rows, cols, len(shapes), max(r for r, _ in shapes), max(c for _, c in shapes), max(r * c for r, c in shapes), out_off_dev, ri, ci,
           status, ld=ld)
    return ri, ci, out_off, pairs, status


def linear_sum_assignment_batch(costs, defer_status=False, with_status=False):
    """scipy.optimize.linear_sum_assignment for a list of 2-D fp32 device cost matrices, solved in one launch of the
    HIP LSAP kernel (csrc/matcher.hip).  Returns a list of (row_ind, col_ind) int64 device tensors (rows ascending, as
    SciPy returns them).  Raises ValueError on NaN / -inf entries or an infeasible matrix, like SciPy."""
    if not costs:
        return []
    shapes = [(int(c.shape[0]), int(c.shape[1])) for c in costs]
    sizes = [r * c for r, c in shapes]
    flat = torch.cat([c.reshape(-1).float() for c in costs]) if sum(sizes) else torch.zeros(1, device=costs[0].device)
    ri, ci, out_off, pairs, status = lsap_blocks(flat.contiguous(), shapes, [sum(sizes[:i]) for i in range(len(sizes))], 0)
    check_lsap_status(status, defer=defer_status)
    res = [(ri[o:o + n], ci[o:o + n]) for o, n in zip(out_off, pairs)]
    return (res, status) if with_status else res      # status (int32 per problem, device): lets a deferred caller gate its writes on the device
