"""Noun -> pronoun distillation pieces of TOIST (BASELINE config 5).

Mirrors the reference's ClusterCriterion (/root/reference/models/mdetr.py:29-312), its k-means
(/root/reference/models/kmeans.py) and the char-span -> token-span lookup shared by the contrastive, nsthl2 and
cluster code (mdetr.py:112-141, 614-643, 684-711).  Everything stays on the device: the memory-bank
replacement assignment (mdetr.py:98-103) runs on the HIP LSAP kernel instead of SciPy on the host, k-means
is a handful of device tensor ops per iteration ([1024, 256] bank, 3 centres), and nothing calls `.cpu()`.
Buffer names and shapes equal the reference's, so a reference checkpoint of the criterion loads unchanged.
"""
import collections

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from .matcher import linear_sum_assignment_batch


# ---- char spans -> token positions --------------------------------------------------------------------------
def char_span_to_tokens(tokenized, batch_index, beg, end):
    """(first, last) token of the character span [beg, end) of caption `batch_index`, or None.  Same lookup
    and the same fallbacks as the reference (note that its retries drop the batch index, mdetr.py:126-139)."""
    first = tokenized.char_to_token(batch_index, beg)
    last = tokenized.char_to_token(batch_index, end - 1)
    if first is None:
        try:
            first = tokenized.char_to_token(beg + 1)
            if first is None:
                first = tokenized.char_to_token(beg + 2)
        except Exception:
            first = None
    if last is None:
        try:
            last = tokenized.char_to_token(end - 2)
            if last is None:
                last = tokenized.char_to_token(end - 3)
        except Exception:
            last = None
    if first is None or last is None:
        return None
    return first, last


def span_positions(tokenized, batch_index, spans, length):
    """int64 token positions (ascending, unique) covered by a list of character spans."""
    hit = torch.zeros(length, dtype=torch.bool)
    for beg, end in spans:
        ft = char_span_to_tokens(tokenized, batch_index, beg, end)
        if ft is not None:
            hit[ft[0]:ft[1] + 1] = True
    return hit.nonzero().reshape(-1)


# Caption-driven index tables of a batch (token weights, substitution masks, task columns) are built once on the host and kept on the device while the
# batch's objects are alive: the steps that use them issue no host -> device copy and read nothing back, so the launch queue never drains inside a
# training step (every small pageable upload is a stream synchronisation).  Keyed by object identity; the entry holds the objects, so an id cannot be reused.
_TABLES = collections.OrderedDict()


def _tok_ref(tokenized):
    """the object whose identity stands for a tokenized batch: its input_ids tensor when there is one (Transformer._tokenize wraps the caller's dict in a
    new TokenizedText on every call, the tensors inside stay the same objects), else the tokenized object itself"""
    try:
        ids = tokenized["input_ids"]
        if torch.is_tensor(ids):
            return ids
    except Exception:
        pass
    return tokenized


def _cached_table(kind, refs, extra, make):
    # identity + in-place version of tensor refs: a caller that refills a static input_ids buffer in place (a captured step's feed) bumps
    # Tensor._version, so a table built for the previous contents is not served again
    key = (kind, tuple((id(r), r._version) if torch.is_tensor(r) else id(r) for r in refs), extra)
    ent = _TABLES.get(key)
    if ent is not None:
        _TABLES.move_to_end(key)
        return ent[1]
    val = make()
    _TABLES[key] = (refs, val)
    while len(_TABLES) > 64:
        _TABLES.popitem(last=False)
    return val


def noun_token_weights(tokenized, targets, L, device):
    """fp32 [B, L]: w[i, l] = sum over the boxes of image i of [token l belongs to the box's noun spans] / (tokens of the box x boxes of the image), so that
    w @ text_feature[i] is the reference's mean over boxes of the mean over each box's noun tokens (mdetr.py:112-145, 684-711).  A box without any token
    makes the reference average an empty set: the row is NaN, as there."""
    def make():
        W = np.zeros((len(targets), L), dtype=np.float32)
        for i, tgt in enumerate(targets):
            boxes = tgt["noun_tokens_positive"]
            for spans in boxes:
                pos = span_positions(tokenized, i, spans, L).numpy()
                if len(pos) == 0:
                    W[i, :] = np.nan
                else:
                    W[i, pos] += np.float32(1.0 / (len(pos) * len(boxes)))
        return torch.from_numpy(W).to(device)
    return _cached_table("noun_w", (_tok_ref(tokenized), *targets), (L, str(device)), make)


def noun_token_features(text_feature, tokenized, targets):
    """[B, d]: per image, the mean over its boxes of the mean text feature of each box's noun tokens
    (mdetr.py:112-145, 684-711); zero rows for images without boxes.  One weighted sum over the tokens with host-built, device-cached weights."""
    B, L, d = text_feature.shape
    W = noun_token_weights(tokenized, targets, L, text_feature.device).to(text_feature.dtype)
    return (W.unsqueeze(-1) * text_feature).sum(1)


def task_index(dataset_name):
    """'task_3_train.json' -> 2 (mdetr.py:162)."""
    return int(dataset_name.split("_")[1]) - 1


# ---- the same tables at FIXED device addresses: one hipGraph of the distillation step serves any batch ------------------------------------
class DistillTables:
    """Fixed-address device image of ONE side's (noun or pronoun) caption-driven tables, the counterpart of matcher.StaticTargets for the distillation
    losses (round 6): the captured step reads every per-batch index from these buffers, `load()` refills them between replays
    (one pinned staging arena, one asynchronous H2D copy, no host sync).  `matcher.StaticTargets.distill` carries the side's tables into
    ClusterCriterion.update_memory / forward and SetCriterion.

    Arena (B = batch, L = tokens of the bucket):
      W_span  f32 [B, L]   noun_token_weights(): W @ text_memory[i] = mean over the image's boxes of the mean over each box's noun tokens (mdetr.py:112-145)
      W_sth   f32 [B, L]   mean weights of the tokens of the word 'something' in the caption (mdetr.py:240-260); zeros on the noun side
      sub_span u8 [L, B]   tokens that receive the prototype on the teacher side (images without boxes: none)
      sub_sth  u8 [L, B]   tokens of 'something'
      task    i32 [B]      task index of the image (dataset_name), -1 = the image takes no part (teacher side: no boxes)
      group_task i32 [B] | group_off i32 [B + 1] | members i32 [B]    the images grouped by task, in batch order inside a group (csrc/kmeans.hip, the
                           per-task LSAP of the memory-bank update); unused groups are empty (group_off stays at the member count)."""

    def __init__(self, batch, tokens, device, pronoun_side):
        self.B, self.L, self.pronoun_side = int(batch), int(tokens), bool(pronoun_side)
        self.device = torch.device(device)
        B, L = self.B, self.L
        sizes = [B * L * 4, B * L * 4, L * B, L * B, B * 4, B * 4, (B + 1) * 4, B * 4]
        offs, total = [], 0
        for n in sizes:
            offs.append(total)
            total += (n + 15) // 16 * 16
        self._pin = self.device.type == "cuda"                   # (a host-only instance packs tables in the CPU tests)
        self._host = torch.zeros(total, dtype=torch.uint8)
        if self._pin:
            self._host = self._host.pin_memory()
        self._dev = torch.zeros(total, dtype=torch.uint8, device=self.device)

        def views(buf):
            cut = lambda i, dt, shape: buf[offs[i]:offs[i] + sizes[i]].view(dt).view(shape)
            return (cut(0, torch.float32, (B, L)), cut(1, torch.float32, (B, L)), cut(2, torch.uint8, (L, B)), cut(3, torch.uint8, (L, B)), cut(4, torch.int32, (B,)),
                    cut(5, torch.int32, (B,)), cut(6, torch.int32, (B + 1,)), cut(7, torch.int32, (B,)))

        self._views = views
        self.W_span, self.W_sth, self.sub_span, self.sub_sth, self.task, self.group_task, self.group_off, self.members = views(self._dev)
        self._event = None

    def pack(self, tokenized, targets, captions=None, out=None):
        """Host image of one batch.  tokenized: the side's BatchEncoding (needs char_to_token); targets: the side's list of dicts with `noun_tokens_positive`,
        `dataset_name`, `boxes`; captions: list[str] (pronoun side: where the word 'something' sits)."""
        B, L = self.B, self.L
        if len(targets) != B:
            raise ValueError(f"DistillTables holds {B} images; got {len(targets)}")
        host = out if out is not None else torch.zeros(self._host.numel(), dtype=torch.uint8)
        if out is None and self._pin:
            host = host.pin_memory()
        host.zero_()
        W_span, W_sth, sub_span, sub_sth, task, group_task, group_off, members = (v.numpy() for v in self._views(host))
        empty = [len(t["boxes"]) == 0 for t in targets]
        for i, tgt in enumerate(targets):
            boxes = tgt["noun_tokens_positive"]
            for spans in boxes:
                pos = span_positions(tokenized, i, spans, L).numpy()
                if len(pos) == 0:
                    W_span[i, :] = np.nan                      # the reference averages an empty set there
                else:
                    W_span[i, pos] += np.float32(1.0 / (len(pos) * len(boxes)))
            if not empty[i]:
                sub_span[span_positions(tokenized, i, [s_ for box in boxes for s_ in box], L).numpy(), i] = 1
        if self.pronoun_side:
            for i in range(B):
                beg = captions[i].find("something")
                first, last = (tokenized.char_to_token(i, beg), tokenized.char_to_token(i, beg + len("something") - 1)) if beg >= 0 else (None, None)
                if first is None or last is None:      # (the reference fails here too: mdetr.py:240-246 indexes with the None it gets back)
                    raise ValueError(f"pronoun caption {i} ({captions[i]!r}) has no token for the word 'something'")
                pos = np.arange(first, last + 1)
                if len(pos) == 0:
                    W_sth[i, :] = np.nan
                else:
                    W_sth[i, pos] = np.float32(1.0 / len(pos))
                    sub_sth[pos, i] = 1
            tasks = [task_index(t["dataset_name"]) for t in targets]               # the student clusters EVERY sample (mdetr.py:230-277)
        else:
            tasks = [-1 if empty[i] else task_index(t["dataset_name"]) for i, t in enumerate(targets)]
        task[:] = np.asarray(tasks, dtype=np.int32)
        live = [(i, t) for i, t in enumerate(tasks) if t >= 0]
        order = sorted(set(t for _, t in live))
        mem, off = [], [0]
        for t in order:
            mem += [i for i, tt in live if tt == t]
            off.append(len(mem))
        group_task[:len(order)] = order
        group_off[:len(off)] = off
        group_off[len(off):] = len(mem)
        members[:len(mem)] = mem
        return host

    def load_packed(self, host):
        self._dev.copy_(host, non_blocking=True)
        return self

    def load(self, tokenized, targets, captions=None):
        if self._event is not None:
            self._event.synchronize()          # the previous upload has left the staging buffer
        self.load_packed(self.pack(tokenized, targets, captions, out=self._host))
        self._event = torch.cuda.Event()
        self._event.record()
        return self


# ---- k-means (kmeans.py) ---------------------------------------------------------------------------------------
def _sq_dist(x, centers):
    return ((x.unsqueeze(1) - centers.unsqueeze(0)) ** 2.0).sum(-1)


def kmeans(X, init_cluster_centers, num_clusters, tol=1e-4, full_label=0):
    """Lloyd iterations until (sum_k |shift_k|)^2 < tol (kmeans.py:21-94).  With full_label == 0 the start is
    `np.random.choice` rows of X (kmeans.py:8-18: parity with the reference is only defined once the bank is full)."""
    X = X.float()
    if full_label == 0:
        centers = X[torch.as_tensor(np.random.choice(len(X), num_clusters, replace=False), device=X.device)].clone()
    else:
        centers = init_cluster_centers
    while True:
        choice = torch.argmin(_sq_dist(X, centers), dim=1)
        before = centers.clone()
        for c in range(num_clusters):          # K = 3: the reference's own per-cluster mean (same reduction order)
            members = X[choice == c]
            if len(members) != 0:              # an empty cluster keeps its centre (kmeans.py:72)
                centers[c] = members.mean(dim=0)
        shift = torch.sqrt(((centers - before) ** 2).sum(1)).sum()
        if float(shift) ** 2 < tol:
            return choice, centers


def kmeans_predict(X, centers):
    return torch.argmin(_sq_dist(X.float(), centers), dim=1)


# ---- ClusterCriterion ------------------------------------------------------------------------------------------
class ClusterCriterion(nn.Module):
    """Per-task memory bank of noun text features + k-means prototypes (mdetr.py:29-312)."""

    def __init__(self, feature_dim, memory_size, cluster_num, task_count, args):
        super().__init__()
        self.args = args
        self.feature_dim, self.memory_size, self.cluster_num, self.task_count = feature_dim, memory_size, cluster_num, task_count
        self.register_buffer("feature_bank", torch.randn(task_count, memory_size, feature_dim))
        self.register_buffer("cluster_centers", torch.randn(task_count, cluster_num, feature_dim))
        self.register_buffer("update_count", torch.zeros(task_count))
        self.register_buffer("full_label", torch.zeros(task_count))

    @staticmethod
    def _world():
        return torch.distributed.get_world_size() if torch.distributed.is_available() and torch.distributed.is_initialized() else 1

    def syn_memory(self):
        """Average the banks and centres over the ranks (mdetr.py:54-61)."""
        world = self._world()
        if world > 1:
            torch.distributed.all_reduce(self.feature_bank)
            torch.distributed.all_reduce(self.cluster_centers)
        self.feature_bank /= world
        self.cluster_centers /= world

    _MEMBERS = {}

    @staticmethod
    def _members(tasks, t, device):
        """int64 device indices of the rows whose task is t, cached per task signature (no upload in steady state)"""
        key = (tuple(tasks), t, str(device))
        ent = ClusterCriterion._MEMBERS.get(key)
        if ent is None:
            if len(ClusterCriterion._MEMBERS) > 512:
                ClusterCriterion._MEMBERS.clear()
            ent = ClusterCriterion._MEMBERS[key] = torch.tensor([i for i, x in enumerate(tasks) if x == t], dtype=torch.int64, device=device)
        return ent

    def update_memory_queue(self, feature_idx, tasks_host=None):
        """feature_idx [B, d+1]: feature | task index (-1 = none).  All ranks' rows enter every rank's bank in rank
        order (mdetr.py:63-103): FIFO until a task's bank has been filled, then FIFO or nearest-replacement.
        tasks_host: the task column as a host list when the caller knows it (one rank: the targets' dataset names) -- nothing is read back then."""
        world = self._world()
        if world > 1:
            parts = [torch.zeros_like(feature_idx) for _ in range(world)]
            torch.distributed.all_gather(parts, feature_idx)
            feature_idx = torch.cat(parts, 0)
            tasks_host = None
        if tasks_host is not None:
            tasks = [int(t) for t in tasks_host]
        else:
            tasks = feature_idx[:, -1].round().to(torch.int64).tolist()   # one host read per step; the values are small integers
        full = self._full_host()
        count = self.__dict__["_count_mirror"]
        for t in sorted(set(tasks) - {-1}):
            new = feature_idx.index_select(0, self._members(tasks, t, feature_idx.device))[:, :-1]
            n = new.shape[0]
            bank = self.feature_bank[t]
            filling = not full[t]
            if filling or self.args.fifo_memory:
                bank.copy_(torch.cat([bank[n:], new], 0))
                if filling:
                    if count[t] > self.memory_size:
                        self.full_label[t] = 1
                        full[t] = True
                    self.update_count[t] += n
                    count[t] += n
            else:   # replace the entries closest (L1) to the new features: one LSAP on the device
                # SciPy raises on an invalid matrix (mdetr.py:100) BEFORE the bank is touched.  The status check here is deferred (no host
                # sync per step), so the write itself is gated on the device: an invalid / infeasible problem leaves the bank as it was
                # and the ValueError surfaces at the next deferred check (or state_dict() / check_lsap_pending()).
                ((rows, cols),), status = linear_sum_assignment_batch([torch.cdist(new, bank, p=1)], defer_status=True, with_status=True)
                bank[cols] = torch.where(status[0] == 0, new[rows], bank[cols])

    # ---- device path: every sample of the batch in ONE launch, no host read (csrc/kmeans.hip) -------------------------------
    _INDEX_TABLES = {}

    def cluster_batch(self, features, tasks):
        """memory_cluster for all samples of a batch at once.  features [B, d] (device), tasks: host list, one task index per
        sample or None (sample skipped).  Samples of the same task are processed in batch order by one workgroup (a later sample starts
        from the centres an earlier one left, as in the sequential reference); the centres are updated in place.  Returns
        (pick int32 [B], chosen centre [B, d]) on the device, or None when a bank involved is still filling (random initialisation on
        the host, kmeans.py:8-18: the per-sample host path handles that phase)."""
        live = [(i, t) for i, t in enumerate(tasks) if t is not None]
        if not live or not features.is_cuda:
            return None
        full = self._full_host()
        if any(not full[t] for _, t in live):
            return None
        key = (tuple(tasks), str(features.device))
        ent = ClusterCriterion._INDEX_TABLES.get(key)
        if ent is None:
            if len(ClusterCriterion._INDEX_TABLES) > 256:
                ClusterCriterion._INDEX_TABLES.clear()
            order = sorted(set(t for _, t in live))
            members, off = [], [0]
            for t in order:
                members += [i for i, tt in live if tt == t]
                off.append(len(members))
            mk = lambda v: torch.tensor(v, dtype=torch.int32, device=features.device)
            ent = ClusterCriterion._INDEX_TABLES[key] = (mk(order), mk(off), mk(members))
        group_task, group_off, members = ent
        B, d = features.shape
        pick = torch.zeros(B, dtype=torch.int32, device=features.device)
        chosen = torch.zeros(B, d, dtype=torch.float32, device=features.device)
        from . import kernels as k
        k.kmeans(self.feature_bank, self.cluster_centers, group_task, group_off, members, features.detach().float().contiguous(), 1e-4, 10000, pick, chosen)
        return pick, chosen

    def _full_host(self):
        """Host mirror of full_label (one read, then kept in step by update_memory_queue; call sync_host_state() after writing the
        buffers directly or loading a checkpoint)."""
        m = self.__dict__.get("_full_mirror")
        if m is None:
            m = self.__dict__["_full_mirror"] = [bool(v) for v in self.full_label.detach().cpu().tolist()]
            self.__dict__["_count_mirror"] = [float(v) for v in self.update_count.detach().cpu().tolist()]
        return m

    def state_dict(self, *a, **kw):
        from .matcher import check_lsap_pending
        check_lsap_pending()        # a deferred LSAP error of the bank update (invalid cost matrix) must not be saved silently
        return super().state_dict(*a, **kw)

    def sync_host_state(self):
        self.__dict__.pop("_full_mirror", None)
        self.__dict__.pop("_count_mirror", None)

    def _load_from_state_dict(self, *a, **kw):
        self.sync_host_state()
        return super()._load_from_state_dict(*a, **kw)

    def memory_cluster(self, feature, t):
        choice, centers = kmeans(self.feature_bank[t], self.cluster_centers[t].clone(), self.cluster_num, full_label=float(self.full_label[t]))
        self.cluster_centers[t] = centers
        pick = kmeans_predict(feature.reshape(1, -1), centers)[0]
        return pick, centers[pick]

    def _substitute(self, memory_cache, i, pos, t, feature):
        """Overwrite the text-memory rows `pos` of sample i in img_memory_mod with the chosen prototype."""
        pick, center = self.memory_cluster(feature.clone().detach(), t)
        L = len(memory_cache["text_memory"])
        memory_cache["img_memory_mod"][-L:, i, :][pos.to(center.device)] = self.cluster_centers[t, pick]
        return center

    # ---- hipGraph replay for ANY batch: every per-batch index comes from a DistillTables image (fixed addresses) ----------------------------
    _CONST = {}

    def _static_ok(self):
        if self.args.fifo_memory or not all(self._full_host()) or self._world() > 1:
            raise RuntimeError("ClusterCriterion on StaticTargets (a captured distillation step) needs the steady state of a single process: every task's "
                               "memory bank full, nearest-replacement updates (fifo_memory off); run the list-of-dicts path until then")

    def _replace_nearest_static(self, rows, tb):
        """update_memory_queue's nearest-replacement branch (mdetr.py:98-103) with the grouping of the batch by task read from the device: one LSAP problem
        per group slot (rows = the group's size, 0 for an unused slot), pair table of fixed capacity B per problem, and a bank write that skips the dead
        slots and the problems whose status is not OK (toist_scatter_rows_f32).  rows [B, d] f32 (detached features)."""
        from . import kernels as k
        from .matcher import check_lsap_status
        B, d = rows.shape
        N, dev = self.memory_size, rows.device
        key = ("bank", B, N, str(dev))
        const = ClusterCriterion._CONST.get(key)
        if const is None:
            slot = torch.arange(B * B, dtype=torch.int64, device=dev)
            const = ClusterCriterion._CONST[key] = (torch.full((B,), N, dtype=torch.int32, device=dev), torch.arange(B, dtype=torch.int64, device=dev) * B, slot // B, slot % B)
        cols, out_off, g_of, s_of = const
        mem = tb.members.to(torch.int64)
        off = tb.group_off.to(torch.int64)
        rs = rows.index_select(0, mem).contiguous()                                  # rows in group order
        trow = tb.task.to(torch.int64).clamp(min=0).index_select(0, mem)
        cost = torch.cdist(rs[:, None, :], self.feature_bank.index_select(0, trow), p=1).reshape(B, N).contiguous()
        n_g = (off[1:] - off[:-1]).to(torch.int32)
        ri = torch.zeros(B * B, dtype=torch.int64, device=dev)
        ci = torch.zeros(B * B, dtype=torch.int64, device=dev)
        status = torch.zeros(B, dtype=torch.int32, device=dev)
        k.lsap(cost, (off[:-1] * N).contiguous(), n_g, cols, B, B, N, B * N, out_off, ri, ci, status, ld=N)
        valid = (s_of < n_g.to(torch.int64)[g_of]) & (status[g_of] == 0)
        src_row = torch.where(valid, off[g_of] + ri, torch.full_like(ri, -1))
        dst_row = torch.where(valid, tb.group_task.to(torch.int64)[g_of] * N + ci, torch.full_like(ci, -1))
        k.scatter_rows(rs, src_row, self.feature_bank.view(-1, d), dst_row)
        check_lsap_status(status, defer=True)

    def _cluster_static(self, features, tb):
        from . import kernels as k
        B, d = features.shape
        pick = torch.zeros(B, dtype=torch.int32, device=features.device)
        chosen = torch.zeros(B, d, dtype=torch.float32, device=features.device)
        k.kmeans(self.feature_bank, self.cluster_centers, tb.group_task, tb.group_off, tb.members, features.detach().float().contiguous(), 1e-4, 10000, pick, chosen)
        return pick, chosen

    def update_memory_static(self, memory_cache_noun, tb):
        """update_memory on a DistillTables image (teacher side)."""
        self._static_ok()
        text = memory_cache_noun["text_memory"].permute(1, 0, 2)
        B, L, d = text.shape
        feats = (tb.W_span.to(text.dtype).unsqueeze(-1) * text).sum(1)
        self._replace_nearest_static(feats.detach().float(), tb)
        memory_cache_noun["img_memory_mod"] = memory_cache_noun["img_memory"].clone()
        _, chosen = self._cluster_static(feats, tb)
        mod = memory_cache_noun["img_memory_mod"]
        mod[-L:] = torch.where(tb.sub_span.bool()[:, :, None], chosen[None].to(mod.dtype), mod[-L:])
        memory_cache_noun["full_label"], memory_cache_noun["update_count"] = self.full_label, self.update_count
        return memory_cache_noun

    def forward_static(self, memory_cache_sth, tb):
        """forward (student side) on a DistillTables image: ('something' -> prototype, loss_cluster_feature)."""
        self._static_ok()
        text = memory_cache_sth["text_memory"].permute(1, 0, 2)
        B, L, _ = text.shape
        memory_cache_sth["img_memory_mod"] = memory_cache_sth["img_memory"].clone()
        features = (tb.W_sth.to(text.dtype).unsqueeze(-1) * text).sum(1)
        _, chosen = self._cluster_static(features, tb)
        mod = memory_cache_sth["img_memory_mod"]
        mod[-L:] = torch.where(tb.sub_sth.bool()[:, :, None], chosen[None].to(mod.dtype), mod[-L:])
        loss = ((features - chosen.to(features.dtype)) ** 2).mean(1).sum() / max(B, 1)
        return memory_cache_sth, {"loss_cluster_choice": torch.zeros((), device=loss.device), "loss_cluster_feature": loss}

    def update_memory(self, memory_cache_noun, targets_noun, captions_noun):
        """Teacher side (mdetr.py:105-205): push this batch's noun features into the banks, then replace the noun
        tokens of the teacher's text memory by their prototype."""
        tb = getattr(targets_noun, "distill", None)
        if tb is not None:                      # matcher.StaticTargets of a captured step
            return self.update_memory_static(memory_cache_noun, tb)
        text = memory_cache_noun["text_memory"].permute(1, 0, 2)
        B, L, d = text.shape
        dev = text.device
        tokenized = memory_cache_noun["tokenized"]
        feats = noun_token_features(text, tokenized, targets_noun)
        empty = [len(t["boxes"]) == 0 for t in targets_noun]
        tasks = [None if empty[i] else task_index(t["dataset_name"]) for i, t in enumerate(targets_noun)]

        def make():     # task column of the queue rows and the [L, B] mask of the noun tokens that receive the prototype
            col = torch.tensor([-1.0 if t is None else float(t) for t in tasks], dtype=torch.float32)
            mask = torch.zeros(L, B, dtype=torch.bool)
            for i, tgt in enumerate(targets_noun):
                if not empty[i]:
                    mask[span_positions(tokenized, i, [s for box in tgt["noun_tokens_positive"] for s in box], L), i] = True
            return col.to(dev), mask.to(dev)
        task_col, sub_mask = _cached_table("noun_meta", (_tok_ref(tokenized), *targets_noun), (L, str(dev)), make)
        rows = torch.cat([feats.detach().float(), task_col[:, None]], dim=1)          # rows of images without boxes: zero feature, task -1
        self.update_memory_queue(rows, tasks_host=[-1 if t is None else t for t in tasks])
        memory_cache_noun["img_memory_mod"] = memory_cache_noun["img_memory"].clone()
        batch = self.cluster_batch(feats, tasks)
        if batch is not None:
            mod = memory_cache_noun["img_memory_mod"]
            mod[-L:] = torch.where(sub_mask[:, :, None], batch[1][None].to(mod.dtype), mod[-L:])
        else:
            for i, tgt in enumerate(targets_noun):
                if empty[i]:
                    continue
                spans = [s for box in tgt["noun_tokens_positive"] for s in box]
                self._substitute(memory_cache_noun, i, span_positions(tokenized, i, spans, L), tasks[i], feats[i])
        memory_cache_noun["full_label"], memory_cache_noun["update_count"] = self.full_label, self.update_count
        return memory_cache_noun

    def _something(self, memory_cache, names, captions, with_loss):
        text = memory_cache["text_memory"].permute(1, 0, 2)
        B, L, _ = text.shape
        dev = text.device
        tokenized = memory_cache["tokenized"]
        memory_cache["img_memory_mod"] = memory_cache["img_memory"].clone()

        def make():     # tokens of the pronoun "something" per caption: mean weights [B, L], substitution mask [L, B], host positions
            W = torch.zeros(B, L, dtype=torch.float32)
            mask = torch.zeros(L, B, dtype=torch.bool)
            positions = []
            for i in range(B):
                beg = captions[i].find("something")
                pos = torch.arange(tokenized.char_to_token(i, beg), tokenized.char_to_token(i, beg + len("something") - 1) + 1)
                positions.append(pos)
                W[i, pos] = 1.0 / max(len(pos), 1) if len(pos) else float("nan")
                mask[pos, i] = True
            if any(len(p_) == 0 for p_ in positions):       # the reference averages an empty set there: NaN
                for i, p_ in enumerate(positions):
                    if len(p_) == 0:
                        W[i, :] = float("nan")
            return W.to(dev), mask.to(dev), positions
        W, sub_mask, positions = _cached_table("sth_meta", (_tok_ref(tokenized),), (tuple(captions), L, str(dev)), make)
        features = (W.to(text.dtype).unsqueeze(-1) * text).sum(1)                      # [B, d]: text[i][pos].mean(0)
        tasks = [task_index(n) for n in names]
        batch = self.cluster_batch(features, tasks) if B else None
        if batch is not None:
            mod = memory_cache["img_memory_mod"]
            mod[-L:] = torch.where(sub_mask[:, :, None], batch[1][None].to(mod.dtype), mod[-L:])
            loss_feature = ((features - batch[1].to(features.dtype)) ** 2).mean(1).sum() if with_loss else torch.zeros((), device=dev)
        else:
            loss_feature = torch.zeros((), device=dev)
            for i in range(B):
                center = self._substitute(memory_cache, i, positions[i], tasks[i], features[i])
                if with_loss:
                    loss_feature = loss_feature + F.mse_loss(features[i], center)
        return memory_cache, loss_feature / max(B, 1)

    def forward(self, memory_cache_sth, targets_sth, captions_sth):
        """Student side (mdetr.py:230-277): the pronoun 'something' is replaced by the prototype closest to its own
        feature; loss_cluster_feature pulls that feature towards the prototype (loss_cluster_choice stays 0)."""
        tb = getattr(targets_sth, "distill", None)
        if tb is not None:
            return self.forward_static(memory_cache_sth, tb)
        mc, loss = self._something(memory_cache_sth, [t["dataset_name"] for t in targets_sth], captions_sth, True)
        return mc, {"loss_cluster_choice": torch.zeros((), device=loss.device), "loss_cluster_feature": loss}

    @torch.no_grad()
    def infer_choice(self, memory_cache_sth, dataset_name_list, captions):
        """mdetr.py:279-312"""
        return self._something(memory_cache_sth, dataset_name_list, captions, False)[0]
