"""Oracle restatement of the distillation pieces (test infrastructure; see oracle/__init__.py).

Follows /root/reference/models/mdetr.py step by step in fp32 on the CPU: the char-span lookup with its fallbacks
(:112-141), ClusterCriterion.update_memory_queue / update_memory / memory_cluster / forward (:63-277), loss_softkd with
its matcher (:520-599), loss_nsthl2 (:668-781), and /root/reference/models/kmeans.py.  Assignments are solved by
oracle/lsap.c (pinned to SciPy) instead of scipy.optimize.  Pinned by tests/test_cpu_distill.py against
tests/golden/distill.npz, which the real reference produced (tests/golden/make_golden_distill.py).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import lsap
from .matcher_ref import cxcywh_to_xyxy, pairwise_giou


def span_tokens(tokenized, i, beg, end):
    """mdetr.py:124-139 (the retries drop the batch index, as the reference does)"""
    a, b = tokenized.char_to_token(i, beg), tokenized.char_to_token(i, end - 1)
    if a is None:
        try:
            a = tokenized.char_to_token(beg + 1)
            if a is None:
                a = tokenized.char_to_token(beg + 2)
        except Exception:
            a = None
    if b is None:
        try:
            b = tokenized.char_to_token(end - 2)
            if b is None:
                b = tokenized.char_to_token(end - 3)
        except Exception:
            b = None
    return None if a is None or b is None else (a, b)


def positions(tokenized, i, spans, length):
    hit = torch.zeros(length)
    for beg, end in spans:
        ab = span_tokens(tokenized, i, beg, end)
        if ab is not None:
            hit[ab[0]:ab[1] + 1] = 1
    return hit.nonzero().reshape(-1)


def noun_features(text, tokenized, targets):
    """mdetr.py:112-145 -> [B, d]"""
    B, L, d = text.shape
    out = torch.zeros(B, d)
    for i, tgt in enumerate(targets):
        rows = [text[i][positions(tokenized, i, spans, L)].mean(0) for spans in tgt["noun_tokens_positive"]]
        if rows:
            out[i] = torch.stack(rows).mean(0)
    return out


def kmeans(X, centers, num_clusters, tol=1e-4):
    """kmeans.py:21-94 with full_label != 0 (given initial centres); returns (assignment, centres)."""
    centers = centers.clone()
    while True:
        d = ((X.unsqueeze(1) - centers.unsqueeze(0)) ** 2.0).sum(-1)
        choice = d.argmin(1)
        prev = centers.clone()
        for c in range(num_clusters):
            sel = X[choice == c]
            if len(sel):
                centers[c] = sel.mean(0)
        if float(torch.sqrt(((centers - prev) ** 2).sum(1)).sum()) ** 2 < tol:
            return choice, centers


def replace_nearest(bank, new):
    """mdetr.py:98-103"""
    rows, cols = lsap.linear_sum_assignment(torch.cdist(new, bank, p=1).double().numpy())
    bank = bank.clone()
    for r, c in zip(rows, cols):
        bank[c] = new[r]
    return bank


def cluster_substitute(img_memory, text_len, i, pos, bank, centers, feature, num_clusters):
    """memory_cluster + prototype substitution for one sample (mdetr.py:207-228, 203): returns (img_memory_mod, centres, centre)."""
    _, centers = kmeans(bank, centers, num_clusters)
    pick = int(((feature.reshape(1, -1).unsqueeze(1) - centers.unsqueeze(0)) ** 2.0).sum(-1).argmin(1)[0])
    img_memory[-text_len:, i, :][pos] = centers[pick]
    return img_memory, centers, centers[pick]


def loss_nsthl2(text_noun, text_sth, tok_noun, tok_sth, targets_noun, targets_sth, matched_counts_sth):
    """mdetr.py:668-781"""
    fn, fs = noun_features(text_noun, tok_noun, targets_noun), noun_features(text_sth, tok_sth, targets_sth)
    keep = [i for i, c in enumerate(matched_counts_sth) if c > 0]
    if not keep:
        return torch.zeros(())
    return sum(F.mse_loss(fs[i], fn[i]) for i in keep) / len(keep)


def loss_softkd(logits_noun, logits_sth, boxes_noun, boxes_sth, idx_noun, idx_sth):
    """mdetr.py:520-599 for one layer; idx_* = per-image (src, tgt) index pairs."""
    def binar(lg):
        p = lg.softmax(-1)
        return torch.cat([p[..., :-1].sum(-1, keepdim=True), p[..., -1:]], -1)

    pn, ps = binar(logits_noun), binar(logits_sth)
    Q = pn.shape[1]
    total = torch.zeros(())
    for i in range(pn.shape[0]):
        sides = []
        for p, bx, (src, tgt) in ((pn, boxes_noun, idx_noun[i]), (ps, boxes_sth, idx_sth[i])):
            tp = torch.zeros(len(src), 2)
            tp[tgt] = p[i][src]
            free = torch.ones(Q, dtype=torch.bool)
            free[src] = False
            sides.append((tp, p[i][free], bx[i][free]))
        (tp_n, fp_n, fb_n), (tp_s, fp_s, fb_s) = sides
        cost = (fp_n * (fp_n.unsqueeze(0).log() - fp_s.log().unsqueeze(1))).sum(-1) + torch.cdist(fb_s, fb_n, p=1) \
            - pairwise_giou(cxcywh_to_xyxy(fb_s), cxcywh_to_xyxy(fb_n))
        rows, cols = lsap.linear_sum_assignment(cost.double().numpy())
        teacher = torch.cat([tp_n, fp_n[torch.as_tensor(cols)]])
        student = torch.cat([tp_s, fp_s[torch.as_tensor(rows)]])
        total = total + F.kl_div(student.log(), teacher, reduction="batchmean")
    return total / pn.shape[0]
