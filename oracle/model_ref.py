"""Oracle restatement of the TOIST/MDETR detection path (test infrastructure; see oracle/__init__.py).

Pure functions over a reference-compatible ``state_dict`` (same key names / shapes as the reference
modules), fp32, CPU, eval semantics (dropout off).  Each function cites the reference lines it
follows.  Sequence tensors are [S, B, d] (sequence first) exactly like the reference API.
"""
import math

import torch
import torch.nn.functional as F

from . import matcher_ref

# ------------------------------------------------------------------------------------------ backbone


def frozen_bn(x, sd, p):
    """FrozenBatchNorm2d.forward, /root/reference/models/backbone.py:48-58 (eps = 1e-5)."""
    scale = sd[p + "weight"] * (sd[p + "running_var"] + 1e-5).rsqrt()
    shift = sd[p + "bias"] - sd[p + "running_mean"] * scale
    return x * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)


RESNET101_BLOCKS = (3, 4, 23, 3)


def bottleneck(x, sd, p, stride, has_down):
    """torchvision Bottleneck (v1.5: the stride sits on the 3x3), as instantiated by
    /root/reference/models/backbone.py:87-89 with norm_layer=FrozenBatchNorm2d."""
    y = F.relu(frozen_bn(F.conv2d(x, sd[p + "conv1.weight"]), sd, p + "bn1."))
    y = F.relu(frozen_bn(F.conv2d(y, sd[p + "conv2.weight"], stride=stride, padding=1), sd, p + "bn2."))
    y = frozen_bn(F.conv2d(y, sd[p + "conv3.weight"]), sd, p + "bn3.")
    if has_down:
        x = frozen_bn(F.conv2d(x, sd[p + "downsample.0.weight"], stride=stride), sd, p + "downsample.1.")
    return F.relu(y + x)


def resnet_body(x, sd, p, blocks=RESNET101_BLOCKS):
    """ResNet stem + layer1..4 (torchvision.models.resnet101 body used at backbone.py:71-75).
    Returns [C2, C3, C4, C5]."""
    y = F.relu(frozen_bn(F.conv2d(x, sd[p + "conv1.weight"], stride=2, padding=3), sd, p + "bn1."))
    y = F.max_pool2d(y, kernel_size=3, stride=2, padding=1)
    feats = []
    for li, nb in enumerate(blocks, start=1):
        for bi in range(nb):
            stride = 2 if (bi == 0 and li > 1) else 1
            y = bottleneck(y, sd, f"{p}layer{li}.{bi}.", stride, has_down=(bi == 0))
        feats.append(y)
    return feats


def downsample_mask(mask, hw):
    """backbone.py:78 -- nearest interpolation of the bool padding mask."""
    return F.interpolate(mask[None].float(), size=hw).bool()[0]


def sine_position(mask, num_pos_feats=128, temperature=10000.0):
    """PositionEmbeddingSine.forward, /root/reference/models/position_encoding.py:30-49
    (normalize=True, scale=2*pi as built at :89-93)."""
    keep = ~mask
    y_embed = keep.cumsum(1, dtype=torch.float32)
    x_embed = keep.cumsum(2, dtype=torch.float32)
    eps, scale = 1e-6, 2 * math.pi
    y_embed = y_embed / (y_embed[:, -1:, :] + eps) * scale
    x_embed = x_embed / (x_embed[:, :, -1:] + eps) * scale
    dim_t = torch.arange(num_pos_feats, dtype=torch.float32)
    dim_t = temperature ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / num_pos_feats)
    px = x_embed[:, :, :, None] / dim_t
    py = y_embed[:, :, :, None] / dim_t
    px = torch.stack((px[..., 0::2].sin(), px[..., 1::2].cos()), dim=4).flatten(3)
    py = torch.stack((py[..., 0::2].sin(), py[..., 1::2].cos()), dim=4).flatten(3)
    return torch.cat((py, px), dim=3).permute(0, 3, 1, 2)


# ------------------------------------------------------------------------------------------ attention / layers


def layer_norm(x, sd, p, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), sd[p + "weight"], sd[p + "bias"], eps)


def mha(sd, p, query, key, value, key_padding_mask, nhead):
    """torch.nn.MultiheadAttention forward (eval) as used at transformer.py:273,337-338:
    packed in_proj, q scaled by dh^-0.5, -inf on padded keys, softmax, out_proj.  [S,B,d] in/out."""
    Sq, B, d = query.shape
    Sk = key.shape[0]
    dh = d // nhead
    W, b = sd[p + "in_proj_weight"], sd[p + "in_proj_bias"]
    q = F.linear(query, W[:d], b[:d]) * (dh ** -0.5)
    k = F.linear(key, W[d:2 * d], b[d:2 * d])
    v = F.linear(value, W[2 * d:], b[2 * d:])
    q = q.reshape(Sq, B * nhead, dh).transpose(0, 1)
    k = k.reshape(Sk, B * nhead, dh).transpose(0, 1)
    v = v.reshape(Sk, B * nhead, dh).transpose(0, 1)
    att = torch.bmm(q, k.transpose(1, 2))
    if key_padding_mask is not None:
        att = att.view(B, nhead, Sq, Sk).masked_fill(key_padding_mask[:, None, None, :], float("-inf")).view(B * nhead, Sq, Sk)
    att = torch.softmax(att, dim=-1)
    out = torch.bmm(att, v).transpose(0, 1).reshape(Sq, B, d)
    return F.linear(out, sd[p + "out_proj.weight"], sd[p + "out_proj.bias"])


def encoder_layer(sd, p, src, pos, key_padding_mask, nhead):
    """TransformerEncoderLayer.forward_post, transformer.py:290-304."""
    qk = src + pos
    src = layer_norm(src + mha(sd, p + "self_attn.", qk, qk, src, key_padding_mask, nhead), sd, p + "norm1.")
    ff = F.linear(F.relu(F.linear(src, sd[p + "linear1.weight"], sd[p + "linear1.bias"])), sd[p + "linear2.weight"], sd[p + "linear2.bias"])
    return layer_norm(src + ff, sd, p + "norm2.")


def decoder_layer(sd, p, tgt, memory, pos, query_pos, memory_key_padding_mask, nhead):
    """TransformerDecoderLayer.forward_post, transformer.py:362-408 (no text cross-attention)."""
    qk = tgt + query_pos
    tgt = layer_norm(tgt + mha(sd, p + "self_attn.", qk, qk, tgt, None, nhead), sd, p + "norm1.")
    tgt = layer_norm(
        tgt + mha(sd, p + "cross_attn_image.", tgt + query_pos, memory + pos, memory, memory_key_padding_mask, nhead), sd, p + "norm3.")
    ff = F.linear(F.relu(F.linear(tgt, sd[p + "linear1.weight"], sd[p + "linear1.bias"])), sd[p + "linear2.weight"], sd[p + "linear2.bias"])
    return layer_norm(tgt + ff, sd, p + "norm4.")


# ------------------------------------------------------------------------------------------ RoBERTa


def roberta(sd, p, input_ids, attention_mask, nhead, eps, pad_id=1):
    """HF RobertaModel forward (eval, no pooler), the call at transformer.py:130.  Returns
    last_hidden_state [B, L, hidden].  eps = config.layer_norm_eps."""
    keep = input_ids.ne(pad_id).int()
    pos_ids = (torch.cumsum(keep, dim=1) * keep).long() + pad_id
    x = sd[p + "embeddings.word_embeddings.weight"][input_ids]
    x = x + sd[p + "embeddings.token_type_embeddings.weight"][torch.zeros_like(input_ids)]
    x = x + sd[p + "embeddings.position_embeddings.weight"][pos_ids]
    x = F.layer_norm(x, (x.shape[-1],), sd[p + "embeddings.LayerNorm.weight"], sd[p + "embeddings.LayerNorm.bias"], eps)
    B, L, d = x.shape
    dh = d // nhead
    ext = (1.0 - attention_mask[:, None, None, :].float()) * torch.finfo(torch.float32).min
    i = 0
    while f"{p}encoder.layer.{i}.attention.self.query.weight" in sd:
        lp = f"{p}encoder.layer.{i}."
        heads = lambda t: t.view(B, L, nhead, dh).permute(0, 2, 1, 3)
        q = heads(F.linear(x, sd[lp + "attention.self.query.weight"], sd[lp + "attention.self.query.bias"]))
        k = heads(F.linear(x, sd[lp + "attention.self.key.weight"], sd[lp + "attention.self.key.bias"]))
        v = heads(F.linear(x, sd[lp + "attention.self.value.weight"], sd[lp + "attention.self.value.bias"]))
        att = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(dh) + ext, dim=-1)
        ctx = (att @ v).permute(0, 2, 1, 3).reshape(B, L, d)
        a = F.linear(ctx, sd[lp + "attention.output.dense.weight"], sd[lp + "attention.output.dense.bias"])
        x = F.layer_norm(a + x, (d,), sd[lp + "attention.output.LayerNorm.weight"], sd[lp + "attention.output.LayerNorm.bias"], eps)
        h = F.gelu(F.linear(x, sd[lp + "intermediate.dense.weight"], sd[lp + "intermediate.dense.bias"]))
        o = F.linear(h, sd[lp + "output.dense.weight"], sd[lp + "output.dense.bias"])
        x = F.layer_norm(o + x, (d,), sd[lp + "output.LayerNorm.weight"], sd[lp + "output.LayerNorm.bias"], eps)
        i += 1
    return x


# ------------------------------------------------------------------------------------------ MDETR


def count_layers(sd, prefix):
    n = 0
    while f"{prefix}{n}.linear1.weight" in sd:
        n += 1
    return n


def mdetr_encode(sd, images, pixel_mask, input_ids, attention_mask, *, nhead=8, text_nhead=12, text_eps=1e-12, prefix="",
                 features=None):
    """MDETR.forward(encode_and_save=True), mdetr.py:377-394 + Transformer.forward encode branch,
    transformer.py:98-168.  `features` (C5 [B,2048,h,w]) may be supplied to skip the backbone."""
    P = prefix
    if features is None:
        features = resnet_body(images, sd, P + "backbone.0.body.")[-1]
    mask = downsample_mask(pixel_mask, features.shape[-2:])
    d = sd[P + "input_proj.weight"].shape[0]
    pos = sine_position(mask, d // 2).to(features.dtype)
    src = F.conv2d(features, sd[P + "input_proj.weight"], sd[P + "input_proj.bias"])
    B = src.shape[0]
    src = src.flatten(2).permute(2, 0, 1)
    pos_embed = pos.flatten(2).permute(2, 0, 1)
    query_embed = sd[P + "query_embed.weight"].unsqueeze(1).repeat(1, B, 1)
    mask = mask.flatten(1)

    hidden = roberta(sd, P + "transformer.text_encoder.", input_ids, attention_mask, text_nhead, text_eps)
    text_memory = hidden.transpose(0, 1)
    text_attention_mask = attention_mask.ne(1).bool()
    # FeatureResizer, transformer.py:473-492 (LayerNorm eps 1e-12)
    text_memory_resized = F.layer_norm(
        F.linear(text_memory, sd[P + "transformer.resizer.fc.weight"], sd[P + "transformer.resizer.fc.bias"]), (d,),
        sd[P + "transformer.resizer.layer_norm.weight"], sd[P + "transformer.resizer.layer_norm.bias"], 1e-12)

    src = torch.cat([src, text_memory_resized], dim=0)
    mask = torch.cat([mask, text_attention_mask], dim=1)
    pos_embed = torch.cat([pos_embed, torch.zeros_like(text_memory_resized)], dim=0)
    out = src
    for i in range(count_layers(sd, P + "transformer.encoder.layers.")):
        out = encoder_layer(sd, f"{P}transformer.encoder.layers.{i}.", out, pos_embed, mask, nhead)
    L = text_memory_resized.shape[0]
    return {
        "text_memory_resized": text_memory_resized, "text_memory": out[-L:], "img_memory": out, "mask": mask,
        "text_attention_mask": text_attention_mask, "pos_embed": pos_embed, "query_embed": query_embed,
        "text_pooled_op": None, "img_pooled_op": None,
    }


def mlp(sd, p, x, n_layers=3):
    """MLP, mdetr.py:1024-1036."""
    for i in range(n_layers):
        x = F.linear(x, sd[f"{p}layers.{i}.weight"], sd[f"{p}layers.{i}.bias"])
        if i < n_layers - 1:
            x = F.relu(x)
    return x


def mdetr_decode(sd, mc, *, nhead=8, prefix="", aux_loss=True, contrastive_align=False):
    """MDETR.forward(encode_and_save=False), mdetr.py:396-462 + decoder, transformer.py:225-267."""
    P = prefix
    tgt = torch.zeros_like(mc["query_embed"])
    inter = []
    out = tgt
    for i in range(count_layers(sd, P + "transformer.decoder.layers.")):
        out = decoder_layer(sd, f"{P}transformer.decoder.layers.{i}.", out, mc["img_memory"], mc["pos_embed"], mc["query_embed"],
                            mc["mask"], nhead)
        inter.append(layer_norm(out, sd, P + "transformer.decoder.norm."))
    hs = torch.stack(inter).transpose(1, 2)  # [L, B, Q, d]
    logits = F.linear(hs, sd[P + "class_embed.weight"], sd[P + "class_embed.bias"])
    boxes = mlp(sd, P + "bbox_embed.", hs).sigmoid()
    res = {"pred_logits": logits[-1], "pred_boxes": boxes[-1], "hs": hs}
    pq = pt = None
    if contrastive_align:
        pq = F.normalize(F.linear(hs, sd[P + "contrastive_align_projection_image.weight"], sd[P + "contrastive_align_projection_image.bias"]),
                         p=2, dim=-1)
        pt = F.normalize(
            F.linear(mc["text_memory"], sd[P + "contrastive_align_projection_text.weight"], sd[P + "contrastive_align_projection_text.bias"])
            .transpose(0, 1), p=2, dim=-1)
        res.update({"proj_queries": pq[-1], "proj_tokens": pt})
    if aux_loss:
        res["aux_outputs"] = []
        for i in range(hs.shape[0] - 1):
            a = {"pred_logits": logits[i], "pred_boxes": boxes[i]}
            if contrastive_align:
                a.update({"proj_queries": pq[i], "proj_tokens": pt})
            res["aux_outputs"].append(a)
    return res


# ------------------------------------------------------------------------------------------ criterion


def _flat_indices(indices):
    """SetCriterion._get_src_permutation_idx, mdetr.py:855-859."""
    batch = torch.cat([torch.full_like(s, i) for i, (s, _) in enumerate(indices)])
    src = torch.cat([s for s, _ in indices])
    return batch, src


def loss_labels(out, targets, positive_map, indices, num_boxes, eos_coef=0.1):
    """mdetr.py:488-518."""
    logp = out["pred_logits"].log_softmax(-1)
    batch, src = _flat_indices(indices)
    offs, acc = [], 0
    for i, (_, tj) in enumerate(indices):
        offs.append(tj + acc)
        acc += len(targets[i]["boxes"])
    tgt_rows = positive_map[torch.cat(offs)]
    sim = torch.zeros_like(logp)
    sim[:, :, -1] = 1
    sim[batch, src] = tgt_rows
    ce = -(logp * sim).sum(-1)
    wgt = torch.full(ce.shape, eos_coef)
    wgt[batch, src] = 1
    return {"loss_ce": (ce * wgt).sum() / num_boxes}


def loss_boxes(out, targets, indices, num_boxes):
    """mdetr.py:805-825."""
    batch, src = _flat_indices(indices)
    pred = out["pred_boxes"][batch, src]
    tgt = torch.cat([t["boxes"][j] for t, (_, j) in zip(targets, indices)], dim=0)
    l1 = (pred - tgt).abs().sum() / num_boxes
    giou = torch.diag(matcher_ref.pairwise_giou(matcher_ref.cxcywh_to_xyxy(pred), matcher_ref.cxcywh_to_xyxy(tgt)))
    return {"loss_bbox": l1, "loss_giou": (1 - giou).sum() / num_boxes}


def loss_cardinality(out, targets):
    """mdetr.py:783-803."""
    logits = out["pred_logits"]
    lengths = torch.as_tensor([len(t["boxes"]) for t in targets], dtype=torch.float32)
    pred = (logits.argmax(-1) != logits.shape[-1] - 1).sum(1).float()
    return {"cardinality_error": (pred - lengths).abs().mean()}


def loss_contrastive_align(out, token_spans, indices, num_boxes, temperature=0.07):
    """mdetr.py:601-666 with the char_to_token lookup replaced by explicit token spans:
    token_spans[b][t] = list of (first_token, last_token) inclusive for target t of image b."""
    logits = out["proj_queries"] @ out["proj_tokens"].transpose(-1, -2) / temperature
    pm = torch.zeros(logits.shape, dtype=torch.bool)
    for b, (si, tj) in enumerate(indices):
        for q, t in zip(si.tolist(), tj.tolist()):
            for beg, end in token_spans[b][t]:
                pm[b, q, beg:end + 1] = True
    pos_logits = -logits.masked_fill(~pm, 0)
    b2t = ((pos_logits.sum(2) / (pm.sum(2) + 1e-6) + logits.logsumexp(2))).masked_fill(~pm.any(2), 0).sum()
    t2b = ((pos_logits.sum(1) / (pm.sum(1) + 1e-6) + logits.logsumexp(1))).masked_fill(~pm.any(1), 0).sum()
    return {"loss_contrastive_align": (b2t + t2b) / 2 / num_boxes}


def set_criterion(out, targets, positive_map, *, eos_coef=0.1, weights=(1.0, 5.0, 2.0), world_size=1, token_spans=None,
                  temperature=0.07, return_indices=False, indices=None):
    """SetCriterion.forward, non-list branch, mdetr.py:990-1021 (labels, boxes, cardinality
    [, contrastive_align]; aux layers re-matched).  `indices` (tests only): assignments to use instead of matching, one list of
    (src, tgt) pairs per layer in the order main, aux 0, aux 1, ... -- a gradient comparison between two implementations
    whose outputs differ in the last bits must not be dominated by a near-tied assignment that flipped."""
    given = list(indices) if indices is not None else None

    def match(o):
        if given is not None:
            return given.pop(0)
        return matcher_ref.hungarian_match(o["pred_logits"], o["pred_boxes"], [t["boxes"] for t in targets], positive_map, *weights)

    num_boxes = max(float(sum(len(t["boxes"]) for t in targets)) / world_size, 1.0)
    all_idx = []

    def layer_losses(o, suffix):
        idx = match(o)
        all_idx.append(idx)
        d = {}
        d.update(loss_labels(o, targets, positive_map, idx, num_boxes, eos_coef))
        d.update(loss_boxes(o, targets, idx, num_boxes))
        d.update(loss_cardinality(o, targets))
        if token_spans is not None and "proj_queries" in o:
            d.update(loss_contrastive_align(o, token_spans, idx, num_boxes, temperature))
        return {k + suffix: v for k, v in d.items()}

    losses = layer_losses(out, "")
    for i, a in enumerate(out.get("aux_outputs", [])):
        losses.update(layer_losses(a, f"_{i}"))
    return (losses, all_idx) if return_indices else losses


def post_process(out, target_sizes):
    """PostProcess.forward, postprocessors.py:19-56."""
    prob = F.softmax(out["pred_logits"], -1)
    scores = 1 - prob[:, :, -1]
    labels = torch.ones(prob.shape[:2], dtype=torch.int64)
    boxes = matcher_ref.cxcywh_to_xyxy(out["pred_boxes"])
    h, w = target_sizes.unbind(1)
    boxes = boxes * torch.stack([w, h, w, h], dim=1)[:, None, :]
    return [{"scores": s, "labels": l, "boxes": b} for s, l, b in zip(scores, labels, boxes)]


# ------------------------------------------------------------------------------------------ segmentation (config 3)


def attention_map(sd, p, hs, memory, mask, nhead=8):
    """MHAttentionMap.forward, /root/reference/models/segmentation.py:262-273: `flatten(3)` = softmax over H x W per head.
    hs [B,Q,d], memory [B,d,h,w], mask [B,h,w] bool -> [B,Q,nhead,h,w]."""
    d = hs.shape[-1]
    q = F.linear(hs, sd[p + "q_linear.weight"], sd[p + "q_linear.bias"])
    k = F.conv2d(memory, sd[p + "k_linear.weight"].unsqueeze(-1).unsqueeze(-1), sd[p + "k_linear.bias"])
    qh = q.view(q.shape[0], q.shape[1], nhead, d // nhead)
    kh = k.view(k.shape[0], nhead, d // nhead, k.shape[-2], k.shape[-1])
    w = torch.einsum("bqnc,bnchw->bqnhw", qh * float(d / nhead) ** -0.5, kh)
    if mask is not None:
        w = w.masked_fill(mask.unsqueeze(1).unsqueeze(1), float("-inf"))
    return F.softmax(w.flatten(3), dim=-1).view_as(w)


def mask_head(sd, p, x, bbox_mask, fpns):
    """MaskHeadSmallConv.forward, segmentation.py:203-241 (GroupNorm(8), nearest 2x upsampling + 1x1 FPN adapters).
    x [B,d,h,w], bbox_mask [B,Q,nhead,h,w], fpns = [C4, C3, C2] -> [B*Q,1,8h,8w]."""
    Q = bbox_mask.shape[1]

    def expand(t, n):
        return t.unsqueeze(1).repeat(1, int(n), 1, 1, 1).flatten(0, 1)

    def block(y, i):
        y = F.conv2d(y, sd[f"{p}lay{i}.weight"], sd[f"{p}lay{i}.bias"], padding=1)
        return F.relu(F.group_norm(y, 8, sd[f"{p}gn{i}.weight"], sd[f"{p}gn{i}.bias"]))

    y = torch.cat([expand(x, Q), bbox_mask.flatten(0, 1)], 1)
    y = block(block(y, 1), 2)
    for i, f in enumerate(fpns, start=1):
        cur = F.conv2d(f, sd[f"{p}adapter{i}.weight"], sd[f"{p}adapter{i}.bias"])
        if cur.shape[0] != y.shape[0]:
            cur = expand(cur, y.shape[0] / cur.shape[0])
        y = cur + F.interpolate(y, size=cur.shape[-2:], mode="nearest")
        y = block(y, i + 2)
    return F.conv2d(y, sd[p + "out_lay.weight"], sd[p + "out_lay.bias"], padding=1)


def segm_decode(sd, mc, out, feats, src_proj, feat_mask, *, nhead=8, prefix="detr."):
    """The mask branch of DETRsegm.forward, segmentation.py:154-167.  `out` comes from mdetr_decode (needs 'hs');
    feats = [C2, C3, C4, C5]; src_proj [B,d,h,w]."""
    L = mc["text_memory"].shape[0]
    memory = mc["img_memory"][:-L].permute(1, 2, 0).reshape(src_proj.shape)
    P = prefix[:-5] if prefix.endswith("detr.") else ""
    bbox_mask = attention_map(sd, P + "bbox_attention.", out["hs"][-1], memory, feat_mask, nhead)
    seg = mask_head(sd, P + "mask_head.", src_proj, bbox_mask, [feats[2], feats[1], feats[0]])
    B, Q = bbox_mask.shape[:2]
    return seg.view(B, Q, seg.shape[-2], seg.shape[-1])


def dice_loss(inputs, targets, num_boxes):
    """segmentation.py:276-291"""
    p = inputs.sigmoid().flatten(1)
    num = 2 * (p * targets).sum(1)
    den = p.sum(-1) + targets.sum(-1)
    return (1 - (num + 1) / (den + 1)).sum() / num_boxes


def sigmoid_focal_loss(inputs, targets, num_boxes, alpha=0.25, gamma=2.0):
    """segmentation.py:294-319"""
    prob = inputs.sigmoid()
    ce = F.binary_cross_entropy_with_logits(inputs, targets, reduction="none")
    p_t = prob * targets + (1 - prob) * (1 - targets)
    loss = ce * ((1 - p_t) ** gamma)
    loss = (alpha * targets + (1 - alpha) * (1 - targets)) * loss
    return loss.mean(1).sum() / num_boxes


def loss_masks(pred_masks, targets, indices, num_boxes):
    """SetCriterion.loss_masks, mdetr.py:827-853: bilinear upsample of the matched predictions to the (padded) GT
    size, focal + dice."""
    batch, src = _flat_indices(indices)
    tb = torch.cat([torch.full_like(j, i) for i, (_, j) in enumerate(indices)])
    tj = torch.cat([j for _, j in indices])
    H = max(t["masks"].shape[-2] for t in targets)
    W = max(t["masks"].shape[-1] for t in targets)
    maxT = max(t["masks"].shape[0] for t in targets)
    padded = torch.zeros(len(targets), maxT, H, W)
    for i, t in enumerate(targets):
        m = t["masks"]
        padded[i, : m.shape[0], : m.shape[1], : m.shape[2]] = m.float()
    sm = pred_masks[batch, src]
    sm = F.interpolate(sm[:, None], size=(H, W), mode="bilinear", align_corners=False)[:, 0].flatten(1)
    tm = padded[tb, tj].flatten(1)
    return {"loss_mask": sigmoid_focal_loss(sm, tm, num_boxes), "loss_dice": dice_loss(sm, tm, num_boxes)}
