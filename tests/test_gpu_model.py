"""End-to-end GPU parity of the MI355X model against the fp32 CPU oracle (oracle/model_ref.py) on the
same state_dict and inputs.  Tolerances (bf16 compute vs fp32): relative Frobenius error of feature
maps / logits <= 3e-2, boxes (post-sigmoid) atol 1e-2, losses rtol 5e-2; matcher indices on the GPU
model's own fp32 outputs must be bit-identical to the oracle matcher."""
import copy
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-12))


@pytest.fixture(scope="module")
def setup(dev):
    import toist_amd
    from toist_amd import harness
    torch.manual_seed(0)
    args = harness.default_args(device="cuda")
    model, criterion, _, weight_dict = toist_amd.build_model(args)
    # give FrozenBN non-trivial statistics
    g = torch.Generator().manual_seed(1)
    for n, b in model.named_buffers():
        if n.endswith("running_var"):
            b.copy_(torch.rand(b.shape, generator=g) * 0.5 + 0.75)
        elif n.endswith("running_mean"):
            b.copy_(torch.randn(b.shape, generator=g) * 0.1)
        elif n.endswith("bn1.weight") or n.endswith("bn2.weight") or n.endswith("bn3.weight") or n.endswith("downsample.1.weight"):
            b.copy_(torch.rand(b.shape, generator=g) * 0.4 + 0.8)
        elif "bn" in n and n.endswith(".bias") or n.endswith("downsample.1.bias"):
            b.copy_(torch.randn(b.shape, generator=g) * 0.05)
    # damp the last BN of every block so 33 residual blocks do not blow activations up
    for n, b in model.named_buffers():
        if n.endswith("bn3.weight"):
            b.mul_(0.3)
    sd = {k_: v.detach().clone().float() for k_, v in model.state_dict().items()}
    model.to(dev)
    return model, criterion, weight_dict, sd, args


def test_backbone_and_encode_decode(dev, setup):
    from oracle import model_ref
    from toist_amd import harness
    model, criterion, weight_dict, sd, args = setup
    model.eval()
    samples, tok, targets, pmap = harness.synthetic_batch(2, 160, 192, tokens=16, seed=5)
    with torch.no_grad():
        feats = model.backbone[0].forward_native(samples.tensors.to(dev), (1, 2, 3, 4))
    ref_feats = model_ref.resnet_body(samples.tensors, sd, "backbone.0.body.")
    for i, (f, r) in enumerate(zip(feats, ref_feats)):
        e = rel_err(f.permute(0, 3, 1, 2), r)
        assert e < 3e-2, f"C{i+2} rel err {e}"
    with torch.no_grad():
        mc = model(samples.to(dev), tok.to(dev), encode_and_save=True)
        out = model(samples.to(dev), tok.to(dev), encode_and_save=False, memory_cache=mc)
    rmc = model_ref.mdetr_encode(sd, samples.tensors, samples.mask, tok["input_ids"], tok["attention_mask"])
    rout = model_ref.mdetr_decode(sd, rmc)
    assert rel_err(mc["text_memory_resized"], rmc["text_memory_resized"]) < 3e-2
    assert rel_err(mc["pos_embed"], rmc["pos_embed"]) < 1e-2
    assert torch.equal(mc["mask"].cpu(), rmc["mask"])
    e = rel_err(mc["img_memory"], rmc["img_memory"])
    assert e < 4e-2, f"img_memory rel err {e}"
    e = rel_err(out["pred_logits"], rout["pred_logits"])
    assert e < 5e-2, f"logits rel err {e}"
    assert float((out["pred_boxes"].cpu() - rout["pred_boxes"]).abs().max()) < 2e-2
    for a, r in zip(out["aux_outputs"], rout["aux_outputs"]):
        assert rel_err(a["pred_logits"], r["pred_logits"]) < 5e-2


# measured (160 x 192, B = 2, eval mode; profiles/r04_small_shape_grad_parity.json): the bounds sit just below the worst tensor
GRAD_COS_MIN = 0.985          # measured minimum 0.9926 (the deepest gradient paths: backbone layer2.0.conv1 0.9937), most tensors >= 0.9988
GRAD_RATIO = (0.95, 1.06)     # measured 0.998 .. 1.035


def _dump_measured(name, worst):
    import json
    import os
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, name), "w") as f:
        json.dump(worst, f, indent=1)


def test_criterion_and_gradients(dev, setup):
    from oracle import model_ref
    from toist_amd import harness
    model, criterion, weight_dict, sd, args = setup
    model.eval()  # dropout off: parity is defined in eval (dropout RNG streams cannot match)
    samples, tok, targets, pmap = harness.synthetic_batch(2, 160, 192, tokens=16, seed=6, max_targets=6)
    t_dev = [{k_: (v.to(dev) if torch.is_tensor(v) else v) for k_, v in t.items()} for t in targets]
    model.zero_grad(set_to_none=True)
    mc = model(samples.to(dev), tok.to(dev), encode_and_save=True)
    out = model(samples.to(dev), tok.to(dev), encode_and_save=False, memory_cache=mc)
    losses = criterion(mc, out, t_dev, pmap.to(dev), None)
    total = sum(losses[k_] * weight_dict[k_] for k_ in losses if k_ in weight_dict)
    total.backward()
    torch.cuda.synchronize()

    # --- oracle on the GPU model's own outputs: matcher indices must be bit-identical, losses equal ---
    lay = out["_stacked"]
    L = lay["pred_logits"].shape[0]
    match = criterion.last_match
    ref_out = {"pred_logits": lay["pred_logits"][-1].detach().cpu().float(), "pred_boxes": lay["pred_boxes"][-1].detach().cpu().float(),
               "aux_outputs": [{"pred_logits": lay["pred_logits"][i].detach().cpu().float(), "pred_boxes": lay["pred_boxes"][i].detach().cpu().float()}
                               for i in range(L - 1)]}
    ref_losses, ref_idx = model_ref.set_criterion(ref_out, targets, pmap, return_indices=True)
    order = [L - 1] + list(range(L - 1))  # oracle list: main layer first, then aux 0..L-2
    for pos, l in enumerate(order):
        got = match.to_list(l)
        for (gi, gj), (ri, rj) in zip(got, ref_idx[pos]):
            assert torch.equal(gi, ri) and torch.equal(gj, rj), f"layer {l}: assignment differs"
    assert set(ref_losses) == set(losses), (sorted(ref_losses), sorted(losses))
    for k_ in ref_losses:
        a, b = float(losses[k_]), float(ref_losses[k_])
        assert abs(a - b) <= 2e-3 + 2e-3 * abs(b), f"{k_}: {a} vs {b}"

    # --- gradients vs fp32 autograd through the oracle ---
    sdr = {k_: v.clone().requires_grad_(v.is_floating_point() and "running" not in k_ and ".bn" not in k_ and "downsample.1" not in k_)
           for k_, v in sd.items()}
    rmc = model_ref.mdetr_encode(sdr, samples.tensors, samples.mask, tok["input_ids"], tok["attention_mask"])
    rout = model_ref.mdetr_decode(sdr, rmc)
    rl = model_ref.set_criterion(rout, targets, pmap)
    rtotal = sum(rl[k_] * weight_dict[k_] for k_ in rl if k_ in weight_dict)
    rtotal.backward()
    assert abs(float(total) - float(rtotal)) < 0.05 * abs(float(rtotal)) + 0.05, (float(total), float(rtotal))
    params = dict(model.named_parameters())
    checks = ["class_embed.weight", "bbox_embed.layers.2.weight", "bbox_embed.layers.0.bias", "query_embed.weight",
              "transformer.decoder.layers.5.linear2.weight", "transformer.decoder.layers.0.cross_attn_image.in_proj_weight",
              "transformer.decoder.norm.weight", "transformer.encoder.layers.5.self_attn.in_proj_weight",
              "transformer.decoder.layers.3.self_attn.in_proj_weight", "transformer.decoder.layers.0.self_attn.in_proj_bias",
              "transformer.decoder.layers.5.cross_attn_image.in_proj_bias", "transformer.decoder.layers.2.cross_attn_image.out_proj.weight",
              "transformer.decoder.layers.1.norm1.weight", "transformer.decoder.layers.4.norm4.bias", "transformer.encoder.layers.2.self_attn.in_proj_bias",
              "transformer.encoder.layers.0.linear1.weight", "transformer.encoder.layers.0.norm1.bias", "input_proj.weight",
              "transformer.resizer.fc.weight", "transformer.text_encoder.encoder.layer.11.output.dense.weight",
              "transformer.text_encoder.encoder.layer.0.attention.self.query.weight",
              "transformer.text_encoder.embeddings.position_embeddings.weight", "backbone.0.body.layer4.2.conv3.weight",
              "backbone.0.body.layer4.0.downsample.0.weight", "backbone.0.body.layer3.10.conv2.weight", "backbone.0.body.layer2.0.conv1.weight"]
    worst = {}
    for n in checks:
        g, r = params[n].grad, sdr[n].grad
        assert g is not None, f"no grad for {n}"
        cos = float(torch.nn.functional.cosine_similarity(g.float().cpu().flatten(), r.flatten(), dim=0))
        ratio = float(g.float().norm().cpu() / (r.norm() + 1e-20))
        worst[n] = (round(cos, 4), round(ratio, 3))
    _dump_measured("model_grad_parity.json", worst)
    bad = {n: v for n, v in worst.items() if v[0] < GRAD_COS_MIN or not (GRAD_RATIO[0] < v[1] < GRAD_RATIO[1])}
    assert not bad, f"gradient mismatch (cos, norm ratio): {bad}\nall: {worst}"
    frozen = params["backbone.0.body.layer1.0.conv1.weight"]
    assert frozen.grad is None and not frozen.requires_grad


@pytest.mark.parametrize("hw", [(128, 160), (120, 180), (104, 136)])
def test_segmentation_model_end_to_end(dev, hw):
    """Config 3: build_model(masks=True) -> DETRsegm; pred_masks vs the oracle, mask losses, a backward pass.
    (120, 180) / (104, 136): image sides that are NOT multiples of 32 -- what the reference's data pipeline delivers (datasets/tdod.py:305-319 resizes to
    arbitrary sizes).  The FPN levels then are not exact doublings of each other (4 x 6 -> 8 x 12 -> 15 x 23 -> 30 x 45) and the head resizes its maps to
    each level's own size with F.interpolate(mode="nearest") (/root/reference/models/segmentation.py:218, 225, 232); pred_masks has C2's size
    ceil(side / 4).  Round 6: these sizes used to raise in the mask program (it assumed 2x steps)."""
    import toist_amd
    from oracle import model_ref
    from toist_amd import harness
    torch.manual_seed(0)
    args = harness.default_args(device="cuda", masks=True, mask_model="smallconv")
    model, criterion, _, weight_dict = toist_amd.build_model(args)
    assert type(model).__name__ == "DETRsegm" and hasattr(model, "detr") and "loss_dice" in weight_dict
    for n, b in model.named_buffers():
        if n.endswith("bn3.weight"):
            b.mul_(0.3)
    sd = {k_: v.detach().clone().float() for k_, v in model.state_dict().items()}
    assert "mask_head.adapter3.bias" in sd and "detr.transformer.encoder.layers.0.linear1.weight" in sd
    model.to(dev).eval()
    samples, tok, targets, pmap = harness.synthetic_batch(2, hw[0], hw[1], tokens=12, seed=9, max_targets=4, with_masks=True)
    t_dev = [{k_: (v.to(dev) if torch.is_tensor(v) else v) for k_, v in t.items()} for t in targets]
    mc = model(samples.to(dev), tok.to(dev), encode_and_save=True)
    out = model(samples.to(dev), tok.to(dev), encode_and_save=False, memory_cache=mc)
    up = lambda v, s_: (v + s_ - 1) // s_
    assert out["pred_masks"].shape == (2, 100, up(hw[0], 4), up(hw[1], 4))
    assert len(mc["features_4_mask"]) == 4 and mc["src_proj_4_mask"].shape == (2, 256, up(hw[0], 32), up(hw[1], 32))
    losses = criterion(mc, out, t_dev, pmap.to(dev), None)
    assert {"loss_mask", "loss_dice"} <= set(losses)
    total = sum(losses[k_] * weight_dict[k_] for k_ in losses if k_ in weight_dict)
    total.backward()
    torch.cuda.synchronize()
    # oracle forward on the same weights / inputs
    dsd = {k_[5:]: v for k_, v in sd.items() if k_.startswith("detr.")}
    feats = model_ref.resnet_body(samples.tensors, dsd, "backbone.0.body.")
    rmc = model_ref.mdetr_encode(dsd, samples.tensors, samples.mask, tok["input_ids"], tok["attention_mask"], features=feats[-1])
    rout = model_ref.mdetr_decode(dsd, rmc)
    src_proj = torch.nn.functional.conv2d(feats[-1], dsd["input_proj.weight"], dsd["input_proj.bias"])
    fmask = model_ref.downsample_mask(samples.mask, feats[-1].shape[-2:])
    rmasks = model_ref.segm_decode(sd, rmc, rout, feats, src_proj, fmask, prefix="detr.")
    e = rel_err(out["pred_masks"], rmasks)
    assert e < 6e-2, f"pred_masks rel err {e}"
    # mask losses on the GPU model's own predictions vs the oracle formulas (same assignment)
    L = out["_stacked"]["pred_logits"].shape[0]
    idx = criterion.last_match.to_list(L - 1)
    nb = max(float(sum(len(t["boxes"]) for t in targets)), 1.0)
    ref = model_ref.loss_masks(out["pred_masks"].detach().float().cpu(), targets, idx, nb)
    for k_ in ("loss_mask", "loss_dice"):
        assert abs(float(losses[k_]) - float(ref[k_])) <= 2e-3 * abs(float(ref[k_])) + 1e-5, (k_, float(losses[k_]), float(ref[k_]))
    g = model.mask_head.lay3.weight.grad
    assert g is not None and torch.isfinite(g).all() and float(g.abs().sum()) > 0
    assert model.detr.backbone[0].body.layer2[0].conv1.weight.grad is not None


def test_stream_overlap_is_bit_identical(dev, setup):
    """Forking the text branch onto a side HIP stream (engine.OVERLAP) only reorders independent launches: loss and every parameter gradient must be bit-identical to the
    single-stream run (train mode, same dropout seeds)."""
    from toist_amd import engine, harness
    model, criterion, weight_dict, sd, args = setup
    samples, tok, targets, pmap = harness.synthetic_batch(2, 160, 192, tokens=16, seed=8, max_targets=6)
    t_dev = [{k_: (v.to(dev) if torch.is_tensor(v) else v) for k_, v in t.items()} for t in targets]

    def run(mode):
        engine.OVERLAP = mode
        model.train()
        model.transformer._step = 1234
        model.zero_grad(set_to_none=True)
        mc = model(samples.to(dev), tok.to(dev), encode_and_save=True)
        out = model(samples.to(dev), tok.to(dev), encode_and_save=False, memory_cache=mc)
        losses = criterion(mc, out, t_dev, pmap.to(dev), None)
        total = sum(losses[k_] * weight_dict[k_] for k_ in losses if k_ in weight_dict)
        total.backward()  # no device-wide synchronize: the clones below run on the main stream and rely on the joins
        return float(total), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}

    try:
        l0, g0 = run("off")
        l0b, g0b = run("off")
        l1, g1 = run("on")
        l2, g2 = run("on")
    finally:
        engine.OVERLAP = "capture"
        model.eval()
    assert l0 == l0b == l1 == l2
    assert set(g0) == set(g1) and len(g0) > 300
    # LayerNorm gamma/beta and embedding-table gradients are summed with fp32 atomics (rounding differs from run
    # to run even on one stream): tolerance for those, bit-identity for every GEMM-produced gradient
    atomic = lambda n: "norm" in n.lower() or "embeddings" in n or n.endswith("query_embed.weight")
    diff = [n for n in g0 if not atomic(n) and not (torch.equal(g0[n], g1[n]) and torch.equal(g0[n], g2[n]) and torch.equal(g0[n], g0b[n]))]
    assert not diff, f"{len(diff)} gradients differ between single-stream and overlapped runs, e.g. {diff[:5]}"
    for n in g0:
        if atomic(n):
            torch.testing.assert_close(g1[n], g0[n], rtol=1e-3, atol=1e-5 * float(g0[n].abs().max()) + 1e-12)


def test_ragged_batch_padding_masks(dev, setup):
    """Images of different sizes padded by NestedTensor.from_tensor_list (util/misc.py:185-209) and captions of
    different lengths: the key-padding masks of both modalities, the mask-aware sine encoding and the <pad>-aware
    RoBERTa position ids must reproduce the oracle (same tolerances as the dense case); matcher bit-identical."""
    from oracle import matcher_ref, model_ref
    from toist_amd import harness
    from toist_amd.misc import NestedTensor
    from toist_amd.transformer import TokenizedText
    model, criterion, weight_dict, sd, args = setup
    model.eval()
    g = torch.Generator().manual_seed(21)
    imgs = [torch.randn(3, 160, 192, generator=g), torch.randn(3, 128, 150, generator=g), torch.randn(3, 96, 192, generator=g)]
    samples = NestedTensor.from_tensor_list(imgs)
    assert samples.mask.any() and not samples.mask.all()
    ids = torch.randint(3, 50265, (3, 12), generator=g)
    att = torch.ones(3, 12, dtype=torch.int64)
    ids[:, 0] = 0
    for b, n in enumerate([12, 7, 9]):          # <s> ... </s> <pad>*
        ids[b, n - 1] = 2
        ids[b, n:] = 1
        att[b, n:] = 0
    tok = TokenizedText({"input_ids": ids, "attention_mask": att})
    with torch.no_grad():
        mc = model(samples.to(dev), tok.to(dev), encode_and_save=True)
        out = model(samples.to(dev), tok.to(dev), encode_and_save=False, memory_cache=mc)
    rmc = model_ref.mdetr_encode(sd, samples.tensors, samples.mask, ids, att)
    rout = model_ref.mdetr_decode(sd, rmc)
    assert torch.equal(mc["mask"].cpu(), rmc["mask"])
    keep = ~rmc["mask"].t().unsqueeze(-1)                      # [S,B,1]: padded positions carry garbage in both
    assert rel_err(mc["img_memory"].cpu() * keep, rmc["img_memory"] * keep) < 3e-2
    assert rel_err(out["pred_logits"], rout["pred_logits"]) < 5e-2
    assert (out["pred_boxes"].cpu() - rout["pred_boxes"]).abs().max() < 2e-2
    # matcher on the GPU model's own outputs, targets with 0 / 3 / 5 boxes
    _, _, targets, pmap = harness.synthetic_batch(3, 64, 64, tokens=12, seed=22, max_targets=5)
    lo, bo = out["pred_logits"].float(), out["pred_boxes"].float()
    got = criterion.matcher({"pred_logits": lo, "pred_boxes": bo}, [{k_: (v.to(dev) if torch.is_tensor(v) else v) for k_, v in t.items()} for t in targets],
                            pmap.to(dev))
    ref = matcher_ref.hungarian_match(lo.cpu(), bo.cpu(), [t["boxes"] for t in targets], pmap)
    for (gi, gj), (ri, rj) in zip(got, ref):
        assert torch.equal(gi.cpu(), ri) and torch.equal(gj.cpu(), rj)


def test_reused_gradient_buffers_match_fresh_ones(dev, setup):
    """engine.REUSE_GRAD_BUFFERS (opt-in for training loops): gradients of consecutive steps equal the ones computed with
    freshly allocated buffers, and gradient accumulation (two backward passes without zero_grad) still sums."""
    from toist_amd import engine, harness
    model, criterion, weight_dict, sd, args = setup
    model.eval()
    samples, tok, targets, pmap = harness.synthetic_batch(2, 160, 192, tokens=16, seed=12, max_targets=6)
    t_dev = [{k_: (v.to(dev) if torch.is_tensor(v) else v) for k_, v in t.items()} for t in targets]

    def fwd_bwd():
        mc = model(samples.to(dev), tok.to(dev), encode_and_save=True)
        out = model(samples.to(dev), tok.to(dev), encode_and_save=False, memory_cache=mc)
        losses = criterion(mc, out, t_dev, pmap.to(dev), None)
        sum(losses[k_] * weight_dict[k_] for k_ in losses if k_ in weight_dict).backward()

    names = ["class_embed.weight", "transformer.encoder.layers.0.linear1.weight", "backbone.0.body.layer3.5.conv2.weight",
             "transformer.text_encoder.encoder.layer.3.output.dense.weight"]
    params = dict(model.named_parameters())
    grads = {}
    try:
        for mode in (False, True):
            engine.REUSE_GRAD_BUFFERS = mode
            out = []
            for _ in range(3):                       # steps 2 and 3 run on the reused buffers when the option is on
                model.zero_grad(set_to_none=True)
                fwd_bwd()
                out.append({n: params[n].grad.clone() for n in names})
            fwd_bwd()                                # accumulation on top of step 3
            out.append({n: params[n].grad.clone() for n in names})
            grads[mode] = out
    finally:
        engine.REUSE_GRAD_BUFFERS = False
        model.zero_grad(set_to_none=True)
    for n in names:
        for step in range(3):
            assert torch.equal(grads[True][step][n], grads[False][0][n]), (n, step)
        torch.testing.assert_close(grads[True][3][n], 2 * grads[False][0][n], rtol=1e-5, atol=1e-8)
        torch.testing.assert_close(grads[False][3][n], 2 * grads[False][0][n], rtol=1e-5, atol=1e-8)


def test_backward_cuts_reproduce_the_uncut_gradients(dev, setup):
    """toist_amd.parallel.enable_backward_cuts splits loss.backward() into segments (head | text | backbone layer4 | layer3 | stem ..
    layer2) for the data-parallel step; the segmented pass must produce the same gradients as the plain one."""
    from toist_amd import harness, parallel
    model, criterion, weight_dict, sd, args = setup
    model.eval()
    samples, tok, targets, pmap = harness.synthetic_batch(2, 160, 192, tokens=16, seed=9, max_targets=6)
    t_dev = [{k_: (v.to(dev) if torch.is_tensor(v) else v) for k_, v in t.items()} for t in targets]

    def run(cuts):
        parallel.enable_backward_cuts(model, cuts)
        model.zero_grad(set_to_none=True)
        mc = model(samples.to(dev), tok.to(dev), encode_and_save=True)
        out = model(samples.to(dev), tok.to(dev), encode_and_save=False, memory_cache=mc)
        losses = criterion(mc, out, t_dev, pmap.to(dev), None)
        total = sum(losses[k_] * weight_dict[k_] for k_ in losses if k_ in weight_dict)
        total.backward()
        if cuts:
            text_w = model.transformer.text_encoder.encoder.layer[0].output.dense.weight
            bb_w = model.backbone[0].body.layer4[2].conv3.weight
            assert text_w.grad is None and bb_w.grad is None          # the first backward stopped at the cuts
            parallel.backward_cut(mc, "text")
            assert text_w.grad is not None and bb_w.grad is None
            # the ResNet body runs as three stage programs: layer4, then layer3, then stem .. layer2, each with its own flat gradient buffer
            l3_w, l2_w = model.backbone[0].body.layer3[10].conv2.weight, model.backbone[0].body.layer2[0].conv1.weight
            assert set(mc["_native"]["cuts"]) == {"text", "backbone", "backbone.layer3", "backbone.layer2"}
            parallel.backward_cut(mc, "backbone")
            assert bb_w.grad is not None and l3_w.grad is None and l2_w.grad is None
            parallel.backward_cut(mc, "backbone.layer3")
            assert l3_w.grad is not None and l2_w.grad is None
            parallel.backward_cut(mc, "backbone.layer2")
            assert l2_w.grad is not None
            flats = {p_.grad.untyped_storage().data_ptr() for p_ in (bb_w, l3_w, l2_w)}
            assert len(flats) == 3, "the three stages must own separate flat gradient buffers"
        return {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}

    try:
        g0 = run(False)
        g1 = run(True)
    finally:
        parallel.enable_backward_cuts(model, False)
    assert set(g0) == set(g1)
    atomic = lambda n: "norm" in n.lower() or "embeddings" in n or n.endswith("query_embed.weight")
    for n in g0:
        if atomic(n):
            torch.testing.assert_close(g1[n], g0[n], rtol=1e-3, atol=1e-5 * float(g0[n].abs().max()) + 1e-12)
        else:
            assert torch.equal(g0[n], g1[n]), n


def test_device_stager_delivers_the_collated_batch(dev):
    """Pinned-memory asynchronous staging (toist_amd.misc.DeviceStager): the tensors that arrive in HBM equal the host batch."""
    from toist_amd.misc import DeviceStager, collate_fn_plain
    g = torch.Generator().manual_seed(4)
    batch = []
    for h, w, t in [(64, 96, 2), (80, 64, 0), (96, 96, 3)]:
        tgt = {"boxes": torch.rand(t, 4, generator=g), "labels": torch.ones(t, dtype=torch.int64), "positive_map": torch.rand(t, 256, generator=g) > 0.9,
               "dataset_name": "task_3_train.json", "caption": "x"}
        batch.append(([torch.randn(3, h, w, generator=g)], [tgt]))
    host = collate_fn_plain(False, batch)
    stager = DeviceStager(dev)
    got = stager.stage(host)
    stager.wait()
    assert got["samples"].tensors.is_cuda and torch.equal(got["samples"].tensors.cpu(), host["samples"].tensors)
    assert torch.equal(got["samples"].mask.cpu(), host["samples"].mask) and torch.equal(got["positive_map"].cpu(), host["positive_map"])
    for a, b in zip(got["targets"], host["targets"]):
        assert "caption" not in a and a["dataset_name"] == b["dataset_name"] and torch.equal(a["boxes"].cpu(), b["boxes"])


def test_graph_replayed_training_matches_eager_training(dev):
    """The whole step (forward, criterion, backward, fused optimizer tail) captured in a hipGraph and replayed must train
    like the eager loop: same loss trajectory from the same initial weights (dropout off), and the loss must move --
    i.e. the replayed graph really consumes the weights the tail has just written (bf16 compute copies included)."""
    import copy
    import toist_amd
    from toist_amd import harness, kernels
    from toist_amd.optim import FusedClipAdamWEMA
    args = harness.default_args(device="cuda", enc_layers=1, dec_layers=2, num_queries=20, dropout=0.0)
    torch.manual_seed(0)
    model0, criterion, _, weight_dict = toist_amd.build_model(args)
    model0.to(dev).train()
    samples, tok, targets, pmap = harness.synthetic_batch(2, 128, 160, tokens=12, seed=5, device=dev, max_targets=4)
    kernels.SEED_DEV = torch.zeros(1, dtype=torch.int64, device=dev)

    def make(model):
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        # the reference's learning rates: Adam's normalised updates at larger rates make the trajectory chaotic (one flipped
        # Hungarian assignment moves the loss by 10 %), which would test the fp32 atomics' rounding order, not the graph
        opt = FusedClipAdamWEMA([{"params": [p for n, p in named if "backbone" not in n and "text_encoder" not in n], "lr": 1e-4},
                                 {"params": [p for n, p in named if "backbone" in n], "lr": 1e-5},
                                 {"params": [p for n, p in named if "text_encoder" in n], "lr": 5e-5}], weight_decay=1e-4, max_norm=0.1)

        def step():
            opt.zero_grad(set_to_none=True)
            mc = model(samples, tok, encode_and_save=True)
            out = model(samples, tok, encode_and_save=False, memory_cache=mc)
            losses = criterion(mc, out, targets, pmap, None)
            total = sum(losses[k_] * weight_dict[k_] for k_ in losses if k_ in weight_dict)
            total.backward()
            opt.step()
            return total.detach()
        return step

    eager_model, graph_model = copy.deepcopy(model0), copy.deepcopy(model0)
    e_step, g_step = make(eager_model), make(graph_model)
    eager = [float(e_step()) for _ in range(6)]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        replayed = [float(g_step()) for _ in range(2)]          # two eager warm-up steps (same trajectory as the eager model)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            static_loss = g_step()
    torch.cuda.current_stream().wait_stream(side)
    # the capture pass itself executed nothing: replay steps 3..6
    for _ in range(4):
        graph.replay()
        replayed.append(float(static_loss))
    assert abs(eager[-1] - eager[0]) > 1e-3 * abs(eager[0]), eager          # the loss moves
    # the first replay (step 3) starts from identical weights: the captured forward must reproduce the eager loss; later steps feel
    # the fp32-atomic summation order of the previous backward through near-tied Hungarian assignments of a random-init model
    assert abs(eager[2] - replayed[2]) <= 1e-4 * abs(eager[2]), (eager, replayed)
    for a, b in zip(eager, replayed):
        assert abs(a - b) <= 5e-2 * abs(a) + 1e-3, (eager, replayed)


def test_training_reduces_loss_on_a_fixed_batch(dev):
    """Forty optimizer steps on one synthetic batch (train mode, dropout on, fused clip + AdamW + EMA tail): the weighted loss must
    fall substantially -- an end-to-end check that every gradient has the right sign and reaches its parameter."""
    import toist_amd
    from toist_amd import harness, kernels
    from toist_amd.optim import FusedClipAdamWEMA
    args = harness.default_args(device="cuda", enc_layers=1, dec_layers=2, num_queries=20)
    torch.manual_seed(0)
    model, criterion, _, weight_dict = toist_amd.build_model(args)
    model.to(dev).train()
    criterion.train()
    samples, tok, targets, pmap = harness.synthetic_batch(2, 128, 160, tokens=12, seed=11, device=dev, max_targets=4)
    kernels.SEED_DEV = torch.zeros(1, dtype=torch.int64, device=dev)
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    opt = FusedClipAdamWEMA([{"params": [p for n, p in named if "backbone" not in n and "text_encoder" not in n], "lr": 1e-4},
                             {"params": [p for n, p in named if "backbone" in n], "lr": 1e-5},
                             {"params": [p for n, p in named if "text_encoder" in n], "lr": 5e-5}], weight_decay=1e-4, max_norm=0.1)
    history = []
    for _ in range(40):
        kernels.SEED_DEV.add_(1000003)
        opt.zero_grad(set_to_none=True)
        mc = model(samples, tok, encode_and_save=True)
        out = model(samples, tok, encode_and_save=False, memory_cache=mc)
        losses = criterion(mc, out, targets, pmap, None)
        total = toist_amd.weighted_total(losses, weight_dict)
        total.backward()
        opt.step()
        history.append(float(total.detach()))
    assert all(math.isfinite(v) for v in history), history
    first, last = sum(history[:3]) / 3, sum(history[-3:]) / 3
    assert last < 0.8 * first, history
