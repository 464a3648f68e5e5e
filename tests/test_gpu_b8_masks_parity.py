"""BASELINE configs[2] at the benchmarked shape -- B = 8, 640 x 640, 100 queries, det + mask head -- against the fp32 oracle (oracle/model_ref.py
restating /root/reference/models/segmentation.py:150-168 DETRsegm.forward, :203-241 MaskHeadSmallConv.forward, :262-273 MHAttentionMap):

  * eval-mode forward: the mask logits of all 800 (image, query) maps, element-wise and in the Frobenius norm;
  * one backward pass of the reference's mask losses (mdetr.py:827-853: sigmoid focal + dice on the matched pairs, the model's own assignment on
    both sides) against fp32 autograd through the oracle, for EVERY parameter of the mask branch (mask_head.lay1..5 / out_lay / adapter1..3 /
    gn1..5, bbox_attention.q_linear / k_linear): cosine and norm ratio.  (A random +-1 functional of the 20 M logits was tried first and is the wrong
    instrument: it cancels the smooth part of the gradient and leaves rounding noise -- cosines 0.93-0.99 against 0.99+ under the real losses.)

Round 4 held this config against the oracle by value only at B = 1 and by gradient only at 128 x 160, B = 2 (VERDICT r4 "weak" item 1); at B = 8
the dispatcher picks the large-batch kernel set (smallconv at 160 x 160 x 800 maps, grouped weight gradients, 128-wide tiles).  The measured
numbers go to gpurun_out/masks_b8_parity.json before any assertion (committed as profiles/r05_masks_grad_parity.json)."""
import json
import os

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

# bounds set from the measured values (profiles/r05_masks_grad_parity.json)
MASK_LOGIT_EXCESS = 1e-1        # |a - b| <= 1e-1 + 5e-2 |b| (the stated deviation of DESIGN.md section 5: five bf16 3x3 convolutions + GroupNorms)
MASK_LOGIT_FRO = 3e-2
GRAD_COS_MIN = 0.998            # measured minimum 0.99927 (bbox_attention.q_linear.weight)
GRAD_RATIO = (0.99, 1.01)       # measured 0.9957 .. 1.0066


def test_config2_b8_mask_logits_and_mask_branch_gradients(dev):
    import toist_amd
    from oracle import model_ref
    from toist_amd import harness
    torch.manual_seed(0)
    args = harness.default_args(device="cuda", masks=True, mask_model="smallconv")
    model, criterion, _, weight_dict = toist_amd.build_model(args)
    for n, b in model.named_buffers():          # keep 33 residual blocks of random-init weights from blowing activations up
        if n.endswith("bn3.weight"):
            b.mul_(0.3)
    sd = {k_: v.detach().clone().float() for k_, v in model.state_dict().items()}
    model.to(dev).eval()                         # deterministic forward (no dropout); gradients still flow
    B = 8
    samples, tok, targets, pmap = harness.synthetic_batch(B, 640, 640, tokens=16, seed=1002, max_targets=10, with_masks=True)
    t_dev = [{k_: (v.to(dev) if torch.is_tensor(v) else v) for k_, v in t.items()} for t in targets]
    mc = model(samples.to(dev), tok.to(dev), encode_and_save=True)
    out = model(samples.to(dev), tok.to(dev), encode_and_save=False, memory_cache=mc)
    assert out["pred_masks"].shape == (B, 100, 160, 160)
    criterion.train()
    losses = criterion(mc, out, t_dev, pmap.to(dev), None)
    toist_amd.weighted_total(losses, weight_dict).backward()          # the mask branch receives gradient from loss_mask / loss_dice only
    torch.cuda.synchronize()
    L = criterion.last_match.src.shape[0]
    idx = criterion.last_match.to_list(L - 1)                          # the main layer's assignment: used on both sides
    nb = max(float(sum(len(t["boxes"]) for t in targets)), 1.0)
    got = out["pred_masks"].detach().float().cpu()
    params = dict(model.named_parameters())
    branch = sorted(n for n in params if n.startswith(("mask_head.", "bbox_attention.")))
    grads = {n: params[n].grad.detach().float().cpu() for n in branch if params[n].grad is not None}
    del out, mc, losses
    torch.cuda.empty_cache()

    sdr = {k_: (v.clone().requires_grad_(True) if k_.startswith(("bbox_attention.", "mask_head.")) else v) for k_, v in sd.items()}
    dsd = {k_[5:]: v for k_, v in sdr.items() if k_.startswith("detr.")}
    with torch.no_grad():
        feats = model_ref.resnet_body(samples.tensors, dsd, "backbone.0.body.")
        rmc = model_ref.mdetr_encode(dsd, samples.tensors, samples.mask, tok["input_ids"], tok["attention_mask"], features=feats[-1])
        rout = model_ref.mdetr_decode(dsd, rmc)
        src_proj = torch.nn.functional.conv2d(feats[-1], dsd["input_proj.weight"], dsd["input_proj.bias"])
        fmask = model_ref.downsample_mask(samples.mask, feats[-1].shape[-2:])
    rmasks = model_ref.segm_decode(sdr, rmc, rout, feats, src_proj, fmask, prefix="detr.")
    rl = model_ref.loss_masks(rmasks, targets, idx, nb)
    (rl["loss_mask"] * weight_dict["loss_mask"] + rl["loss_dice"] * weight_dict["loss_dice"]).backward()
    ref = rmasks.detach()
    d = (got - ref).abs()
    rep = {"pred_masks (max abs err, worst excess over 5e-2|ref|, rel Frobenius)":
           [round(float(d.max()), 5), round(float((d - 5e-2 * ref.abs()).max()), 5), round(float(d.norm() / ref.norm()), 5)],
           "pred_masks max |ref|": round(float(ref.abs().max()), 3), "gradients (cosine, norm ratio)": {}}
    for n in branch:
        if n not in grads or sdr[n].grad is None:
            continue
        r = sdr[n].grad
        if float(r.norm()) == 0:
            continue
        cos = float(torch.nn.functional.cosine_similarity(grads[n].flatten(), r.flatten(), dim=0))
        rep["gradients (cosine, norm ratio)"][n] = [round(cos, 5), round(float(grads[n].norm() / r.norm()), 4)]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "masks_b8_parity.json"), "w") as f:
        json.dump(rep, f, indent=1)
    print(json.dumps(rep))
    mx, excess, fro = rep["pred_masks (max abs err, worst excess over 5e-2|ref|, rel Frobenius)"]
    assert excess <= MASK_LOGIT_EXCESS, f"mask logits exceed 1e-1 + 5e-2|ref| by {excess - MASK_LOGIT_EXCESS} (max abs {mx})"
    assert fro < MASK_LOGIT_FRO, fro
    checked = rep["gradients (cosine, norm ratio)"]
    # every weight of the branch must have been compared; k_linear.bias is a per-(query, head) constant under the softmax (exact gradient 0)
    need = {n for n in branch if not n.endswith("k_linear.bias")}
    assert need <= set(checked), sorted(need - set(checked))
    bad = {n: v for n, v in checked.items() if not n.endswith("k_linear.bias") and (v[0] < GRAD_COS_MIN or not (GRAD_RATIO[0] < v[1] < GRAD_RATIO[1]))}
    assert not bad, f"mask-branch gradient mismatch at B = 8 (cosine, norm ratio): {bad}"
