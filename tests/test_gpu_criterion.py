"""GPU parity of the fused set-criterion kernel (csrc/criterion.hip) against the oracle criterion
(oracle/model_ref.py, pinned to the reference's SetCriterion by tests/golden/criterion.npz):
loss values rtol 1e-4 on identical inputs, gradients w.r.t. logits / boxes rtol 1e-3 against fp32
autograd through the oracle, assignment indices bit-identical."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _criterion(dev):
    from toist_amd import harness
    from toist_amd.matcher import HungarianMatcher
    from toist_amd.mdetr import SetCriterion
    return SetCriterion(harness.default_args(), 255, matcher=HungarianMatcher(1.0, 5.0, 2.0), eos_coef=0.1,
                        losses=["labels", "boxes", "cardinality"], temperature=0.07).to(dev)


def test_reference_golden_losses(dev):
    """Same stacked outputs / targets as the fixture the REAL reference produced its 30 losses on."""
    d = np.load(os.path.join(G, "criterion.npz"))
    sizes = d["sizes"].tolist()
    targets = [{"boxes": torch.from_numpy(d[f"boxes{i}"]).to(dev), "labels": torch.ones(s, dtype=torch.int64, device=dev)} for i, s in enumerate(sizes)]
    logits, boxes = torch.from_numpy(d["pred_logits"]).to(dev), torch.from_numpy(d["pred_boxes"]).to(dev)
    out = {"pred_logits": logits[-1], "pred_boxes": boxes[-1], "_stacked": {"pred_logits": logits, "pred_boxes": boxes}}
    crit = _criterion(dev)
    losses = crit(None, out, targets, torch.from_numpy(d["pm"]).to(dev), None)
    ref = dict(zip([str(n) for n in d["names"]], d["values"]))
    for k_, v in losses.items():
        assert abs(float(v) - ref[k_]) <= 1e-4 * abs(ref[k_]) + 1e-5, f"{k_}: {float(v)} vs reference {ref[k_]}"
    assert set(losses) == {k_ for k_ in ref if "contrastive" not in k_}


@pytest.mark.parametrize("seed,B,sizes", [(0, 4, [3, 0, 7, 1]), (1, 8, [0] * 8), (2, 2, [10, 10])])
def test_losses_and_gradients_vs_oracle(dev, seed, B, sizes):
    from oracle import model_ref
    g = torch.Generator().manual_seed(seed)
    L, Q, K = 3, 100, 256
    logits = (torch.randn(L, B, Q, K, generator=g) * 2).requires_grad_(True)
    raw = torch.randn(L, B, Q, 4, generator=g)
    boxes = torch.cat([torch.sigmoid(raw[..., :2]) * 0.6 + 0.2, torch.sigmoid(raw[..., 2:]) * 0.35 + 0.05], -1).requires_grad_(True)
    targets, rows = [], []
    for t in sizes:
        bx = torch.cat([torch.rand(t, 2, generator=g) * 0.6 + 0.2, torch.rand(t, 2, generator=g) * 0.35 + 0.05], -1)
        targets.append({"boxes": bx, "labels": torch.ones(t, dtype=torch.int64)})
        pm = torch.rand(t, K, generator=g)
        rows.append(pm / pm.sum(-1, keepdim=True) * (0.5 + torch.rand(t, 1, generator=g)))
    pmap = torch.cat(rows) if sum(sizes) else torch.zeros(0, K)
    weights = {"loss_ce": 1.0, "loss_bbox": 5.0, "loss_giou": 2.0}

    def total_of(losses):
        return sum(v * weights[k_.split("_")[0] + "_" + k_.split("_")[1]] for k_, v in losses.items() if k_.startswith("loss_"))

    ref_out = {"pred_logits": logits[-1], "pred_boxes": boxes[-1],
               "aux_outputs": [{"pred_logits": logits[i], "pred_boxes": boxes[i]} for i in range(L - 1)]}
    ref_losses, ref_idx = model_ref.set_criterion(ref_out, targets, pmap, return_indices=True)
    total_of(ref_losses).backward()

    lg, bx = logits.detach().to(dev).requires_grad_(True), boxes.detach().to(dev).requires_grad_(True)
    out = {"pred_logits": lg[-1], "pred_boxes": bx[-1], "_stacked": {"pred_logits": lg, "pred_boxes": bx}}
    crit = _criterion(dev)
    t_dev = [{k_: v.to(dev) for k_, v in t.items()} for t in targets]
    losses = crit(None, out, t_dev, pmap.to(dev), None)
    total_of(losses).backward()
    for k_ in ref_losses:
        a, b = float(losses[k_]), float(ref_losses[k_])
        assert abs(a - b) <= 1e-4 * abs(b) + 1e-5, f"{k_}: {a} vs {b}"
    order = [L - 1] + list(range(L - 1))
    for pos, l in enumerate(order):
        for (gi, gj), (ri, rj) in zip(crit.last_match.to_list(l), ref_idx[pos]):
            assert torch.equal(gi, ri) and torch.equal(gj, rj)
    for got, ref, name in [(lg.grad, logits.grad, "dlogits"), (bx.grad, boxes.grad, "dboxes")]:
        err = (got.cpu() - ref).abs().max()
        scale = ref.abs().max() + 1e-12
        assert float(err) <= 1e-3 * float(scale) + 1e-7, f"{name}: max err {float(err)} (scale {float(scale)})"


def test_weighted_total_equals_the_key_by_key_sum(dev):
    """engine.py:77 sums the loss dict key by key; weighted_total forms the same value and the same gradients from the stacked tensors."""
    import toist_amd
    from toist_amd import harness
    torch.manual_seed(0)
    args = harness.default_args(device="cuda", enc_layers=1, dec_layers=3, num_queries=12)
    _, criterion, _, weight_dict = toist_amd.build_model(args)
    _, _, targets, pmap = harness.synthetic_batch(3, 64, 64, tokens=8, seed=2, device=dev, max_targets=4)

    def run(fused):
        g = torch.Generator().manual_seed(1)
        logits = torch.randn(3, 3, 12, 256, generator=g).to(dev).requires_grad_(True)
        boxes = (torch.rand(3, 3, 12, 4, generator=g) * 0.5 + 0.2).to(dev).requires_grad_(True)
        outputs = {"pred_logits": logits[-1], "pred_boxes": boxes[-1],
                   "aux_outputs": [{"pred_logits": logits[l], "pred_boxes": boxes[l]} for l in range(2)]}
        losses = criterion({}, outputs, targets, pmap, None)
        assert hasattr(losses, "groups") and "loss_giou_1" in losses
        total = toist_amd.weighted_total(losses, weight_dict) if fused else sum(losses[k_] * weight_dict[k_] for k_ in losses if k_ in weight_dict)
        total.backward()
        return float(total), logits.grad.clone(), boxes.grad.clone()
    a, b = run(True), run(False)
    assert abs(a[0] - b[0]) <= 1e-5 * abs(b[0])
    assert torch.allclose(a[1], b[1], rtol=1e-5, atol=1e-8) and torch.allclose(a[2], b[2], rtol=1e-5, atol=1e-8)
    # a plain dict (e.g. after reduce_dict) takes the key-by-key route
    plain = {"loss_ce": torch.tensor(2.0, device=dev), "other": torch.tensor(5.0, device=dev)}
    assert float(toist_amd.weighted_total(plain, {"loss_ce": 3.0})) == 6.0
