"""BASELINE configs[4] at its own size on one GPU: noun-pronoun distillation, B = 4 (noun, pronoun) pairs of 640 x 640
images + 16-token captions, 100 queries, 6 + 6 layers, 1024-slot memory banks with full_label = 1 -- the step `bench.py --distill` times
(/root/reference/engine.py:152-204, models/mdetr.py:29-312, 520-599, 668-781, 887-987).

An fp32 oracle forward of two full models takes minutes on the CPU, so -- as tests/test_gpu_baseline_shapes.py does for configs[1] -- everything
is checked on the HIP path's OWN outputs: the 12 Hungarian assignments of each side bit-identical to the oracle matcher, every LSAP the step
solves on the device (24 softkd problems of (Q - c)^2, the nearest-replacement bank updates of 1 x 1024) bit-identical to oracle/lsap.c
(pinned to SciPy) on the very cost matrices the device saw, the whole loss dict against oracle/model_ref.set_criterion (noun_ / sth_ keys) and
oracle/distill_ref (softkd per layer, nsthl2, bank update, k-means prototypes, substitution, loss_cluster_feature), finite gradients in both
models."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _damp(model):
    for n, b in model.named_buffers():          # keep 33 residual blocks of random-init weights from blowing activations up
        if n.endswith("bn3.weight"):
            b.mul_(0.3)


def test_config4_distillation_step_at_baseline_size(dev):
    import toist_amd
    from oracle import distill_ref, lsap, model_ref
    from toist_amd import harness
    from toist_amd import matcher as tm
    from toist_amd.distill import task_index
    B, Q, LAYERS, MEM, D, TOK = 4, 100, 6, 1024, 256, 16
    args = harness.default_args(device="cuda", distillation=True, cluster=True, nsthl2_loss=True, softkd_loss=True, train_batch_size=B)
    assert (args.num_queries, args.enc_layers, args.dec_layers, args.cluster_memory_size) == (Q, 6, 6, MEM)
    torch.manual_seed(0)
    model, criterion, cc, weight_dict = toist_amd.build_model(args)
    model_noun, _, _, _ = toist_amd.build_model(args)
    for m in (model, model_noun):
        _damp(m)
        m.to(dev).train()
    cc.to(dev)
    # banks in steady state: full, three separated groups per task (k-means is then insensitive to the last bits of a distance), stored centres near them
    g = torch.Generator().manual_seed(11)
    proto = torch.randn(14, 3, D, generator=g) * 0.5
    bank0 = (proto[:, torch.arange(MEM) % 3] + 0.05 * torch.randn(14, MEM, D, generator=g)).contiguous()
    cc.feature_bank.copy_(bank0)
    cc.cluster_centers.copy_(proto + 0.02 * torch.randn(14, 3, D, generator=g))
    centers0 = cc.cluster_centers.detach().cpu().clone()
    cc.full_label.fill_(1)
    cc.update_count.fill_(5000)
    cc.sync_host_state()
    batch = harness.synthetic_distill_batch(B, 640, 640, tokens=TOK, seed=1000, device=dev)
    assert sum(len(t["boxes"]) for t in batch["targets"][0]) > 0

    # every LSAP the step hands to the device kernel, with the cost buffer it saw
    seen, real = [], tm.lsap_blocks

    def spy(cost, shapes, offsets, ld):
        out = real(cost, shapes, offsets, ld)
        seen.append((cost.detach().clone(), list(shapes), list(offsets), ld, out))
        return out

    tm.lsap_blocks = spy
    try:
        s_noun, s_sth = batch["samples"]
        t_noun, t_sth = batch["targets"]
        c_noun, c_sth = batch["captions"]
        k_noun, k_sth = batch["tokenized"]
        mc_noun = model_noun(s_noun, k_noun, encode_and_save=True)
        text_noun = mc_noun["text_memory"].detach().float().cpu().clone()          # [L, B, d]
        img_noun = mc_noun["img_memory"].detach().float().cpu().clone()
        mc_noun = cc.update_memory(mc_noun, t_noun, c_noun)
        bank1 = cc.feature_bank.detach().cpu().clone()
        centers1 = cc.cluster_centers.detach().cpu().clone()
        mod_noun = mc_noun["img_memory_mod"].detach().float().cpu().clone()
        out_noun = model_noun(s_noun, k_noun, encode_and_save=False, memory_cache=mc_noun)
        mc_sth = model(s_sth, k_sth, encode_and_save=True)
        text_sth = mc_sth["text_memory"].detach().float().cpu().clone()
        img_sth = mc_sth["img_memory"].detach().float().cpu().clone()
        mc_sth, loss_cluster = cc(mc_sth, t_sth, c_sth)
        centers2 = cc.cluster_centers.detach().cpu().clone()
        mod_sth = mc_sth["img_memory_mod"].detach().float().cpu().clone()
        out_sth = model(s_sth, k_sth, encode_and_save=False, memory_cache=mc_sth)
        losses = criterion([mc_noun, mc_sth], [out_noun, out_sth], [t_noun, t_sth], batch["positive_map"], batch.get("example_rel"))
        losses.update(loss_cluster)
        total = toist_amd.weighted_total(losses, weight_dict)
        total.backward()
        torch.cuda.synchronize()
        tm.check_lsap_pending()
    finally:
        tm.lsap_blocks = real
    assert bool(torch.isfinite(total))
    never = {k_ for k_ in weight_dict if k_.startswith(("loss_nsthl2_", "loss_cluster_"))}          # mdetr.py:1093-1097: keys no loss produces
    assert set(weight_dict) - never <= set(losses)

    # ---- (1) every device LSAP against oracle/lsap.c on the same matrix: indices bit-identical ------------------------------------------
    n_soft = n_bank = 0
    for cost, shapes, offsets, ld, (ri, ci, out_off, pairs, status) in seen:
        assert int(status.abs().sum()) == 0
        flat, ri, ci = cost.reshape(-1).cpu(), ri.cpu(), ci.cpu()
        for (r, c), off, o, n in zip(shapes, offsets, out_off, pairs):
            stride = ld if ld else c
            block = flat[off:off + (r - 1) * stride + c].clone() if r else flat[:0]
            mat = torch.as_strided(block, (r, c), (stride, 1)).double().numpy() if r and c else np.zeros((r, c))
            want_r, want_c = lsap.linear_sum_assignment(mat)
            assert n == len(want_r)
            assert np.array_equal(ri[o:o + n].numpy(), want_r) and np.array_equal(ci[o:o + n].numpy(), want_c), (r, c)
            n_soft += int(r == c and r > 1)
            n_bank += int(r == 1 and c == MEM)
    assert n_soft == LAYERS * B and n_bank == B, (n_soft, n_bank)          # 24 softkd problems, one bank update per (distinct) task of the batch

    # ---- (2) the two sides' detection losses and assignments against the oracle criterion on the model's own outputs -----------------------
    host = lambda ts: [{k_: (v.cpu() if torch.is_tensor(v) else v) for k_, v in t.items()} for t in ts]
    tn_h, ts_h = host(t_noun), host(t_sth)
    pm_n, pm_s = (p.cpu() for p in batch["positive_map"])
    sides = {}
    for prefix, out, tg, pm in (("noun", out_noun, tn_h, pm_n), ("sth", out_sth, ts_h, pm_s)):
        st = out["_stacked"]
        lg, bx = st["pred_logits"].detach().float().cpu(), st["pred_boxes"].detach().float().cpu()
        ref_out = {"pred_logits": lg[-1], "pred_boxes": bx[-1], "aux_outputs": [{"pred_logits": lg[i], "pred_boxes": bx[i]} for i in range(LAYERS - 1)]}
        ref_losses, ref_idx = model_ref.set_criterion(ref_out, tg, pm, return_indices=True)
        for k_, v in ref_losses.items():
            got = float(losses[f"{prefix}_{k_}"])
            assert abs(got - float(v)) <= 2e-3 * abs(float(v)) + 1e-5, (prefix, k_, got, float(v))
        by_layer = {l: ref_idx[pos] for pos, l in enumerate([LAYERS - 1] + list(range(LAYERS - 1)))}     # the oracle lists the main layer first
        sides[prefix] = (lg, bx, by_layer)
    match = criterion.last_match                                             # the student's (sth) assignment of all layers
    for l in range(LAYERS):
        for (gi, gj), (ri_, rj_) in zip(match.to_list(l), sides["sth"][2][l]):
            assert torch.equal(gi, ri_) and torch.equal(gj, rj_), f"layer {l}: matcher assignment differs from the oracle"

    # ---- (3) softkd per layer and nsthl2 -------------------------------------------------------------------------------------------------
    (lg_n, bx_n, idx_n), (lg_s, bx_s, idx_s) = sides["noun"], sides["sth"]
    for l in range(LAYERS):
        want = float(distill_ref.loss_softkd(lg_n[l], lg_s[l], bx_n[l], bx_s[l], idx_n[l], idx_s[l]))
        got = float(losses["loss_softkd" + ("" if l == LAYERS - 1 else f"_{l}")])
        assert abs(got - want) <= 2e-3 * abs(want) + 1e-6, (l, got, want)
    tok_n, tok_s = k_noun.to("cpu"), k_sth.to("cpu")
    want = float(distill_ref.loss_nsthl2(text_noun.permute(1, 0, 2), text_sth.permute(1, 0, 2), tok_n, tok_s, tn_h, ts_h, [len(s) for s, _ in idx_s[LAYERS - 1]]))
    assert abs(float(losses["loss_nsthl2"]) - want) <= 1e-3 * abs(want) + 1e-9, (float(losses["loss_nsthl2"]), want)

    # ---- (4) memory bank, k-means prototypes, substitution, loss_cluster_feature -----------------------------------------------------------
    bank, centers = bank0.clone(), centers0.clone()
    feats = distill_ref.noun_features(text_noun.permute(1, 0, 2), tok_n, tn_h)
    live = [i for i, t in enumerate(tn_h) if len(t["boxes"])]
    tasks = [task_index(t["dataset_name"]) for t in tn_h]
    for i in live:                      # update_memory_queue: nearest (L1) bank row replaced -- the replaced ROW must be the oracle's
        bank[tasks[i]] = distill_ref.replace_nearest(bank[tasks[i]], feats[i:i + 1])
        changed_ref = (bank[tasks[i]] != bank0[tasks[i]]).any(1).nonzero().reshape(-1)
        changed_got = (bank1[tasks[i]] != bank0[tasks[i]]).any(1).nonzero().reshape(-1)
        assert torch.equal(changed_ref, changed_got), (i, changed_ref, changed_got)
    np.testing.assert_allclose(bank1.numpy(), bank.numpy(), rtol=1e-4, atol=1e-5)
    mod = img_noun.clone()
    for i in live:
        pos = distill_ref.positions(tok_n, i, [s for box in tn_h[i]["noun_tokens_positive"] for s in box], TOK)
        mod, centers[tasks[i]], _ = distill_ref.cluster_substitute(mod, TOK, i, pos, bank[tasks[i]], centers[tasks[i]], feats[i], 3)
    np.testing.assert_allclose(centers1.numpy(), centers.numpy(), rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(mod_noun.numpy(), mod.numpy(), rtol=1e-3, atol=1e-4)
    mod_s, loss_f = img_sth.clone(), 0.0
    for i, cap in enumerate(c_sth):
        beg = cap.find("something")
        pos = torch.arange(tok_s.char_to_token(i, beg), tok_s.char_to_token(i, beg + len("something") - 1) + 1)
        feature = text_sth.permute(1, 0, 2)[i][pos].mean(0)
        t = task_index(ts_h[i]["dataset_name"])
        mod_s, centers[t], centre = distill_ref.cluster_substitute(mod_s, TOK, i, pos, bank[t], centers[t], feature, 3)
        loss_f += float(torch.nn.functional.mse_loss(feature, centre))
    np.testing.assert_allclose(centers2.numpy(), centers.numpy(), rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(mod_sth.numpy(), mod_s.numpy(), rtol=1e-3, atol=1e-4)
    assert abs(float(losses["loss_cluster_feature"]) - loss_f / B) <= 1e-3 * (loss_f / B) + 1e-9
    assert float(losses["loss_cluster_choice"]) == 0.0

    # ---- (5) the backward pass reached both models with finite gradients -------------------------------------------------------------------
    for m, name in ((model, "student"), (model_noun, "teacher")):
        bad = [n for n, p in m.named_parameters() if p.requires_grad and (p.grad is None or not bool(torch.isfinite(p.grad).all()))]
        assert not bad, (name, bad[:8])
    assert float(model.transformer.text_encoder.encoder.layer[0].output.dense.weight.grad.abs().sum()) > 0
