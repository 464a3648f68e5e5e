"""toist_amd.harness.CapturedTrainStep: the reference's training step (engine.py:54-101) replayed from one hipGraph per padded input
shape, fed with batches of different image sizes, caption lengths and numbers of targets -- against the same steps launched eagerly on
the same padded inputs (dropout off: the replayed graph bakes the host-side seed counter of its capture)."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _batches():
    from toist_amd import harness
    # (height, width, tokens, max targets): two image sizes and two caption lengths -> two buckets at pad_hw = 64, pad_tokens = 8;
    # 0 .. 5 targets per image, incl. an image without targets
    spec = [(128, 160, 12, 4), (192, 128, 16, 5), (120, 150, 10, 0), (128, 190, 9, 3), (180, 100, 14, 5), (100, 130, 16, 2)]
    out = []
    for i, (h, w, t, mt) in enumerate(spec):
        out.append(harness.synthetic_batch(2, h, w, tokens=t, seed=40 + i, max_targets=mt))
    return out


def test_captured_step_matches_eager_on_a_mixed_stream(dev):
    import toist_amd
    from toist_amd import engine, harness, kernels
    from toist_amd.matcher import StaticTargets
    from toist_amd.misc import NestedTensor
    from toist_amd.optim import FusedClipAdamWEMA
    from toist_amd.transformer import TokenizedText
    args = harness.default_args(device="cuda", enc_layers=1, dec_layers=2, num_queries=20, dropout=0.0, contrastive_align_loss=True)
    torch.manual_seed(0)
    model0, criterion, _, weight_dict = toist_amd.build_model(args)
    model0.to(dev).train()
    model0.transformer.text_encoder.config.hidden_dropout_prob = 0.0
    model0.transformer.text_encoder.config.attention_probs_dropout_prob = 0.0
    criterion.train()
    kernels.SEED_DEV = torch.zeros(1, dtype=torch.int64, device=dev)
    old_reuse = engine.REUSE_GRAD_BUFFERS

    def opt_of(model):
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        return FusedClipAdamWEMA([{"params": [p for n, p in named if "backbone" not in n and "text_encoder" not in n], "lr": 1e-4},
                                  {"params": [p for n, p in named if "backbone" in n], "lr": 1e-5},
                                  {"params": [p for n, p in named if "text_encoder" in n], "lr": 5e-5}], weight_decay=1e-4, max_norm=0.1)

    try:
        cap_model, eag_model = copy.deepcopy(model0), copy.deepcopy(model0)
        cap = harness.CapturedTrainStep(cap_model, criterion, opt_of(cap_model), weight_dict, batch=2, max_targets_per_image=6, pad_hw=64, pad_tokens=8,
                                        max_graphs=4)
        eag_opt = opt_of(eag_model)
        stream = _batches() + _batches()[:4]                  # 10 steps: every bucket is hit eagerly once, then replayed
        got, ref, keys, drift = [], [], [], []
        cap_opt = cap.optimizer
        for samples, tok, targets, pmap in stream:
            key = cap.bucket_of(samples, tok)
            keys.append(key)
            # both loops start every step from the SAME weights and optimizer moments (a random-init model trained on changing batches
            # is chaotic: one flipped near-tied Hungarian assignment moves the loss by 10 %): each step is an independent comparison
            with torch.no_grad():
                for p_e, p_c in zip(eag_model.parameters(), cap_model.parameters()):
                    p_e.copy_(p_c)
                for a_e, a_c in zip(eag_opt.exp_avg + eag_opt.exp_avg_sq, cap_opt.exp_avg + cap_opt.exp_avg_sq):
                    a_e.copy_(a_c)
                eag_opt.state.copy_(cap_opt.state)
            engine.bump_weight_epoch()
            got.append(float(cap.step(samples.to(dev), tok.to(dev), targets, pmap).detach()))
            # the same step, eagerly, on the same padded inputs
            Hp, Wp, Lp = key
            B, _, H, W = samples.tensors.shape
            img = torch.zeros(B, 3, Hp, Wp)
            msk = torch.ones(B, Hp, Wp, dtype=torch.bool)
            img[:, :, :H, :W] = samples.tensors
            msk[:, :H, :W] = samples.mask
            ids = torch.full((B, Lp), 1, dtype=torch.int64)
            att = torch.zeros(B, Lp, dtype=torch.int64)
            L = tok["input_ids"].shape[1]
            ids[:, :L] = tok["input_ids"]
            att[:, :L] = tok["attention_mask"]
            s2, t2 = NestedTensor(img.to(dev), msk.to(dev)), TokenizedText({"input_ids": ids.to(dev), "attention_mask": att.to(dev)})
            st = StaticTargets(2, 6, 20, 256, dev)
            st.load(targets, pmap, criterion.token_masks_host(targets, tok))
            eag_opt.zero_grad(set_to_none=True)
            mc = eag_model(s2, t2, encode_and_save=True)
            out = eag_model(s2, t2, encode_and_save=False, memory_cache=mc)
            total = toist_amd.weighted_total(criterion(mc, out, st, None, None), weight_dict)
            total.backward()
            eag_opt.step()
            ref.append(float(total.detach()))
            worst = 0.0
            for (n, p), (_, q) in zip(cap_model.named_parameters(), eag_model.named_parameters()):
                if p.requires_grad:
                    worst = max(worst, float((p - q).detach().abs().max()))
            drift.append(worst)
        assert len(set(keys)) >= 2 and cap.captures == len(set(keys)) and cap.replays == len(stream) - len(set(keys)), (keys, cap.captures, cap.replays)
        assert all(map(lambda v: v == v and abs(v) < 1e6, got)), got
        # same weights, same inputs: the replayed forward reproduces the eager loss (fp32 atomics in the norm / embedding gradients only
        # touch the step AFTER), and one optimizer step moves no parameter further apart than two learning-rate-sized updates
        for i, (a, b) in enumerate(zip(got, ref)):
            assert abs(a - b) <= 2e-3 * abs(b) + 1e-4, (i, keys[i], got, ref)
        assert max(drift) <= 4e-4, drift
        assert len(set(round(v, 3) for v in got)) > 3                    # different batches really went through the graphs
        # LRU: a third and a fourth bucket evict nothing yet, a sixth would
        assert len(cap._buckets) <= cap.max_graphs
    finally:
        engine.REUSE_GRAD_BUFFERS = old_reuse


def test_captured_step_lru_evicts_the_oldest_bucket(dev):
    import toist_amd
    from toist_amd import engine, harness, kernels
    from toist_amd.optim import FusedClipAdamWEMA
    args = harness.default_args(device="cuda", enc_layers=1, dec_layers=1, num_queries=10, dropout=0.0)
    torch.manual_seed(0)
    model, criterion, _, weight_dict = toist_amd.build_model(args)
    model.to(dev).train()
    kernels.SEED_DEV = torch.zeros(1, dtype=torch.int64, device=dev)
    old_reuse = engine.REUSE_GRAD_BUFFERS
    try:
        opt = FusedClipAdamWEMA([{"params": [p for p in model.parameters() if p.requires_grad]}], lr=1e-5, max_norm=0.1)
        cap = harness.CapturedTrainStep(model, criterion, opt, weight_dict, batch=1, max_targets_per_image=4, pad_hw=64, pad_tokens=8, max_graphs=2)
        for h, w in ((64, 64), (64, 128), (128, 64), (64, 64)):
            samples, tok, targets, pmap = harness.synthetic_batch(1, h, w, tokens=8, seed=h + w, max_targets=3)
            loss = cap.step(samples.to(dev), tok.to(dev), targets, pmap)
            assert bool(torch.isfinite(loss))
        assert len(cap._buckets) == 2 and (64, 64, 8) in cap._buckets and (128, 64, 8) in cap._buckets and cap.captures == 4
    finally:
        engine.REUSE_GRAD_BUFFERS = old_reuse


def test_caption_length_is_not_padded_by_default(dev):
    """extra pad tokens enter loss_contrastive_align (the reference takes its log-sum-exp over every token column, mdetr.py:646-663),
    so the default bucket keeps the batch's own caption length; only the image sides are rounded up.  The loss dict of a captured step on
    a 10-token batch then equals the eager criterion on the UNPADDED batch, contrastive term included."""
    import toist_amd
    from toist_amd import harness, kernels
    from toist_amd.optim import FusedClipAdamWEMA
    args = harness.default_args(device="cuda", enc_layers=1, dec_layers=2, num_queries=20, dropout=0.0, contrastive_align_loss=True)
    torch.manual_seed(0)
    model, criterion, _, weight_dict = toist_amd.build_model(args)
    model.to(dev).train()
    model.transformer.text_encoder.config.hidden_dropout_prob = 0.0
    model.transformer.text_encoder.config.attention_probs_dropout_prob = 0.0
    kernels.SEED_DEV = torch.zeros(1, dtype=torch.int64, device=dev)
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    opt = FusedClipAdamWEMA([{"params": [p for _, p in named], "lr": 0.0}], weight_decay=0.0, max_norm=0.1)      # lr 0: the weights stay put
    cap = harness.CapturedTrainStep(model, criterion, opt, weight_dict, batch=2, max_targets_per_image=6)
    samples, tok, targets, pmap = harness.synthetic_batch(2, 128, 128, tokens=10, seed=3, max_targets=4)
    assert cap.bucket_of(samples, tok) == (128, 128, 10)
    t_dev = [{k_: (v.to(dev) if torch.is_tensor(v) else v) for k_, v in t.items()} for t in targets]
    # no_grad: a grad-enabled forward that is never followed by backward() keeps its AccumulateGrad nodes pinned to the stream it ran
    # on; the backward pass captured later on the step's own stream then drags that stream into the capture (torch warns "the
    # AccumulateGrad node's stream does not match ..." and hipStreamEndCapture fails) -- evaluation between training steps belongs
    # under no_grad, as in the reference's evaluate() (engine.py:104 @torch.no_grad)
    with torch.no_grad():
        mc = model(samples.to(dev), tok.to(dev), encode_and_save=True)
        out = model(samples.to(dev), tok.to(dev), encode_and_save=False, memory_cache=mc)
        want = float(toist_amd.weighted_total(criterion(mc, out, t_dev, pmap.to(dev), None), weight_dict))
    got = [float(cap.step(samples, tok, targets, pmap)) for _ in range(3)]          # eager first step, capture, replay
    for g in got:
        assert abs(g - want) <= 2e-3 * abs(want), (got, want)


@pytest.mark.parametrize("hw,tol", [((128, 192), 3e-3), ((120, 180), 4e-2)])
def test_captured_step_with_mask_losses(dev, hw, tol):
    """(120 x 180: image sides that are NOT multiples of pad_hw -- ADVICE r5.  The bucket is 128 x 192; StaticTargets.valid_hw makes the mask losses
    those of the 120 x 180 batch (geometry and normalisation: tests/test_gpu_segm.py::test_mask_losses_of_a_batch_do_not_depend_on_its_bucket is the
    exact statement), what remains is the padded margin seen by the convolutions next to the image border -- the same effect a smaller image has
    inside a reference batch; bounded here.)
    configs[2] through CapturedTrainStep: the ground-truth masks travel in StaticTargets(mask_hw=...), the mask losses run on a
    fixed-capacity pair table whose live slots / image indices are computed on the device (segmentation.mask_losses_static), so ONE graph serves
    batches with different numbers of targets.  Each step is compared with the same step launched eagerly through the LIST-of-dicts criterion path
    (segmentation.mask_losses: an independent implementation of the pair tables) from the same weights."""
    import toist_amd
    from toist_amd import engine, harness, kernels
    from toist_amd.optim import FusedClipAdamWEMA
    args = harness.default_args(device="cuda", enc_layers=1, dec_layers=2, num_queries=20, dropout=0.0, masks=True, mask_model="smallconv")
    torch.manual_seed(0)
    model0, criterion, _, weight_dict = toist_amd.build_model(args)
    assert "masks" in criterion.losses
    model0.to(dev).train()
    det = model0.detr
    det.transformer.text_encoder.config.hidden_dropout_prob = 0.0
    det.transformer.text_encoder.config.attention_probs_dropout_prob = 0.0
    criterion.train()
    kernels.SEED_DEV = torch.zeros(1, dtype=torch.int64, device=dev)
    old_reuse = engine.REUSE_GRAD_BUFFERS

    def opt_of(model):
        return FusedClipAdamWEMA([{"params": [p for p in model.parameters() if p.requires_grad], "lr": 1e-4}], weight_decay=1e-4, max_norm=0.1)

    try:
        cap_model, eag_model = copy.deepcopy(model0), copy.deepcopy(model0)
        cap = harness.CapturedTrainStep(cap_model, criterion, opt_of(cap_model), weight_dict, batch=2, max_targets_per_image=6, pad_hw=64, pad_tokens=1)
        assert cap.masks
        eag_opt = opt_of(eag_model)
        cap_opt = cap.optimizer
        # one bucket (128 x 192, 12 tokens: sides are multiples of pad_hw, so the padded batch IS the batch), 0 .. 5 targets per image
        stream = [harness.synthetic_batch(2, hw[0], hw[1], tokens=12, seed=70 + i, max_targets=mt, with_masks=True) for i, mt in enumerate((4, 5, 0, 2, 5, 3))]
        got, ref = [], []
        for samples, tok, targets, pmap in stream:
            with torch.no_grad():
                for p_e, p_c in zip(eag_model.parameters(), cap_model.parameters()):
                    p_e.copy_(p_c)
                for a_e, a_c in zip(eag_opt.exp_avg + eag_opt.exp_avg_sq, cap_opt.exp_avg + cap_opt.exp_avg_sq):
                    a_e.copy_(a_c)
                eag_opt.state.copy_(cap_opt.state)
            engine.bump_weight_epoch()
            got.append(float(cap.step(samples.to(dev), tok.to(dev), targets, pmap).detach()))
            t_dev = [{k_: (v.to(dev) if torch.is_tensor(v) else v) for k_, v in t.items()} for t in targets]
            eag_opt.zero_grad(set_to_none=True)
            mc = eag_model(samples.to(dev), tok.to(dev), encode_and_save=True)
            out = eag_model(samples.to(dev), tok.to(dev), encode_and_save=False, memory_cache=mc)
            losses = criterion(mc, out, t_dev, pmap.to(dev), None)
            assert "loss_mask" in losses and "loss_dice" in losses
            total = toist_amd.weighted_total(losses, weight_dict)
            total.backward()
            eag_opt.step()
            ref.append(float(total.detach()))
        assert cap.captures == 1 and cap.replays == len(stream) - 1
        print("captured vs eager totals", hw, [round(a / b - 1, 5) for a, b in zip(got, ref)])
        for i, (a, b) in enumerate(zip(got, ref)):
            assert abs(a - b) <= tol * abs(b) + 1e-4, (i, got, ref)
        assert len(set(round(v, 3) for v in got)) > 3
    finally:
        engine.REUSE_GRAD_BUFFERS = old_reuse
