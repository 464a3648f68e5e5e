"""XCD-resident decoder stack (csrc/xdec.hip, toist_xdec_fwd): ONE launch for the six decoder layers of
/root/reference/models/transformer.py:225-267, 362-408 against the per-op launches of toist_amd.tlayer on the same model and batch.

Both paths draw their dropout seeds in the same order and use the same hashes, so they are compared IN TRAINING MODE, dropout on:
outputs agree to bf16 rounding, every gradient (the backward pass is the per-op one in both cases and consumes what the forward saved:
q | k | v, contexts, (max, 1 / sum), pre-norm sums, statistics, hidden activations) to cosine > 0.995."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def _model(dev, **over):
    import toist_amd
    from toist_amd import harness
    torch.manual_seed(0)
    args = harness.default_args(device="cuda", contrastive_align_loss=True, **over)
    model, _, _, _ = toist_amd.build_model(args)
    for n, b in model.named_buffers():
        if n.endswith("bn3.weight"):
            b.mul_(0.3)
    return model.to(dev)


def _run(model, dev, samples, tok, w, fused, train, fused_bwd=False):
    from toist_amd import kernels as k
    from toist_amd import tlayer
    calls = []
    real = k.xdec_fwd
    k.xdec_fwd = lambda *a, **kw: (calls.append(a[:3]), real(*a, **kw))[1]
    old, old_b = tlayer.XDEC, tlayer.XDEC_BWD
    tlayer.XDEC, tlayer.XDEC_BWD = fused, fused_bwd
    bcalls = []
    real_b = k.xdec_bwd
    k.xdec_bwd = lambda *a, **kw: (bcalls.append(a[:3]), real_b(*a, **kw))[1]
    try:
        model.train(train)
        model.transformer._step = 0          # both runs draw the same seeds
        model.zero_grad(set_to_none=True)
        mc = model(samples.to(dev), tok.to(dev), encode_and_save=True)
        out = model(samples.to(dev), tok.to(dev), encode_and_save=False, memory_cache=mc)
        st = out["_stacked"]
        loss = (st["pred_logits"] * w[0]).sum() + 30 * (st["pred_boxes"] * w[1]).sum() + (st["proj_queries"] * w[2]).sum()
        loss.backward()
        torch.cuda.synchronize()
        k.xdec_check()
    finally:
        tlayer.XDEC, tlayer.XDEC_BWD = old, old_b
        k.xdec_fwd = real
        k.xdec_bwd = real_b
    grads = {n: p.grad.detach().float().clone() for n, p in model.named_parameters() if p.grad is not None}
    return st["pred_logits"].detach().float().clone(), st["pred_boxes"].detach().float().clone(), float(loss.detach()), grads, (calls, bcalls)


@pytest.mark.parametrize("B,hw,train,bwd", [(2, (160, 192), True, False), (2, (160, 192), True, True), (8, (96, 128), True, True), (10, (64, 96), True, True),
                                            (3, (128, 160), False, True), (8, (416, 512), True, True)])
def test_xcd_resident_decoder_matches_the_per_op_launches(dev, B, hw, train, bwd):
    """bwd = the data-gradient chain of the backward pass as ONE launch too (toist_xdec_bwd); (416, 512) gives 208 + 16 memory tokens: two key splits in
    the cross-attention backward, i.e. folded dq shares."""
    from toist_amd import harness
    from toist_amd import kernels as k
    model = _model(dev)
    samples, tok, _, _ = harness.synthetic_batch(B, hw[0], hw[1], tokens=16, seed=6 + B, max_targets=6)
    g = torch.Generator().manual_seed(3)
    w = [torch.randn(6, B, 100, n, generator=g).to(dev) for n in (256, 4, 64)]
    S = (hw[0] // 32) * (hw[1] // 32) + 16
    if not k.xdec_supported(B, 100, S, 6):
        pytest.skip("device without 8 XCDs x 32 CUs")
    lg_a, bx_a, loss_a, g_a, (calls_a, bcalls_a) = _run(model, dev, samples, tok, w, True, train, bwd)
    lg_b, bx_b, loss_b, g_b, (calls_b, bcalls_b) = _run(model, dev, samples, tok, w, False, train)
    assert len(calls_a) == 1 and calls_a[0][:2] == (B, 100) and not calls_b           # the fused launch really ran (once: all six layers)
    assert len(bcalls_a) == (1 if bwd else 0) and not bcalls_b
    rel = float((lg_a - lg_b).norm() / lg_b.norm())
    assert rel < 1.5e-2, rel                                                            # six layers of bf16 activations; same dropout masks
    assert float((bx_a - bx_b).abs().max()) < 2e-2
    assert set(g_a) == set(g_b)
    worst = {}
    for n in g_b:
        if g_b[n].norm() == 0 or "decoder.layers.0.self_attn.in_proj" in n or n.endswith("attention.self.key.bias"):
            continue
        cos = float(torch.nn.functional.cosine_similarity(g_a[n].flatten(), g_b[n].flatten(), dim=0))
        ratio = float(g_a[n].norm() / g_b[n].norm())
        if cos < 0.995 or not (0.97 < ratio < 1.03):
            worst[n] = (round(cos, 5), round(ratio, 4))
    assert not worst, f"{len(worst)} gradients differ between the XCD-resident and the per-op decoder forward: {dict(list(worst.items())[:12])}"


def test_xcd_resident_decoder_with_ragged_rows_and_padded_keys(dev):
    """30 queries (the last 4-row block of an image has 2 live rows, the last 16-row tile 14), images of different sizes in one batch (key padding inside
    the memory of the smaller ones), 5 decoder layers: both launches against the per-op path, training mode."""
    from toist_amd import harness, misc
    from toist_amd import kernels as k
    model = _model(dev, num_queries=30, dec_layers=5)
    parts = [harness.synthetic_batch(1, h, w_, tokens=11, seed=50 + i, max_targets=3) for i, (h, w_) in enumerate(((160, 224), (128, 160), (96, 224)))]
    samples = misc.NestedTensor.from_tensor_list([p[0].tensors[0] for p in parts])
    tok = parts[0][1]
    tok = type(tok)({key: torch.cat([p[1][key] for p in parts]) for key in ("input_ids", "attention_mask")})
    B = 3
    assert bool(samples.mask.any()) and not bool(samples.mask.all())
    g = torch.Generator().manual_seed(4)
    w = [torch.randn(5, B, 30, n, generator=g).to(dev) for n in (256, 4, 64)]
    S = (samples.tensors.shape[-2] // 32) * (samples.tensors.shape[-1] // 32) + 11
    if not k.xdec_supported(B, 30, S, 5):
        pytest.skip("device without 8 XCDs x 32 CUs")
    lg_a, bx_a, _, g_a, (calls_a, bcalls_a) = _run(model, dev, samples, tok, w, True, True, True)
    lg_b, bx_b, _, g_b, (calls_b, bcalls_b) = _run(model, dev, samples, tok, w, False, True)
    assert len(calls_a) == 1 and len(bcalls_a) == 1 and not calls_b and not bcalls_b
    assert float((lg_a - lg_b).norm() / lg_b.norm()) < 1.5e-2
    assert float((bx_a - bx_b).abs().max()) < 2e-2
    worst = {}
    for n in g_b:
        if g_b[n].norm() == 0 or "decoder.layers.0.self_attn.in_proj" in n or n.endswith("attention.self.key.bias"):
            continue
        cos = float(torch.nn.functional.cosine_similarity(g_a[n].flatten(), g_b[n].flatten(), dim=0))
        ratio = float(g_a[n].norm() / g_b[n].norm())
        if cos < 0.995 or not (0.97 < ratio < 1.03):
            worst[n] = (round(cos, 5), round(ratio, 4))
    assert not worst, f"{len(worst)} gradients differ: {dict(list(worst.items())[:12])}"


def test_other_dim_feedforward_takes_the_per_op_path(dev):
    """The launches are compiled for dim_feedforward = 2048 (csrc/xdec.hip XFF); `--dim_feedforward` is a reference option
    (/root/reference/main.py, models/transformer.py:654).  Any other width must run on the per-op launches (it used to read
    linear1 / linear2 out of bounds), and the C entry point refuses a descriptor that says so."""
    from toist_amd import harness
    from toist_amd import kernels as k
    from toist_amd import _lib
    if not k.xdec_supported(2, 100, 46, 6):
        pytest.skip("device without 8 XCDs x 32 CUs")
    assert not k.xdec_supported(2, 100, 46, 6, ff=1024)
    model = _model(dev, dim_feedforward=1024)
    samples, tok, _, _ = harness.synthetic_batch(2, 160, 192, tokens=16, seed=8, max_targets=4)
    g = torch.Generator().manual_seed(3)
    w = [torch.randn(6, 2, 100, n, generator=g).to(dev) for n in (256, 4, 64)]
    lg, bx, loss, grads, (calls, bcalls) = _run(model, dev, samples, tok, w, True, True, True)
    assert not calls and not bcalls
    assert torch.isfinite(lg).all() and all(torch.isfinite(v).all() for v in grads.values())
    assert grads["transformer.decoder.layers.3.linear1.weight"].shape == (1024, 256)
    d = _lib.Xdec()
    d.B, d.Q, d.S, d.L, d.ff = 2, 100, 46, 6, 1024
    assert _lib.lib().toist_xdec_fwd(__import__("ctypes").byref(d), None) != 0
    assert "dim_feedforward" in _lib.last_error()


def test_expired_spin_is_loud_and_falls_back_to_the_per_op_launches(dev):
    """ADVICE r5 / VERDICT r5 weak #13: the launches assume 32 co-resident workgroups per XCD.  `XDEC_TEST_ABSENT = 1` makes one workgroup per XCD leave
    at once, so every group's bounded spins expire exactly as they would beside a second process.  Required behaviour: (a) TRAINING: the outputs of that
    launch are NaN (the reference's finite-loss guard trips without anyone reading a status word), `xdec_check()` raises, sets XDEC_FAILED, and the
    next step runs on the per-op launches; (b) INFERENCE (eval + no_grad): the decode notices by itself, warns, repeats the decoder on the per-op
    launches and returns their (finite) result."""
    import warnings
    from toist_amd import harness
    from toist_amd import kernels as k
    if not k.xdec_supported(2, 100, 46, 6):
        pytest.skip("device without 8 XCDs x 32 CUs")
    model = _model(dev)
    samples, tok, _, _ = harness.synthetic_batch(2, 160, 192, tokens=16, seed=9, max_targets=4)
    g = torch.Generator().manual_seed(3)
    w = [torch.randn(6, 2, 100, n, generator=g).to(dev) for n in (256, 4, 64)]
    lg_ref, bx_ref, _, _, _ = _run(model, dev, samples, tok, w, False, True)
    try:
        k.XDEC_TEST_ABSENT = 1
        with pytest.raises(RuntimeError, match="not co-resident"):
            _run(model, dev, samples, tok, w, True, True, True)          # (_run ends with xdec_check())
        assert k.XDEC_FAILED and not k.xdec_supported(2, 100, 46, 6)
        k.XDEC_FAILED = False
        # the same failing step, looked at the way a training loop would: NaN everywhere it matters
        from toist_amd import tlayer
        tlayer_state = (tlayer.XDEC, tlayer.XDEC_BWD)
        tlayer.XDEC, tlayer.XDEC_BWD = True, True
        try:
            model.train()
            model.zero_grad(set_to_none=True)
            mc = model(samples.to(dev), tok.to(dev), encode_and_save=True)
            out = model(samples.to(dev), tok.to(dev), encode_and_save=False, memory_cache=mc)
            assert torch.isnan(out["pred_logits"]).all()          # (the box head's ReLUs are IEEE maxnum: they turn NaN into 0, the class head is linear)
            assert all(torch.isnan(a["pred_logits"]).all() for a in out["aux_outputs"])
            assert k.xdec_check(raise_on_failure=False) and k.XDEC_FAILED
            # (a) the next step: per-op launches, finite, equal to the reference run
            k.XDEC_TEST_ABSENT = 0
            lg, bx, _, grads, (calls, bcalls) = _run(model, dev, samples, tok, w, True, True, True)
            assert not calls and not bcalls and torch.isfinite(lg).all()
            assert float((lg - lg_ref).norm() / lg_ref.norm()) < 1e-3
            # (b) inference: notices, warns, repeats
            k.XDEC_FAILED, k.XDEC_TEST_ABSENT = False, 1
            tlayer.XDEC = True
            model.eval()
            with torch.no_grad(), warnings.catch_warnings(record=True) as seen:
                warnings.simplefilter("always")
                mc = model(samples.to(dev), tok.to(dev), encode_and_save=True)
                out = model(samples.to(dev), tok.to(dev), encode_and_save=False, memory_cache=mc)
            assert any("per-op" in str(m.message) for m in seen), [str(m.message) for m in seen]
            assert torch.isfinite(out["pred_logits"]).all() and k.XDEC_FAILED
        finally:
            tlayer.XDEC, tlayer.XDEC_BWD = tlayer_state
    finally:
        k.XDEC_TEST_ABSENT = 0
        k.XDEC_FAILED = False
        torch.cuda.synchronize()
        k.xdec_check(raise_on_failure=False)
        k.XDEC_FAILED = False


def test_xcd_resident_launch_beside_a_saturating_second_stream(dev):
    """the launches assume the 32 workgroups of an image are co-resident.  A second stream of the SAME process that saturates HBM (large device
    copies: workgroups without LDS, so they share CUs with the 147 KB decoder workgroups) must not break that: either the launch succeeds (the XCD barriers just
    take longer: 0.84 -> 1.6 us in profiles/r05_xcd_barrier.txt) or the inference path notices an expired spin and repeats the decoder on the per-op launches --
    in both cases the result equals the undisturbed per-op forward."""
    import warnings
    from toist_amd import harness
    from toist_amd import kernels as k
    from toist_amd import tlayer
    if not k.xdec_supported(8, 100, 136, 6):
        pytest.skip("device without 8 XCDs x 32 CUs")
    model = _model(dev).eval()
    samples, tok, _, _ = harness.synthetic_batch(8, 256, 480, tokens=16, seed=12, max_targets=4)
    s_dev, t_dev = samples.to(dev), tok.to(dev)

    def forward(fused):
        old = tlayer.XDEC
        tlayer.XDEC = fused
        try:
            with torch.no_grad():
                mc = model(s_dev, t_dev, encode_and_save=True)
                return model(s_dev, t_dev, encode_and_save=False, memory_cache=mc)["pred_logits"].float().clone()
        finally:
            tlayer.XDEC = old
    ref = forward(False)
    torch.cuda.synchronize()
    big = torch.empty(1 << 27, dtype=torch.float32, device=dev)            # 512 MB: one copy = 1 GB of HBM traffic
    dst = torch.empty_like(big)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    launches = k.XDEC_LAUNCHES
    with torch.cuda.stream(side):
        for _ in range(60):                                                  # ~ 15 ms of saturated HBM beside the forward pass
            dst.copy_(big, non_blocking=True)
    try:
        with warnings.catch_warnings(record=True) as seen:
            warnings.simplefilter("always")
            got = forward(True)
        torch.cuda.synchronize()
        assert k.XDEC_LAUNCHES > launches                                    # the fused launch was issued beside the copies
        fell_back = any("per-op" in str(m.message) for m in seen)
        assert fell_back == k.XDEC_FAILED
        assert torch.isfinite(got).all()
        assert float((got - ref).norm() / ref.norm()) < 1.5e-2, float((got - ref).norm() / ref.norm())
    finally:
        k.XDEC_FAILED = False
        torch.cuda.synchronize()
        k.xdec_check(raise_on_failure=False)
        k.XDEC_FAILED = False
