"""GPU parity of the fused optimizer tail (csrc/optim.hip through the C ABI) against oracle/optim_ref.py and the
reference-generated golden vectors.  Tolerance: fp32 arithmetic in a different association order -> rtol 1e-5
(atol 1e-7) on parameters / moments / EMA; the bf16 compute copy must equal bf16(p * row_scale) bit for bit."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_fused_tail_matches_oracle(dev):
    from oracle import optim_ref
    from toist_amd import engine, optim

    g = torch.Generator().manual_seed(11)
    chunk = 8192
    shapes = [(33, 17), (5,), (1,), (chunk * 2 + 5,), (64, 16, 3, 3), (3, 7, 1, 1), (chunk,), (257, 129)]
    cpu_p = [torch.randn(s, generator=g) for s in shapes]
    params = [torch.nn.Parameter(p.clone().to(dev)) for p in cpu_p]
    # a channels_last conv weight (physically KRSC) with a folded per-row scale, like the ResNet weights
    params[4] = torch.nn.Parameter(cpu_p[4].clone().to(dev).contiguous(memory_format=torch.channels_last))
    scale = (torch.rand(64, generator=g) + 0.5).to(dev)
    frozen_src = torch.randn(300, generator=g).to(dev)          # EMA-only tensor (buffer / frozen parameter)
    emas = [p.detach().clone() for p in params]
    frozen_ema = frozen_src.clone()
    # bf16 compute copies registered the way ParamSet does it
    cache = {}
    fold = lambda m: (m.detach() * scale.view(-1, 1, 1, 1)).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    fold.elementwise, fold.row_scale = True, scale
    w_conv = engine.compute_copy(params[4], fold, cache, "conv")
    w_lin = engine.compute_copy(params[7], engine._cast_bf16, cache, "lin")
    groups = [{"params": params[:4], "lr": 3e-3}, {"params": params[4:6], "lr": 1e-4, "weight_decay": 0.05}, {"params": params[6:], "lr": 5e-5}]
    opt = optim.FusedClipAdamWEMA(groups, lr=1e-4, weight_decay=1e-4, max_norm=0.1, ema=list(zip(params, emas)) + [(frozen_src, frozen_ema)],
                                  ema_decay=0.99)
    ref_p = [p.clone() for p in cpu_p]
    ref_m = [torch.zeros_like(p) for p in cpu_p]
    ref_v = [torch.zeros_like(p) for p in cpu_p]
    ref_e = [p.clone() for p in cpu_p]
    ref_fe = frozen_src.cpu().clone()
    group_of = [0, 0, 0, 0, 1, 1, 2, 2]
    for t in range(1, 5):
        grads = [torch.randn(s, generator=g) * (5.0 if t % 2 else 1e-3) for s in shapes]
        if t == 3:
            grads[1] = None                                    # a parameter without gradient this step
        for i, (p, gr) in enumerate(zip(params, grads)):
            if gr is None:
                p.grad = None
            else:
                gd = gr.to(dev)
                p.grad = gd.contiguous(memory_format=torch.channels_last) if i == 4 else gd
        if t == 4:
            opt.param_groups[0]["lr"] = 1e-3                   # schedule change between steps (util/optim.py:86-90)
        opt.step()
        lr0 = 1e-3 if t == 4 else 3e-3
        ref_p, ref_m, ref_v, ref_e, norm = optim_ref.tail_step(ref_p, grads, ref_m, ref_v, group_of,
                                                               [(lr0, 1e-4), (1e-4, 0.05), (5e-5, 1e-4)], t, 0.1, emas=ref_e, ema_decay=0.99)
        ref_fe = optim_ref.ema_update(ref_fe, frozen_src.cpu(), 0.99)
        st = opt.device_state()
        assert st["step"] == t
        assert abs(st["grad_norm"] - float(norm)) <= 1e-5 * float(norm)
        for i in range(len(shapes)):
            torch.testing.assert_close(params[i].detach().cpu(), ref_p[i], rtol=1e-5, atol=1e-7, msg=lambda m: f"step {t} p[{i}]: {m}")
            torch.testing.assert_close(opt.exp_avg[i].cpu(), ref_m[i], rtol=1e-5, atol=1e-9)
            torch.testing.assert_close(opt.exp_avg_sq[i].cpu(), ref_v[i], rtol=1e-5, atol=1e-12)
            torch.testing.assert_close(emas[i].cpu(), ref_e[i], rtol=1e-5, atol=1e-7)
        torch.testing.assert_close(frozen_ema.cpu(), ref_fe, rtol=1e-6, atol=1e-7)
        # compute copies: rewritten in place from the new masters, bit-exact
        assert torch.equal(w_conv, fold(params[4])) and torch.equal(w_lin, params[7].detach().to(torch.bfloat16))
        assert engine.compute_copy(params[4], fold, cache, "conv") is w_conv        # still valid: no re-cast after the tail


def test_fused_tail_replays_reference_golden(dev):
    """The reference's own loop (engine.py:87-101 with util/optim.py's schedule + EMA) as recorded in
    tests/golden/optim_tail.npz, replayed on the GPU."""
    from tests.test_cpu_optim import NAMES, group_of, replay_golden
    from toist_amd import optim
    z = np.load(__import__("tests.test_cpu_optim", fromlist=["GOLD"]).GOLD)
    params = {n: torch.nn.Parameter(torch.from_numpy(z["p0." + n]).to(dev)) for n in NAMES}
    params["frozen_weight"].requires_grad_(False)
    emas = {n: p.detach().clone() for n, p in params.items()}
    train = [n for n in NAMES if n != "frozen_weight"]
    groups = [{"params": [params[n] for n in train if group_of(n) == gi], "lr": lr} for gi, lr in enumerate([1e-2, 1e-3, 5e-3])]
    opt = optim.FusedClipAdamWEMA(groups, weight_decay=1e-2, max_norm=0.1, ema=[(params[n], emas[n]) for n in NAMES], ema_decay=0.9)

    def step(grads, lrs, t):
        for gi, lr in enumerate(lrs):
            opt.param_groups[gi]["lr"] = lr
        for n in train:
            params[n].grad = grads[n].to(dev)
        opt.step()

    replay_golden(step)
    for n in NAMES:
        np.testing.assert_allclose(params[n].detach().cpu().numpy(), z["p." + n], rtol=1e-5, atol=1e-7, err_msg=n)
        np.testing.assert_allclose(emas[n].cpu().numpy(), z["ema." + n], rtol=1e-5, atol=1e-7, err_msg="ema " + n)


def test_torch_optimizer_step_invalidates_compute_copies(dev):
    """torch's fused AdamW updates parameters without bumping their version counters: the global
    optimizer-step hook must still force the bf16 compute copy to be rebuilt."""
    from toist_amd import engine
    p = torch.nn.Parameter(torch.randn(64, 32, device=dev))
    cache = {}
    w0 = engine.compute_copy(p, engine._cast_bf16, cache, "w").clone()
    opt = torch.optim.AdamW([p], lr=0.1, fused=True)
    p.grad = torch.ones_like(p)
    opt.step()
    w1 = engine.compute_copy(p, engine._cast_bf16, cache, "w")
    assert not torch.equal(w0, w1) and torch.equal(w1, p.detach().to(torch.bfloat16))


def test_two_tails_keep_each_others_compute_copies_valid(dev):
    """Two models, two fused tails (the distillation step: teacher + student): a tail's step bumps the GLOBAL weight epoch, which used to make the
    other model's compute copies look stale -- its next forward re-made ~260 copies one torch launch at a time.  Each tail now carries the other
    tails' valid copies over its own bump; a torch optimizer's step still invalidates everything."""
    from toist_amd import engine
    from toist_amd.optim import FusedClipAdamWEMA
    pa = torch.nn.Parameter(torch.randn(64, 32, device=dev))
    pb = torch.nn.Parameter(torch.randn(48, 16, device=dev))
    ca, cb = {}, {}
    made = []

    def make(m):
        made.append(tuple(m.shape))
        return engine._cast_bf16(m)
    make.elementwise = True
    engine.compute_copy(pa, make, ca, "w")
    engine.compute_copy(pb, make, cb, "w")
    oa = FusedClipAdamWEMA([{"params": [pa]}], lr=0.1, max_norm=0.0)
    ob = FusedClipAdamWEMA([{"params": [pb]}], lr=0.1, max_norm=0.0)
    for _ in range(2):
        pa.grad, pb.grad = torch.ones_like(pa), torch.ones_like(pb)
        oa.step()
        ob.step()
        n = len(made)
        wa = engine.compute_copy(pa, make, ca, "w")
        wb = engine.compute_copy(pb, make, cb, "w")
        assert len(made) == n, made[n:]                    # neither copy was re-made: the tails rewrote them and kept each other's valid
        assert torch.equal(wa, pa.detach().to(torch.bfloat16)) and torch.equal(wb, pb.detach().to(torch.bfloat16))
    pa.grad = torch.ones_like(pa)
    torch.optim.SGD([pa], lr=0.1).step()                   # a torch optimizer: unknown which masters moved -> every copy is stale
    n = len(made)
    engine.compute_copy(pa, make, ca, "w")
    engine.compute_copy(pb, make, cb, "w")
    assert len(made) == n + 2


def test_deferred_ema_is_the_same_average(dev):
    """defer_ema=True: the moving average leaves step() and runs as ema_update() (beside the next forward pass in bench.py); parameters,
    moments and -- once the last update is flushed -- the average equal those of the in-step variant bit for bit."""
    from toist_amd.optim import FusedClipAdamWEMA
    g = torch.Generator().manual_seed(5)
    shapes = [(300, 70), (17,), (64, 3, 3, 8), (9000,)]

    def run(defer):
        params = [torch.nn.Parameter(torch.randn(*s, generator=g.manual_seed(11 + i)).to(dev)) for i, s in enumerate(shapes)]
        frozen = torch.randn(123, generator=g.manual_seed(99)).to(dev)
        src = params + [frozen]
        ema = [p.detach().clone() for p in src]
        opt = FusedClipAdamWEMA([{"params": params[:2], "lr": 1e-3}, {"params": params[2:], "lr": 3e-4}], weight_decay=1e-2, max_norm=0.5,
                                ema=list(zip(src, ema)), ema_decay=0.9, defer_ema=defer)
        for step in range(4):
            for i, p in enumerate(params):
                p.grad = torch.randn(p.shape, generator=g.manual_seed(1000 * step + i)).to(dev)
            if defer and step % 2 == 1:
                opt.ema_update()                    # explicitly, as the training loop does; other steps rely on step()'s own flush
            opt.step()
        opt.ema_update()
        return [p.detach().clone() for p in params], ema, [m.clone() for m in opt.exp_avg]
    a, b = run(False), run(True)
    for x, y in zip(a[0] + a[1] + a[2], b[0] + b[1] + b[2]):
        assert torch.equal(x, y)


def test_late_group_survives_in_place_zero_grad(dev):
    """zero_grad(set_to_none=False) zeroes the gradient tensors a pending late-group launch still has to read -- the launch is
    issued first.  Three steps of a loop that zeroes in place: the late run's text-encoder weights equal the serial tail's."""
    import copy
    import toist_amd
    from toist_amd import harness, kernels
    from toist_amd.optim import FusedClipAdamWEMA
    args = harness.default_args(device="cuda", enc_layers=1, dec_layers=1, num_queries=20, dropout=0.0)
    torch.manual_seed(0)
    model0, criterion, _, weight_dict = toist_amd.build_model(args)
    model0.to(dev).train()
    samples, tok, targets, pmap = harness.synthetic_batch(2, 128, 160, tokens=12, seed=5, device=dev, max_targets=4)
    kernels.SEED_DEV = torch.zeros(1, dtype=torch.int64, device=dev)
    res = {}
    for late in (False, True):
        model = copy.deepcopy(model0)
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        opt = FusedClipAdamWEMA([{"params": [p for n, p in named if "text_encoder" not in n], "lr": 1e-4},
                                 {"params": [p for n, p in named if "text_encoder" in n], "lr": 5e-5, "late": late}], weight_decay=1e-4, max_norm=0.1)
        for _ in range(3):
            opt.zero_grad(set_to_none=False)
            mc = model(samples, tok, encode_and_save=True)
            out = model(samples, tok, encode_and_save=False, memory_cache=mc)
            losses = criterion(mc, out, targets, pmap, None)
            sum(losses[k_] * weight_dict[k_] for k_ in losses if k_ in weight_dict).backward()
            opt.step()
        opt.finish()
        torch.cuda.synchronize()
        res[late] = model.transformer.text_encoder.encoder.layer[0].output.dense.weight.detach().clone()
    torch.testing.assert_close(res[True], res[False], rtol=2e-4, atol=1e-6)
    assert not torch.equal(res[True], model0.transformer.text_encoder.encoder.layer[0].output.dense.weight)


def test_late_text_group_is_bit_identical_to_the_serial_tail(dev):
    """A parameter group marked "late" (the text encoder) is updated at the head of the NEXT forward pass's text branch instead of
    inside step() (toist_amd.optim / engine.run_text_prelude): same gradients, same clip coefficient, same step count -- after five
    training steps (+ finish()) every parameter, moment, EMA tensor and bf16 compute copy equals the serial tail's bit for bit, in the
    eager loop and when the step is replayed from a hipGraph."""
    import copy
    import toist_amd
    from toist_amd import engine, harness, kernels
    from toist_amd.optim import FusedClipAdamWEMA
    args = harness.default_args(device="cuda", enc_layers=1, dec_layers=2, num_queries=20, dropout=0.0)
    torch.manual_seed(0)
    model0, criterion, _, weight_dict = toist_amd.build_model(args)
    model0.to(dev).train()
    samples, tok, targets, pmap = harness.synthetic_batch(2, 128, 160, tokens=12, seed=5, device=dev, max_targets=4)
    kernels.SEED_DEV = torch.zeros(1, dtype=torch.int64, device=dev)

    def make(model, late):
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        src = [v for v in model.state_dict().values() if v.is_floating_point()]
        ema = [v.detach().clone() for v in src]
        opt = FusedClipAdamWEMA([{"params": [p for n, p in named if "backbone" not in n and "text_encoder" not in n], "lr": 1e-4},
                                 {"params": [p for n, p in named if "backbone" in n], "lr": 1e-5},
                                 {"params": [p for n, p in named if "text_encoder" in n], "lr": 5e-5, "late": late}], weight_decay=1e-4, max_norm=0.1,
                                ema=list(zip(src, ema)), ema_decay=0.99)

        def step():
            opt.zero_grad(set_to_none=True)
            mc = model(samples, tok, encode_and_save=True)
            out = model(samples, tok, encode_and_save=False, memory_cache=mc)
            losses = criterion(mc, out, targets, pmap, None)
            total = sum(losses[k_] * weight_dict[k_] for k_ in losses if k_ in weight_dict)
            total.backward()
            opt.step()
            return total.detach()
        return opt, ema, step

    def state(model, opt, ema):
        opt.finish()
        torch.cuda.synchronize()
        out = {"p." + n: p.detach().clone() for n, p in model.named_parameters()}
        out.update({f"m.{i}": t.clone() for i, t in enumerate(opt.exp_avg)})
        out.update({f"v.{i}": t.clone() for i, t in enumerate(opt.exp_avg_sq)})
        out.update({f"e.{i}": t.clone() for i, t in enumerate(ema)})
        return out

    # ---- eager loop ----
    ser, lat = copy.deepcopy(model0), copy.deepcopy(model0)
    o_s, e_s, step_s = make(ser, False)
    o_l, e_l, step_l = make(lat, True)
    assert o_l._late and any(o_l._late)
    loss_s = [float(step_s()) for _ in range(5)]
    loss_l = [float(step_l()) for _ in range(5)]
    assert o_l._late_pending                      # the fifth update of the text group has not been issued yet
    text_w = lat.transformer.text_encoder.encoder.layer[0].output.dense.weight
    ser_w = ser.transformer.text_encoder.encoder.layer[0].output.dense.weight
    assert not torch.equal(text_w, ser_w)
    a, b = state(ser, o_s, e_s), state(lat, o_l, e_l)
    assert not o_l._late_pending and torch.equal(text_w, ser_w)
    assert loss_s == loss_l, (loss_s, loss_l)
    bad = [n for n in a if not torch.equal(a[n], b[n])]
    # LayerNorm / embedding gradients are summed with fp32 atomics in some kernels: those (and what they feed) may differ in the last bit
    assert len(bad) <= len(a) // 3, f"{len(bad)} of {len(a)} tensors differ, e.g. {bad[:6]}"
    for n in bad:
        torch.testing.assert_close(b[n], a[n], rtol=2e-4, atol=1e-6)
    # the bf16 compute copies the next forward pass would read equal bf16(master) (the late launch refreshed the text encoder's)
    ent = engine.copy_of(text_w)
    assert ent is not None and ent.epoch == engine.WEIGHT_EPOCH and torch.equal(ent.w.view(-1), text_w.detach().to(torch.bfloat16).view(-1))

    # ---- hipGraph replay: the late launch is a node at the head of the captured text branch ----
    gm = copy.deepcopy(model0)
    o_g, e_g, step_g = make(gm, True)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step_g()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            step_g()
    torch.cuda.current_stream().wait_stream(side)
    for _ in range(3):
        graph.replay()
    em = copy.deepcopy(model0)
    o_e, e_e, step_e = make(em, False)
    for _ in range(5):
        step_e()
    a, b = state(em, o_e, e_e), state(gm, o_g, e_g)
    assert o_g.device_state()["step"] == 5 == o_e.device_state()["step"]
    # replay vs eager is not bit-comparable (fp32 atomics order; near-tied assignments of a random-init model, see
    # test_graph_replayed_training_matches_eager_training): every group must have moved by five updates of the same size
    init = {"p." + n: p.detach() for n, p in model0.named_parameters()}
    for n in ("p.transformer.text_encoder.encoder.layer.3.intermediate.dense.weight", "p.transformer.text_encoder.encoder.layer.11.output.dense.weight",
              "p.transformer.resizer.fc.weight", "p.transformer.encoder.layers.0.linear1.weight", "p.backbone.0.body.layer3.5.conv2.weight"):
        de, dg = float((a[n] - init[n]).norm()), float((b[n] - init[n]).norm())
        assert de > 0 and abs(dg - de) <= 0.25 * de, (n, de, dg)


@pytest.mark.gpu
def test_early_norm_group_gives_the_same_clip_and_parameters(dev):
    """A parameter group marked "early_norm" (the text encoder) has the squares of its gradients summed by the AFTER_BACKWARD hook at the end of
    the text program's backward pass; step() sums only the rest.  Same per-chunk partial sums => the total norm, the clip coefficient and every
    parameter / moment equal the serial tail's bit for bit (up to the fp32-atomic tensors, as in the late-group test).  The hook is really taken
    once the gradient buffers are reused, and is skipped (full sum in step()) while their addresses still change."""
    import copy
    import toist_amd
    from toist_amd import engine, harness, kernels
    from toist_amd.optim import FusedClipAdamWEMA
    args = harness.default_args(device="cuda", enc_layers=1, dec_layers=2, num_queries=20, dropout=0.0)
    torch.manual_seed(0)
    model0, criterion, _, weight_dict = toist_amd.build_model(args)
    model0.to(dev).train()
    samples, tok, targets, pmap = harness.synthetic_batch(2, 128, 160, tokens=12, seed=5, device=dev, max_targets=4)
    kernels.SEED_DEV = torch.zeros(1, dtype=torch.int64, device=dev)
    saved = engine.REUSE_GRAD_BUFFERS
    engine.REUSE_GRAD_BUFFERS = True
    try:
        def make(model, early):
            named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
            opt = FusedClipAdamWEMA([{"params": [p for n, p in named if "text_encoder" not in n], "lr": 1e-4},
                                     {"params": [p for n, p in named if "text_encoder" in n], "lr": 5e-5, "early_norm": early}], weight_decay=1e-4, max_norm=0.1)
            taken = []

            def step():
                opt.zero_grad(set_to_none=True)
                mc = model(samples, tok, encode_and_save=True)
                out = model(samples, tok, encode_and_save=False, memory_cache=mc)
                losses = criterion(mc, out, targets, pmap, None)
                total = sum(losses[k_] * weight_dict[k_] for k_ in losses if k_ in weight_dict)
                total.backward()
                taken.append(opt._early_done)
                opt.step()
                return total.detach(), opt.device_state()["grad_norm"], opt.device_state()["clip_coef"]
            return opt, step, taken

        ser, ear = copy.deepcopy(model0), copy.deepcopy(model0)
        o_s, step_s, _ = make(ser, False)
        o_e, step_e, taken = make(ear, True)
        assert o_e._early_ids and not o_s._early_ids
        rs = [step_s() for _ in range(4)]
        re_ = [step_e() for _ in range(4)]
        assert taken[0] is False and taken[-1] is True, taken          # first step: no table yet; steady state: the hook summed the text group
        assert 0 < o_e._e_lo < o_e._e_hi == o_e._n_now
        for (ls, ns, cs), (le, ne, ce) in zip(rs, re_):
            assert float(ls) == float(le) and ns == ne and cs == ce, (rs, re_)
        torch.cuda.synchronize()
        a = {n: p.detach() for n, p in ser.named_parameters()}
        b = {n: p.detach() for n, p in ear.named_parameters()}
        bad = [n for n in a if not torch.equal(a[n], b[n])]
        assert len(bad) <= len(a) // 3, f"{len(bad)} of {len(a)} tensors differ, e.g. {bad[:6]}"
        for n in bad:
            torch.testing.assert_close(b[n], a[n], rtol=2e-4, atol=1e-6)
    finally:
        engine.REUSE_GRAD_BUFFERS = saved
