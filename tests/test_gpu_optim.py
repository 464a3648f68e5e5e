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
