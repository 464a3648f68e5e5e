"""Optimizer-tail oracle (oracle/optim_ref.py) against the reference-generated golden vectors and against
torch's own clip_grad_norm_ / AdamW (CPU only)."""
import os
import types

import numpy as np
import torch

from oracle import optim_ref

GOLD = os.path.join(os.path.dirname(__file__), "golden", "optim_tail.npz")
NAMES = ["head_weight", "head_bias", "backbone_conv_weight", "text_encoder_emb_weight", "text_encoder_norm_bias", "frozen_weight"]


def group_of(name):
    return 1 if "backbone" in name else 2 if "text_encoder" in name else 0


def replay_golden(step_fn):
    """Drives `step_fn(params, grads, lrs, step) -> None` through the fixture; returns the fixture."""
    z = np.load(GOLD)
    args = types.SimpleNamespace(lr=1e-2, lr_backbone=1e-3, text_encoder_lr=5e-3, weight_decay=1e-2, schedule="linear_with_warmup",
                                 fraction_warmup_steps=0.5, lr_drop=10, epochs=20)
    steps = int(z["steps"])
    lrs = [args.lr, args.lr_backbone, args.text_encoder_lr]
    for s in range(steps):
        np.testing.assert_allclose(lrs, z["lrs"][s], rtol=1e-12)
        grads = {n: (torch.from_numpy(z[f"g{s}." + n]) if f"g{s}." + n in z.files else None) for n in NAMES}
        step_fn(grads, lrs, s + 1)
        lrs = optim_ref.learning_rates(args, 0, s, steps)  # engine.py:93-99: the schedule is advanced AFTER the step
    return z


def test_oracle_matches_reference_golden():
    z = np.load(GOLD)
    params = [torch.from_numpy(z["p0." + n]) for n in NAMES]
    state = {"p": params, "m": [torch.zeros_like(p) for p in params], "v": [torch.zeros_like(p) for p in params],
             "e": [p.clone() for p in params]}
    buf_ema = [torch.from_numpy(z["ema.running_stat"]).clone()]

    def step(grads, lrs, t):
        groups = [(lr, 1e-2) for lr in lrs]
        p, m, v, e, _ = optim_ref.tail_step(state["p"], [grads[n] for n in NAMES], state["m"], state["v"], [group_of(n) for n in NAMES],
                                            groups, t, 0.1, emas=state["e"], ema_decay=0.9)
        state.update(p=p, m=m, v=v, e=e)

    replay_golden(step)
    for i, n in enumerate(NAMES):
        np.testing.assert_allclose(state["p"][i].numpy(), z["p." + n], rtol=2e-6, atol=1e-7, err_msg=n)
        np.testing.assert_allclose(state["e"][i].numpy(), z["ema." + n], rtol=2e-6, atol=1e-7, err_msg="ema " + n)
    # a constant buffer is a fixed point of the average (up to rounding)
    np.testing.assert_allclose(buf_ema[0].numpy(), z["ema.running_stat"], rtol=1e-6)
    assert not np.allclose(z["p.head_weight"], z["p0.head_weight"]) and np.array_equal(z["p.frozen_weight"], z["p0.frozen_weight"])


def test_oracle_matches_torch_adamw_and_clip():
    g = torch.Generator().manual_seed(3)
    shapes = [(33, 17), (5,), (4, 3, 3, 3), (1000,)]
    ps = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in shapes]
    opt = torch.optim.AdamW([{"params": ps[:2], "lr": 3e-3}, {"params": ps[2:], "lr": 1e-4, "weight_decay": 0.05}], weight_decay=1e-4,
                            foreach=False)
    ref = [p.detach().clone() for p in ps]
    m = [torch.zeros_like(p) for p in ps]
    v = [torch.zeros_like(p) for p in ps]
    for t in range(1, 6):
        grads = [torch.randn(s, generator=g) * (10.0 if t % 2 else 0.001) for s in shapes]
        for p, gr in zip(ps, grads):
            p.grad = gr.clone()
        norm_t = torch.nn.utils.clip_grad_norm_(ps, 0.1)
        opt.step()
        ref, m, v, _, norm = optim_ref.tail_step(ref, grads, m, v, [0, 0, 1, 1], [(3e-3, 1e-4), (1e-4, 0.05)], t, 0.1)
        assert abs(float(norm) - float(norm_t)) <= 1e-5 * float(norm_t)
        for a, b in zip(ref, ps):
            torch.testing.assert_close(a, b.detach(), rtol=2e-6, atol=1e-7)


def test_clip_disabled_when_max_norm_not_positive():
    g = [torch.full((4,), 100.0)]
    norm, coef = optim_ref.clip_coef(g, 0.0)
    assert float(coef) == 1.0 and abs(float(norm) - 200.0) < 1e-3
