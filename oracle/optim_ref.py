"""Oracle restatement of the optimizer tail (test infrastructure; see oracle/__init__.py).

Follows /root/reference/engine.py:87-101 -- `clip_grad_norm_(model.parameters(), max_norm)`,
`optimizer.step()` (torch.optim.AdamW over the three groups of /root/reference/main.py:351-392),
`adjust_learning_rate` (/root/reference/util/optim.py:29-90) and `update_ema`
(/root/reference/util/optim.py:9-26) -- as plain fp32 tensor arithmetic on the CPU.

AdamW and clip_grad_norm_ live in a third-party dependency (torch==1.10.2, requirements.txt); their
published algorithms are restated here (Loshchilov & Hutter 2019, decoupled weight decay; torch's
`clip_coef = max_norm / (total_norm + 1e-6)` clamped to 1) and pinned by tests/test_cpu_optim.py against
torch's own implementations and against tests/golden/optim_tail.npz, which was produced by the
reference's update_ema / adjust_learning_rate driving torch.optim.AdamW (tests/golden/make_golden_optim.py).
"""
from bisect import bisect_right

import torch


def clip_coef(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_ (norm_type 2): returns (total_norm, coefficient applied to every gradient)."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    if max_norm <= 0:  # engine.py:89 `if max_norm > 0`
        return total, torch.tensor(1.0)
    return total, torch.clamp(max_norm / (total + 1e-6), max=1.0)


def adamw_update(p, g, m, v, step, lr, weight_decay, beta1=0.9, beta2=0.999, eps=1e-8):
    """One torch.optim.AdamW update (amsgrad off) -> (p, m, v); `step` counts from 1."""
    p = p * (1.0 - lr * weight_decay)
    m = m + (g - m) * (1.0 - beta1)
    v = beta2 * v + (1.0 - beta2) * g * g
    bias1 = 1.0 - beta1 ** step
    bias2 = 1.0 - beta2 ** step
    denom = v.sqrt() / (bias2 ** 0.5) + eps
    p = p - (lr / bias1) * (m / denom)
    return p, m, v


def ema_update(ema, value, decay):
    """util/optim.py:26"""
    return ema * decay + (1.0 - decay) * value


def learning_rates(args, epoch, curr_step, num_training_steps):
    """util/optim.py:29-90 -> [lr, lr_backbone, text_encoder_lr] for this step."""
    warm = round(args.fraction_warmup_steps * num_training_steps)

    def linear():
        if curr_step < warm:
            return float(curr_step) / float(max(1, warm))
        return max(0.0, float(num_training_steps - curr_step) / float(max(1, num_training_steps - warm)))

    if args.schedule == "step":
        gamma = text_gamma = 0.1 ** (epoch // args.lr_drop)
    elif args.schedule == "multistep":
        gamma = text_gamma = 0.5 ** bisect_right(list(range(args.lr_drop, args.epochs, 50)), epoch)
    elif args.schedule == "linear_with_warmup":
        gamma, text_gamma = 0.1 ** (epoch // args.lr_drop), linear()
    elif args.schedule == "all_linear_with_warmup":
        gamma = text_gamma = linear()
    else:
        raise NotImplementedError(args.schedule)
    return [args.lr * gamma, args.lr_backbone * gamma, args.text_encoder_lr * text_gamma]


def tail_step(params, grads, exp_avg, exp_avg_sq, group_of, groups, step, max_norm, emas=None, ema_decay=0.9998,
              betas=(0.9, 0.999), eps=1e-8):
    """One whole tail on lists of tensors.  groups[i] = (lr, weight_decay); emas[i] may be None; gradients that
    are None leave their parameter untouched (it is still averaged).  Returns (params, exp_avg, exp_avg_sq, emas, norm)."""
    norm, coef = clip_coef([g for g in grads if g is not None], max_norm)
    out_p, out_m, out_v, out_e = [], [], [], []
    for i, p in enumerate(params):
        m, v = exp_avg[i], exp_avg_sq[i]
        if grads[i] is not None:
            lr, wd = groups[group_of[i]]
            p, m, v = adamw_update(p, grads[i] * coef, m, v, step, lr, wd, betas[0], betas[1], eps)
        out_p.append(p)
        out_m.append(m)
        out_v.append(v)
        out_e.append(None if emas is None or emas[i] is None else ema_update(emas[i], p, ema_decay))
    return out_p, out_m, out_v, out_e, norm
