"""Golden vectors of the optimizer tail: the REAL reference's update_ema / adjust_learning_rate
(/root/reference/util/optim.py) driving torch.optim.AdamW and clip_grad_norm_ exactly as
/root/reference/engine.py:87-101 does, on closed-form parameters and gradients.  Runs only in the build
container.  Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_optim.py
"""
import copy
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import formula  # noqa: E402

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
from util.optim import adjust_learning_rate, update_ema  # noqa: E402

SHAPES = {"head.weight": (7, 5), "head.bias": (7,), "backbone.conv.weight": (6, 4, 3, 3), "text_encoder.emb.weight": (11, 8),
          "text_encoder.norm.bias": (8,), "frozen.weight": (3, 3)}
STEPS = 4


class Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        for name, shape in SHAPES.items():
            self.register_parameter(name.replace(".", "_"), torch.nn.Parameter(formula.tensor("optim.p." + name, shape, 0.8)))
        self.frozen_weight.requires_grad_(False)
        self.register_buffer("running_stat", formula.tensor("optim.buf", (5,), 1.0, 2.0))


def main():
    args = types.SimpleNamespace(lr=1e-2, lr_backbone=1e-3, text_encoder_lr=5e-3, weight_decay=1e-2, schedule="linear_with_warmup",
                                 fraction_warmup_steps=0.5, lr_drop=10, epochs=20, ema_decay=0.9)
    model = Net()
    model_ema = copy.deepcopy(model)
    named = list(model.named_parameters())
    groups = [  # main.py:351-366
        {"params": [p for n, p in named if "backbone" not in n and "text_encoder" not in n and p.requires_grad]},
        {"params": [p for n, p in named if "backbone" in n and p.requires_grad], "lr": args.lr_backbone},
        {"params": [p for n, p in named if "text_encoder" in n and p.requires_grad], "lr": args.text_encoder_lr},
    ]
    opt = torch.optim.AdamW(groups, lr=args.lr, weight_decay=args.weight_decay)
    out = {"steps": STEPS}
    for n, p in named:
        out["p0." + n] = p.detach().clone()
    lrs = []
    for step in range(STEPS):
        opt.zero_grad()
        for n, p in named:
            if p.requires_grad:
                p.grad = formula.tensor(f"optim.g{step}." + n, tuple(p.shape), 3.0 if step % 2 else 0.02)  # clipped and unclipped steps
                out[f"g{step}." + n] = p.grad.clone()
        lrs.append([g["lr"] for g in opt.param_groups])
        torch.nn.utils.clip_grad_norm_(model.parameters(), 0.1)          # engine.py:89-90
        opt.step()                                                        # engine.py:91
        adjust_learning_rate(opt, 0, step, num_training_steps=STEPS, args=args)   # engine.py:93-99
        update_ema(model, model_ema, args.ema_decay)                      # engine.py:100-101
    out["lrs"] = np.asarray(lrs, dtype=np.float64)
    for n, p in named:
        out["p." + n] = p.detach().clone()
    for n, v in model_ema.state_dict().items():
        out["ema." + n] = v.clone()
    np.savez_compressed(os.path.join(HERE, "optim_tail.npz"), **{k: (v.numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in out.items()})
    print("wrote optim_tail.npz", len(out), "arrays; lrs", lrs)


if __name__ == "__main__":
    main()
