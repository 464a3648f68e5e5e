import math, sys, os
sys.path.insert(0, os.getcwd())
import torch
import toist_amd
from toist_amd import harness, kernels, engine
from toist_amd.optim import FusedClipAdamWEMA

def run(lse_only):
    engine.LSE_ONLY = lse_only
    dev = torch.device("cuda:0")
    args = harness.default_args(device="cuda", enc_layers=1, dec_layers=2, num_queries=20)
    torch.manual_seed(0)
    model, criterion, _, weight_dict = toist_amd.build_model(args)
    model.to(dev).train(); criterion.train()
    samples, tok, targets, pmap = harness.synthetic_batch(2, 128, 160, tokens=12, seed=11, device=dev, max_targets=4)
    kernels.SEED_DEV = torch.zeros(1, dtype=torch.int64, device=dev)
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    opt = FusedClipAdamWEMA([{"params": [p for n, p in named if "backbone" not in n and "text_encoder" not in n], "lr": 1e-4},
                             {"params": [p for n, p in named if "backbone" in n], "lr": 1e-5},
                             {"params": [p for n, p in named if "text_encoder" in n], "lr": 5e-5}], weight_decay=1e-4, max_norm=0.1)
    hist = []
    for it in range(12):
        kernels.SEED_DEV.add_(1000003)
        opt.zero_grad(set_to_none=True)
        mc = model(samples, tok, encode_and_save=True)
        out = model(samples, tok, encode_and_save=False, memory_cache=mc)
        criterion._pending_status = []
        losses = criterion(mc, out, targets, pmap, None)
        total = toist_amd.weighted_total(losses, weight_dict)
        total.backward()
        bad = [n for n, p in named if p.grad is not None and not bool(torch.isfinite(p.grad).all())]
        st = opt.device_state() if it else None
        opt.step()
        hist.append(float(total.detach()))
        print(lse_only, it, hist[-1], "bad grads:", bad[:6], len(bad), "gradnorm", opt.device_state()["grad_norm"], flush=True)
        if bad:
            break

run(False)
run(True)
