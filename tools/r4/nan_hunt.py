import sys
import torch
sys.path.insert(0, ".")
import toist_amd
from toist_amd import harness, kernels, tlayer
from toist_amd.optim import FusedClipAdamWEMA

dev = torch.device("cuda")
small = "small" in sys.argv
if "norows" in sys.argv:
    tlayer.ENABLED = False
if "noattn2" in sys.argv:
    tlayer.ATTN2 = False
args = harness.default_args(device="cuda", enc_layers=1, dec_layers=2, num_queries=20, dropout=0.0) if small else harness.default_args(device="cuda", contrastive_align_loss=True)
torch.manual_seed(0)
model, criterion, _, weight_dict = toist_amd.build_model(args)
model.to(dev).train()
B, H_, W_ = (2, 128, 160) if small else (8, 640, 640)
samples, tok, targets, pmap = harness.synthetic_batch(B, H_, W_, tokens=12 if small else 16, seed=5, device=dev, max_targets=4)
kernels.SEED_DEV = torch.zeros(1, dtype=torch.int64, device=dev)
named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
opt = FusedClipAdamWEMA([{"params": [p for n, p in named if "backbone" not in n and "text_encoder" not in n], "lr": 1e-4},
                         {"params": [p for n, p in named if "backbone" in n], "lr": 1e-5},
                         {"params": [p for n, p in named if "text_encoder" in n], "lr": 5e-5}], weight_decay=1e-4, max_norm=0.1)
for it in range(3):
    opt.zero_grad(set_to_none=True)
    mc = model(samples, tok, encode_and_save=True)
    nat = mc["_native"]
    print(it, "src_proj finite", bool(torch.isfinite(nat["src_proj"]).all()), float(nat["src_proj"].float().abs().max()), "text", bool(torch.isfinite(mc["text_memory_resized"]).all()),
          "C5", bool(torch.isfinite(nat["features"][-1]).all()), float(nat["features"][-1].float().abs().max()))
    print(it, "img_memory finite", bool(torch.isfinite(mc["img_memory"]).all()), float(mc["img_memory"].abs().max()))
    out = model(samples, tok, encode_and_save=False, memory_cache=mc)
    st = out["_stacked"]
    print(it, "logits finite", bool(torch.isfinite(st["pred_logits"]).all()), "boxes", bool(torch.isfinite(st["pred_boxes"]).all()), float(st["pred_logits"].abs().max()))
    losses = criterion(mc, out, targets, pmap, None)
    total = sum(losses[k_] * weight_dict[k_] for k_ in losses if k_ in weight_dict)
    print(it, "loss", float(total))
    total.backward()
    torch.cuda.synchronize()
    bad = [n for n, p in model.named_parameters() if p.grad is not None and not bool(torch.isfinite(p.grad).all())]
    print(it, "non-finite grads:", len(bad), bad[:8])
    big = sorted(((float(p.grad.abs().max()), n) for n, p in model.named_parameters() if p.grad is not None), reverse=True)[:4]
    print(it, "largest grads", big)
    if "noopt" not in sys.argv:
        opt.step()
    torch.cuda.synchronize()
    print(it, opt.device_state())
    badp = [n for n, p in model.named_parameters() if not bool(torch.isfinite(p).all())]
    print(it, "non-finite params:", len(badp), badp[:8])
