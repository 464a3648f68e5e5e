"""Find what breaks the hipGraph capture of the (fixed-batch) distillation step: capture with the synchronisation debug mode set to "error" and print
the traceback of the first failing call."""
import sys
import traceback

import torch

sys.path.insert(0, ".")
import toist_amd  # noqa: E402
from toist_amd import engine, harness, kernels  # noqa: E402
from toist_amd.optim import FusedClipAdamWEMA  # noqa: E402

dev = torch.device("cuda")
FULL = "full" in sys.argv
B = 4 if FULL else 2
small = {} if FULL else dict(enc_layers=1, dec_layers=2, num_queries=20)
args = harness.default_args(device="cuda", distillation=True, cluster=True, nsthl2_loss=True, softkd_loss=True, train_batch_size=B, **small)
torch.manual_seed(0)
model, criterion, cluster_criterion, weight_dict = toist_amd.build_model(args)
model_noun, _, _, _ = toist_amd.build_model(args)
for m in (model, model_noun):
    m.to(dev).train()
cluster_criterion.to(dev)
cluster_criterion.full_label.fill_(1)
engine.REUSE_GRAD_BUFFERS = True
kernels.SEED_DEV = torch.zeros(1, dtype=torch.int64, device=dev)
def tail(m):
    named = [(n, p) for n, p in m.named_parameters() if p.requires_grad]
    groups = [{"params": [p for n, p in named if "backbone" not in n and "text_encoder" not in n]},
              {"params": [p for n, p in named if "backbone" in n], "lr": 1e-5},
              {"params": [p for n, p in named if "text_encoder" in n], "lr": 5e-5}]
    src = [v for v in m.state_dict().values() if v.is_floating_point()]
    ema = list(zip(src, [v.detach().clone() for v in src])) if "ema" in sys.argv else None
    return FusedClipAdamWEMA(groups, lr=1e-4, weight_decay=1e-4, max_norm=0.1, ema=ema, ema_decay=0.9998)


opts = [tail(model), tail(model_noun)]
from toist_amd import parallel  # noqa: E402
import contextlib  # noqa: E402
sync = parallel.GradSync([model, model_noun]) if "sync" in sys.argv else None
batch = harness.synthetic_distill_batch(B, 640, 640, tokens=16, seed=1000, device=dev) if FULL else harness.synthetic_distill_batch(B, 192, 160, tokens=16, seed=1000, device=dev, max_targets=4)
stage = sys.argv[1] if len(sys.argv) > 1 else "all"


def step():
    kernels.SEED_DEV.add_(1000003)
    for o in opts:
        o.zero_grad(set_to_none=True)
    with (sync if sync is not None else contextlib.nullcontext()):
        total, _ = harness.distillation_step(model, model_noun, criterion, cluster_criterion, weight_dict, batch)
        if stage in ("all", "bwd"):
            total.backward()
        if sync is not None:
            sync.finish()
    if stage == "all":
        for o in opts:
            o.step()
    return total


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side), kernels.tables_beside_graph():
    for _ in range(3):
        step()
    for o in opts:
        o.zero_grad(set_to_none=True)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    try:
        torch.cuda.set_sync_debug_mode("error")
        with torch.cuda.graph(g, stream=side):
            step()
        torch.cuda.set_sync_debug_mode("default")
        print("capture OK, stage", stage, flush=True)
        for i in range(3):
            g.replay()
            torch.cuda.synchronize()
            print("replay", i, "OK", flush=True)
    except Exception:
        torch.cuda.set_sync_debug_mode("default")
        traceback.print_exc()
