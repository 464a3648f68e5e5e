"""Where does the eager distillation step (configs[4]) synchronise the host with the GPU, and is the step host- or GPU-bound?
torch.cuda.set_sync_debug_mode("warn") reports every synchronising torch call of one step; then the host time to ISSUE a step against its wall time."""
import sys
import time
import warnings

import torch

sys.path.insert(0, ".")
import toist_amd  # noqa: E402
from toist_amd import engine, harness, kernels  # noqa: E402
from toist_amd.optim import FusedClipAdamWEMA  # noqa: E402

dev = torch.device("cuda")
B = 4
args = harness.default_args(device="cuda", distillation=True, cluster=True, nsthl2_loss=True, softkd_loss=True, train_batch_size=B)
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
    return FusedClipAdamWEMA([{"params": [p for n, p in named]}], lr=1e-4, weight_decay=1e-4, max_norm=0.1)


opts = [tail(model), tail(model_noun)]
batch = harness.synthetic_distill_batch(B, 640, 640, tokens=16, seed=1000, device=dev)


def step():
    kernels.SEED_DEV.add_(1000003)
    for o in opts:
        o.zero_grad(set_to_none=True)
    total, _ = harness.distillation_step(model, model_noun, criterion, cluster_criterion, weight_dict, batch)
    total.backward()
    for o in opts:
        o.step()
    return total


for _ in range(3):
    step()
torch.cuda.synchronize()
print("--- synchronising calls of one step (torch sync debug mode) ---", flush=True)
torch.cuda.set_sync_debug_mode("warn")
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    step()
torch.cuda.set_sync_debug_mode("default")
import collections
seen = collections.Counter()
for x in w:
    if "synchroniz" in str(x.message).lower():
        seen[f"{x.filename.split('/')[-1]}:{x.lineno}  {str(x.message)[:90]}"] += 1
for k_, n in seen.most_common():
    print(f"  {n:3d} x {k_}")
torch.cuda.synchronize()
for label in ("issue + wait", "issue + wait"):
    t0 = time.perf_counter()
    for _ in range(5):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"5 steps: host issue {1000 * (t1 - t0) / 5:.1f} ms per step, wall {1000 * (t2 - t0) / 5:.1f} ms per step", flush=True)
