"""cProfile of eager training steps (host side): where the Python launch path spends its time.  GPU only."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import toist_amd  # noqa: E402
from toist_amd import harness, kernels  # noqa: E402
from toist_amd.optim import FusedClipAdamWEMA  # noqa: E402

dev = torch.device("cuda:0")
args = harness.default_args(device="cuda")
torch.manual_seed(0)
model, criterion, _, wd = toist_amd.build_model(args)
model.to(dev).train()
named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
opt = FusedClipAdamWEMA([{"params": [p for _, p in named]}], max_norm=0.1)
kernels.SEED_DEV = torch.zeros(1, dtype=torch.int64, device=dev)
samples, tok, targets, pmap = harness.synthetic_batch(8, 640, 640, tokens=16, seed=1000, device=dev)


def step():
    opt.zero_grad(set_to_none=True)
    mc = model(samples, tok, encode_and_save=True)
    out = model(samples, tok, encode_and_save=False, memory_cache=mc)
    losses = criterion(mc, out, targets, pmap, None)
    total = sum(losses[k] * wd[k] for k in losses if k in wd)
    total.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
