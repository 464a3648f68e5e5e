"""Where the distillation step (bench.py --distill) spends its wall time, by stage (GPU only; synchronises per stage)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import toist_amd  # noqa: E402
from toist_amd import harness, kernels  # noqa: E402

dev = torch.device("cuda")
args = harness.default_args(device="cuda", distillation=True, cluster=True, nsthl2_loss=True, softkd_loss=True, train_batch_size=4)
torch.manual_seed(0)
model, criterion, cc, wd = toist_amd.build_model(args)
model_noun, _, _, _ = toist_amd.build_model(args)
model.to(dev).train(); model_noun.to(dev).train(); cc.to(dev); cc.full_label.fill_(1)
kernels.SEED_DEV = torch.zeros(1, dtype=torch.int64, device=dev)
batch = harness.synthetic_distill_batch(4, 640, 640, tokens=16, seed=1000, device=dev)
T = {}


def lap(name, t0):
    torch.cuda.synchronize()
    T[name] = T.get(name, 0.0) + time.perf_counter() - t0
    return time.perf_counter()


for it in range(6):
    if it == 2:
        T.clear()
    model.zero_grad(set_to_none=True); model_noun.zero_grad(set_to_none=True)
    s_n, s_s = batch["samples"]; t_n, t_s = batch["targets"]; c_n, c_s = batch["captions"]; k_n, k_s = batch["tokenized"]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    mc_n = model_noun(s_n, k_n, encode_and_save=True); t0 = lap("encode teacher", t0)
    mc_n = cc.update_memory(mc_n, t_n, c_n); t0 = lap("update_memory (bank + k-means + substitution)", t0)
    o_n = model_noun(s_n, k_n, encode_and_save=False, memory_cache=mc_n); t0 = lap("decode teacher", t0)
    mc_s = model(s_s, k_s, encode_and_save=True); t0 = lap("encode student", t0)
    mc_s, lc = cc(mc_s, t_s, c_s); t0 = lap("cluster forward (k-means + substitution)", t0)
    o_s = model(s_s, k_s, encode_and_save=False, memory_cache=mc_s); t0 = lap("decode student", t0)
    losses = criterion([mc_n, mc_s], [o_n, o_s], [t_n, t_s], batch["positive_map"], None); t0 = lap("paired criterion (matcher x2, softkd x6, nsthl2)", t0)
    losses.update(lc)
    total = sum(losses[k] * wd[k] for k in losses if k in wd)
    total.backward(); t0 = lap("backward (both models)", t0)
for k, v in T.items():
    print(f"{k:55s} {1000 * v / 4:8.2f} ms")
