import sys, time, torch
sys.path.insert(0, "/root/repo")
import toist_amd
from toist_amd import harness, parallel
dev = torch.device("cuda:0")
args = harness.default_args(device="cuda")
torch.manual_seed(0)
model, criterion, _, wd = toist_amd.build_model(args)
model.to(dev).train()
named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
opt = torch.optim.AdamW([p for _, p in named], lr=1e-4, fused=True)
samples, tok, targets, pmap = harness.synthetic_batch(8, 640, 640, tokens=16, seed=1000, device=dev)
def phase(name, fn, store):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    store.setdefault(name, []).append((1e3 * (t1 - t0), 1e3 * (t2 - t0)))
    return r
acc = {}
for it in range(6):
    opt.zero_grad(set_to_none=True)
    mc = phase("encode", lambda: model(samples, tok, encode_and_save=True), acc)
    out = phase("decode", lambda: model(samples, tok, encode_and_save=False, memory_cache=mc), acc)
    losses = phase("criterion", lambda: criterion(mc, out, targets, pmap, None), acc)
    total = sum(losses[k] * wd[k] for k in losses if k in wd)
    phase("backward", lambda: total.backward(), acc)
    phase("clip", lambda: torch.nn.utils.clip_grad_norm_([p for _, p in named], 0.1, foreach=True), acc)
    phase("opt", lambda: opt.step(), acc)
for k, v in acc.items():
    h = sum(x[0] for x in v[2:]) / len(v[2:]); t = sum(x[1] for x in v[2:]) / len(v[2:])
    print(f"{k:10s} host enqueue {h:7.2f} ms   with sync {t:7.2f} ms")
