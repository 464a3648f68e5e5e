"""Census of torch operators (device tensors) per source line in one EAGER distillation step (configs[4]): the step is host-bound, every small torch
launch costs ~10-20 us of issue time.  usage (GPU box): python tools/r5/distill_glue.py [--batch 4]"""
import argparse, collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import toist_amd
from toist_amd import harness, kernels, parallel, engine
from toist_amd.optim import FusedClipAdamWEMA
from torch.utils._python_dispatch import TorchDispatchMode

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4)
ap.add_argument("--size", type=int, default=640)
a = ap.parse_args()
dev = torch.device("cuda:0")
args = harness.default_args(device="cuda", distillation=True, cluster=True, nsthl2_loss=True, softkd_loss=True, train_batch_size=a.batch)
torch.manual_seed(0)
model, criterion, cluster_criterion, weight_dict = toist_amd.build_model(args)
model_noun, _, _, _ = toist_amd.build_model(args)
for m in (model, model_noun):
    m.to(dev)
    m.train()
cluster_criterion.to(dev)
cluster_criterion.full_label.fill_(1)
engine.REUSE_GRAD_BUFFERS = True
kernels.SEED_DEV = torch.zeros(1, dtype=torch.int64, device=dev)


def tail(m):
    named = [(n, p) for n, p in m.named_parameters() if p.requires_grad]
    groups = [{"params": [p for n, p in named if "backbone" not in n and "text_encoder" not in n]},
              {"params": [p for n, p in named if "backbone" in n], "lr": args.lr_backbone},
              {"params": [p for n, p in named if "text_encoder" in n], "lr": args.text_encoder_lr}]
    src = [v for v in m.state_dict().values() if v.is_floating_point()]
    return FusedClipAdamWEMA(groups, lr=args.lr, weight_decay=args.weight_decay, max_norm=args.clip_max_norm,
                             ema=list(zip(src, [v.detach().clone() for v in src])), ema_decay=0.9998)


opts = [tail(model), tail(model_noun)]
batch = harness.synthetic_distill_batch(a.batch, a.size, a.size, tokens=16, seed=1000, device=dev)
sync = parallel.GradSync([model, model_noun])


def step():
    kernels.SEED_DEV.add_(1000003)
    for o in opts:
        o.zero_grad(set_to_none=True)
    with sync:
        total, _ = harness.distillation_step(model, model_noun, criterion, cluster_criterion, weight_dict, batch)
        total.backward()
        sync.finish()
    for o in opts:
        o.step()
    return total


NO_KERNEL = ("aten.empty", "aten.new_empty", "aten.empty_like", "aten.empty_strided", "aten.resize_", "aten.set_", "aten.record_stream", "aten._local_scalar_dense",
             "aten.lift_fresh", "aten.is_", "aten.sym_", "aten._has_compatible", "aten.is_pinned")
agg = collections.OrderedDict()


class Glue(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if any(name.startswith(v) for v in NO_KERNEL):
            return out
        try:
            rets = func._schema.returns
            if rets and all(r.alias_info is not None and not r.alias_info.is_write for r in rets):
                return out
        except Exception:
            pass
        if not any(torch.is_tensor(x) and x.is_cuda for x in list(args) + list((kwargs or {}).values()) + ([out] if torch.is_tensor(out) else [])):
            return out
        fr = "(no toist frame)"
        for f in reversed(traceback.extract_stack()[:-1]):
            if "toist_amd/" in f.filename and not f.filename.endswith("kernels.py"):
                fr = f"toist_amd/{f.filename.split('toist_amd/')[-1]}:{f.lineno} {f.name}"
                break
        agg.setdefault((fr, name), [0])[0] += 1
        return out


for _ in range(3):
    step()
torch.cuda.synchronize()
with Glue():
    step()
torch.cuda.synchronize()
print(f"torch operators on device tensors in one eager distillation step (views excluded): {sum(v[0] for v in agg.values())}")
for (frame, name), (calls,) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print(f"  {calls:4d} x {name:36s} {frame}")
