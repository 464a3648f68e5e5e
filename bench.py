#!/usr/bin/env python
"""Headline benchmark: train images/s (640x640 synthetic images + 16-token captions, batch 8 per GPU)
for the MI355X-native TOIST/MDETR hot path (BASELINE.json: metric / configs[1]).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

One step = zero_grad -> encode -> decode -> SetCriterion (labels, boxes, cardinality, 5 aux layers) ->
backward (+ flat-buffer RCCL gradient all-reduce when N > 1) -> clip_grad_norm_(0.1) -> AdamW -> EMA,
in train mode (dropout 0.1 on).  Rank 0 prints ONE JSON line.  `roofline` times the dominant kernel
(the bf16 MFMA implicit-GEMM family, toist_amd/csrc/gemm.hip) with HIP events on the launch stream
inside the timed region; `cpu_baseline` times the fp32 oracle (a port of the reference's CPU path) on
this box's host cores at N = 1.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
GFLOP_PER_IMG_TRAIN = 397.7  # detection train step, stem + layer1 frozen


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU")
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--profile-all", action="store_true", help="time every GEMM launch (diagnostic; adds host overhead)")
    ap.add_argument("--masks", action="store_true", help="config 3: add the segmentation head and the mask losses")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel from Python instead of replaying a captured hipGraph")
    return ap.parse_args()


def cpu_baseline(size):
    """fp32 oracle (port of the reference CPU path) on the host cores: B=1 detection train step
    (forward + SetCriterion with 6 matcher calls + backward), bounded to a couple of iterations."""
    from oracle import model_ref
    import toist_amd
    from toist_amd import harness
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    args = harness.default_args(device="cpu")
    torch.manual_seed(0)
    model, _, _, weight_dict = toist_amd.build_model(args)
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point() and "running" not in k and ".bn" not in k and "downsample.1" not in k
                                              and ".layer1." not in k and "body.conv1" not in k)
          for k, v in model.state_dict().items()}
    del model
    samples, tok, targets, pmap = harness.synthetic_batch(1, size, size, tokens=16, seed=1000, max_targets=10)

    def step():
        mc = model_ref.mdetr_encode(sd, samples.tensors, samples.mask, tok["input_ids"], tok["attention_mask"])
        out = model_ref.mdetr_decode(sd, mc)
        losses = model_ref.set_criterion(out, targets, pmap)
        total = sum(losses[k] * weight_dict[k] for k in losses if k in weight_dict)
        total.backward()
        for v in sd.values():
            v.grad = None

    step()
    t0 = time.time()
    n = 0
    while n < 2 or (time.time() - t0 < 12 and n < 6):
        step()
        n += 1
    dt = (time.time() - t0) / n
    return {"value": round(1.0 / dt, 4), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"oracle fp32 detection train step (fwd + criterion + bwd), B=1 {size}x{size}, 1 warm-up + {n} timed iterations"}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        torch.distributed.init_process_group("nccl", init_method="env://", world_size=world, rank=rank, device_id=dev)

    import toist_amd
    from toist_amd import harness, kernels, parallel
    args = harness.default_args(device="cuda", masks=a.masks, mask_model="smallconv" if a.masks else "none")
    torch.manual_seed(0)
    model, criterion, _, weight_dict = toist_amd.build_model(args)
    model.to(dev)
    parallel.broadcast_parameters(model)
    model.train()
    criterion.train()

    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    groups = [
        {"params": [p for n, p in named if "backbone" not in n and "text_encoder" not in n]},
        {"params": [p for n, p in named if "backbone" in n], "lr": args.lr_backbone},
        {"params": [p for n, p in named if "text_encoder" in n], "lr": args.text_encoder_lr},
    ]
    use_graph = (world == 1) and not a.no_graph and not a.profile_all and not a.masks
    opt = torch.optim.AdamW(groups, lr=args.lr, weight_decay=args.weight_decay, fused=True, capturable=use_graph)
    kernels.SEED_DEV = torch.zeros(1, dtype=torch.int64, device=dev)  # bumped every step: fresh dropout masks per replay
    ema_src = [v for v in model.state_dict().values() if v.is_floating_point()]
    ema = [v.detach().clone() for v in ema_src]
    all_params = [p for _, p in named]

    samples, tok, targets, pmap = harness.synthetic_batch(a.batch, a.size, a.size, tokens=16, seed=1000 + rank, device=dev, with_masks=a.masks)
    sync = parallel.GradSync()

    def step(zero=True):
        if zero:
            opt.zero_grad(set_to_none=True)
        kernels.SEED_DEV.add_(1000003)
        with sync:
            mc = model(samples, tok, encode_and_save=True)
            out = model(samples, tok, encode_and_save=False, memory_cache=mc)
            losses = criterion(mc, out, targets, pmap, None)
            total = sum(losses[k] * weight_dict[k] for k in losses if k in weight_dict)
            total.backward()
            sync.finish()
        torch.nn.utils.clip_grad_norm_(all_params, args.clip_max_norm, foreach=True)
        opt.step()
        with torch.no_grad():
            torch._foreach_mul_(ema, 0.9998)
            torch._foreach_add_(ema, ema_src, alpha=1.0 - 0.9998)
        return total

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    graph = None
    if use_graph:
        # The whole step (forward, criterion, backward, clip, AdamW, EMA: ~1500 kernel launches) is captured
        # once into a hipGraph and replayed: the Python/ctypes launch path (~10-20 us per launch) would
        # otherwise bound the step.  Inputs are static device tensors; dropout masks change per replay
        # through the device-side seed word.
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(a.warmup, 2)):
                step()
            opt.zero_grad(set_to_none=True)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                static_loss = step(zero=False)
        torch.cuda.current_stream().wait_stream(side)
        graph.replay()

        def run_step():
            graph.replay()
            return static_loss
    else:
        for _ in range(a.warmup):
            step()
        run_step = step
    prof = None
    if rank == 0 and not a.no_roofline and not use_graph:
        prof = {"key": None if a.profile_all else (65, kernels.A_CONV, kernels.B_ROWK), "records": [], "other": {}}
    barrier()
    kernels.PROFILE = prof
    t0 = time.perf_counter()
    for _ in range(a.steps):
        last = run_step()
    barrier()
    dt = time.perf_counter() - t0
    kernels.PROFILE = None
    loss_val = float(last)
    if rank == 0 and not a.no_roofline and use_graph:
        # kernel-level timing needs per-launch HIP events, which a replayed graph cannot carry: time the
        # same K steps once more, eagerly, on the same stream right after the timed region
        prof = {"key": (65, kernels.A_CONV, kernels.B_ROWK), "records": [], "other": {}}
        kernels.PROFILE = prof
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize()
        kernels.PROFILE = None
    if world > 1:
        t = torch.tensor([dt], device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t)

    if rank == 0:
        ips = a.batch * world * a.steps / dt
        res = {
            "metric": "train images/sec/node (640x640, bs=8/GPU) + matcher index bit-match", "value": round(ips, 3), "unit": "images/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1000 * dt / a.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": ("configs[2] (det + mask head + mask losses): " if a.masks else "") + f"configs[1]: ResNet-101 + RoBERTa-base + 6+6 transformer, 100 queries, batch {a.batch}/GPU {a.size}x{a.size}, "
                                   "16-token captions, detection loss (labels+boxes+cardinality, 5 aux layers), dropout 0.1, "
                                   "clip 0.1 + AdamW + EMA; random-init weights",
                       "global_batch": a.batch * world, "parallelism": f"dp{world}", "final_loss": round(loss_val, 4), "launch": "hipGraph replay" if use_graph else "eager",
                       "mfma_frac_whole_step": round(ips / world * GFLOP_PER_IMG_TRAIN / 1000.0 / PEAK_BF16_TFLOPS, 5)},
        }
        if prof is not None and prof["records"]:
            tot_ms, tot_fl, per_key = 0.0, 0.0, {}
            for e0, e1, fl, key in prof["records"]:
                ms = e0.elapsed_time(e1)
                tot_ms += ms
                tot_fl += fl
                k_ = per_key.setdefault(str(key), [0.0, 0.0, 0])
                k_[0] += ms
                k_[1] += fl
                k_[2] += 1
            n = len(prof["records"])
            ach = tot_fl / (tot_ms * 1e-3) / 1e12
            res["roofline"] = {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                               "frac": round(ach / PEAK_BF16_TFLOPS, 5), "traffic": None,
                               "kernel": "gemm_kernel<64,64,64,A_CONV,B_ROWK> (implicit-GEMM conv forward)" if prof["key"] else "all gemm_kernel launches",
                               "timed": "HIP events around each launch, %d eager steps %s" % (a.steps, "after the graph-replayed timed region" if use_graph else "inside the timed region"),
                               "launches": n, "avg_launch_us": round(1000 * tot_ms / n, 2), "avg_gflop_per_launch": round(tot_fl / n / 1e9, 3)}
            if a.profile_all:
                res["roofline"]["per_variant"] = {k_: {"ms_per_step": round(v[0] / a.steps, 3), "tflops": round(v[1] / (v[0] * 1e-3) / 1e12, 1),
                                                      "launches_per_step": v[2] // a.steps} for k_, v in per_key.items()}
        if not a.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline(a.size)
        print(json.dumps(res))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
