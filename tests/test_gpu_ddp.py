"""Two ranks on ONE GPU (gloo): the reference's DistributedDataParallel wrap (main.py:335-337) and its step sequence
(engine.py:54-101: encode, decode, criterion, weighted sum, zero_grad, backward, clip, step) run UNCHANGED around the MI355X
model, and so does toist_amd.parallel.DistributedDataParallel.  After backward every rank must hold the same gradients, equal to
the mean of the two ranks' local gradients; the num_boxes all-reduce of the criterion (mdetr.py:997-1001) must see both ranks."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


PROBES = ["class_embed.weight", "bbox_embed.layers.2.bias", "transformer.decoder.layers.0.cross_attn_image.in_proj_weight",
          "transformer.encoder.layers.0.linear1.weight", "transformer.decoder.norm.weight", "transformer.resizer.fc.weight",
          "transformer.text_encoder.encoder.layer.0.attention.self.query.weight", "input_proj.weight", "backbone.0.body.layer4.2.conv3.weight",
          "backbone.0.body.layer2.0.conv1.weight", "query_embed.weight"]


def _step(model, criterion, weight_dict, batch, dev, clip=True):
    """engine.py:54-91 verbatim in structure."""
    samples, tok, targets, pmap = batch
    memory_cache = model(samples, tok, encode_and_save=True)
    outputs = model(samples, tok, encode_and_save=False, memory_cache=memory_cache)
    loss_dict = criterion(memory_cache, outputs, targets, pmap, None)
    losses = sum(loss_dict[k] * weight_dict[k] for k in loss_dict.keys() if k in weight_dict)
    model.zero_grad(set_to_none=True)
    losses.backward()
    if clip:
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1e9)
    return float(losses)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import toist_amd
    from toist_amd import harness, parallel
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    args = harness.default_args(device="cuda", dropout=0.0, enc_layers=2, dec_layers=2)
    batches = [harness.synthetic_batch(2, 128, 160, tokens=8, seed=40 + r, device=dev, max_targets=4) for r in range(world)]

    def build():
        torch.manual_seed(0)
        model, criterion, _, weight_dict = toist_amd.build_model(args)
        for n, b in model.named_buffers():
            if n.endswith("bn3.weight"):
                b.mul_(0.3)
        model.to(dev).eval()      # gradients flow as in training; eval() only switches the dropout layers off (RoBERTa's and the resizer's
        return model, criterion, weight_dict      # are not governed by args.dropout), so the expectation below is deterministic

    def probe(m):
        named = dict(m.named_parameters())
        return {n: named[n].grad.detach().float().cpu().clone() for n in PROBES}

    # expected: mean over ranks of the local gradients; num_boxes is all-reduced inside the criterion, so the local runs below are
    # issued by BOTH ranks in lock-step (rank r feeds batch j when it is j's turn) and rescaled to the two-rank num_boxes
    model, criterion, weight_dict = build()
    nb_world = max(sum(len(t["boxes"]) for b in batches for t in b[2]) / world, 1.0)
    want = None
    for j in range(world):
        nb_local = max(float(sum(len(t["boxes"]) for t in batches[j][2])), 1.0)   # both ranks feed batch j: world sum / world = local count
        _step(model, criterion, weight_dict, batches[j], dev, clip=False)
        g = {n: v * (nb_local / nb_world) / world for n, v in probe(model).items()}
        want = g if want is None else {n: want[n] + g[n] for n in g}
    res = {}

    def compare(got):
        """(smallest cosine, smallest and largest norm ratio) over the probed tensors"""
        cos_, rat = [], []
        for n in PROBES:
            a, b = got[n].flatten(), want[n].flatten()
            cos_.append(float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30)))
            rat.append(float(a.norm() / (b.norm() + 1e-30)))
        return min(cos_), min(rat), max(rat)

    # 1. torch's own DistributedDataParallel, as the reference wraps the model
    model, criterion, weight_dict = build()
    ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0], find_unused_parameters=True)
    _step(ddp, criterion, weight_dict, batches[rank], dev)
    got = probe(ddp.module)
    res["torch_ddp_cos"] = compare(got)
    gather = [None] * world
    dist.all_gather_object(gather, {n: float(v.double().sum()) for n, v in got.items()})
    res["torch_ddp_same_on_all_ranks"] = all(abs(gather[0][n] - gather[1][n]) <= 1e-6 * (abs(gather[0][n]) + 1e-12) for n in PROBES)
    del ddp
    # 2. the flat-buffer wrapper with the same constructor / .module surface
    model, criterion, weight_dict = build()
    ddp2 = parallel.DistributedDataParallel(model, device_ids=[0], find_unused_parameters=True)
    _step(ddp2, criterion, weight_dict, batches[rank], dev)
    res["toist_ddp_cos"] = compare(probe(ddp2.module))
    out[rank] = res
    dist.barrier()
    dist.destroy_process_group()


def test_reference_ddp_wrap_and_step_sequence(dev):
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    for r in range(world):
        assert out[r]["torch_ddp_same_on_all_ranks"], out[r]
        for key in ("torch_ddp_cos", "toist_ddp_cos"):
            cos_, lo, hi = out[r][key]
            assert cos_ > 0.995 and 0.97 < lo and hi < 1.03, out[r]


@pytest.mark.parametrize("world,steps", [(2, 2), (4, 3)])
def test_bench_ranks_on_one_gpu_gloo(dev, world, steps):
    """The N > 1 control flow of bench.py (six hipGraphs: forward + transformer backward | text backward || backbone layer 4 | layer 3 |
    layer 2 | tail, flat-buffer gradient all-reduces in between, num_boxes all-reduce, max-over-ranks timing) with two and with FOUR ranks
    sharing this GPU over gloo -- RCCL needs one GPU per rank, which the test box does not have; the launch line is the driver's.  Asserted:
    the graph structure, one collective per backward segment in the order the segments finish, their sizes, and bit-identical
    parameters on every rank after the timed steps (three at four ranks)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TOIST_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(root, "bench.py"), "--gpus", str(world), "--steps", str(steps), "--warmup", "1", "--no-cpu-baseline", "--no-roofline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env, cwd=root)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and lines, out.stderr[-3000:]
    res = json.loads(lines[-1])
    assert res["n_gpus"] == world and res["config"]["global_batch"] == 8 * world and res["value"] > 0 and res["scaling"] == "weak" and res["steps"] == steps
    assert list(res["collectives"]) == ["transformer+heads", "text_encoder", "backbone stage 1 of 3 (layer4 first)", "backbone stage 2 of 3 (layer4 first)",
                                        "backbone stage 3 of 3 (layer4 first)"], list(res["collectives"])        # the order the segments are enqueued in
    assert "6 hipGraphs" in res["config"]["launch"], res["config"]["launch"]
    # one collective per backward segment: the backbone's gradients travel per stage (layer4 first), under the stages below
    assert set(res["collectives"]) == {"transformer+heads", "text_encoder", "backbone stage 1 of 3 (layer4 first)", "backbone stage 2 of 3 (layer4 first)",
                                       "backbone stage 3 of 3 (layer4 first)"}, sorted(res["collectives"])
    assert all(c["ms"] > 0 and c["busbw_GBps"] >= 0 for c in res["collectives"].values())       # (gloo on a loaded box: the rate may round to 0.0)
    sizes = [res["collectives"]["backbone stage %d of 3 (layer4 first)" % i]["bytes"] for i in (1, 2, 3)]
    assert sizes[0] > 50e6 and sizes[1] > 90e6 and sizes[2] < 10e6, sizes       # layer4 ~ 60 MB, layer3 ~ 104 MB, layer2 ~ 5 MB of fp32 gradients
    assert res["config"]["parameters_identical_across_ranks"] is True
    # ranks that share a device must not use the XCD-resident decoder launches (each needs all 256 CUs) -- the head segment's
    # graph then holds the per-op decoder kernels; one rank per GPU (the driver's 8-GPU run, test_bench_split_graphs_on_one_rank below) uses them
    assert res["config"]["decoder"].startswith("per-op launches"), res["config"]["decoder"]


def test_bench_split_graphs_on_one_rank(dev):
    """The configuration the first real multi-GPU run executes on every rank -- six hipGraphs with the XCD-resident decoder launches inside the
    head segment -- on the one GPU of the test box (`--split-graph`: the N > 1 graph structure at world size 1).  Asserted: the structure, that the
    decoder really ran as the two XCD-resident launches (no expired spin: bench.py calls kernels.xdec_check), a finite loss."""
    import json
    import math
    import subprocess
    import sys
    from toist_amd import kernels as k
    if not k.xdec_supported(8, 100, 416, 6):
        pytest.skip("device without 8 XCDs x 32 CUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--split-graph", "--steps", "3", "--warmup", "1", "--repeats", "1", "--no-cpu-baseline", "--no-roofline",
           "--no-secondary"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and lines, out.stderr[-3000:]
    res = json.loads(lines[-1])
    assert "6 hipGraphs" in res["config"]["launch"], res["config"]["launch"]
    assert res["config"]["decoder"].startswith("2 XCD-resident launches"), res["config"]["decoder"]
    assert res["n_gpus"] == 1 and res["value"] > 0 and math.isfinite(res["config"]["final_loss"])
