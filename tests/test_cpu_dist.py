"""world_size-2 gloo tests of the data-parallel pieces (CPU): reduce_dict, the num_boxes all-reduce of
the set criterion (sum, / world, clamp >= 1; reference mdetr.py:997-1001) and the flat-buffer gradient
all-reduce (mean) of toist_amd.parallel.GradSync."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from toist_amd import dist as tdist
    from toist_amd import functions, parallel
    from toist_amd.mdetr import SetCriterion
    res = {}
    red = tdist.reduce_dict({"b": torch.tensor(float(rank + 1)), "a": torch.tensor(10.0 * (rank + 1))})
    res["reduce"] = (float(red["a"]), float(red["b"]))
    crit = SetCriterion(None, 255, matcher=None, eos_coef=0.1, losses=["labels"], temperature=0.07)
    targets = [{"labels": torch.ones(3 if rank == 0 else 0)}]
    res["num_boxes"] = float(crit._num_boxes(targets, torch.device("cpu")))
    res["num_boxes_empty"] = float(crit._num_boxes([{"labels": torch.ones(0)}], torch.device("cpu")))
    sync = parallel.GradSync()
    flat = torch.full((10,), float(rank + 1))
    with sync:
        assert functions.GRAD_SYNC is not None
        functions.GRAD_SYNC(flat)
        sync.finish()
    assert functions.GRAD_SYNC is None
    res["flat"] = flat.tolist()
    lin = torch.nn.Linear(2, 2)
    torch.manual_seed(rank)
    torch.nn.init.normal_(lin.weight)
    parallel.broadcast_parameters(lin)
    res["w"] = lin.weight.detach().flatten().tolist()
    out[rank] = res
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_two_gloo():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    r0, r1 = out[0], out[1]
    assert r0["reduce"] == r1["reduce"] == (15.0, 1.5)
    assert r0["num_boxes"] == r1["num_boxes"] == 1.5   # (3 + 0) / 2
    assert r0["num_boxes_empty"] == 1.0                # clamp(min=1)
    assert r0["flat"] == r1["flat"] == [1.5] * 10      # mean of 1 and 2
    assert r0["w"] == r1["w"]


# ---- DistributedDataParallel compatibility (reference main.py:335-337, engine.py:54-101) ------------------------------
class _Toy(torch.nn.Module):
    """A module whose `lin` runs through functions.run_program (one autograd node, flat gradient buffer) and whose `plain`
    is differentiated by torch.autograd itself -- the two ways a parameter of the product receives its gradient.  The
    program body uses torch ops so the test runs without a GPU; the gradient-delivery machinery under test is the real one."""

    def __init__(self):
        super().__init__()
        self.lin = torch.nn.Linear(3, 2)
        self.plain = torch.nn.Linear(2, 1)
        self.unused = torch.nn.Linear(2, 2)        # find_unused_parameters=True of the reference
        self._cache = {}

    def forward(self, x):
        from collections import OrderedDict
        from toist_amd import engine, functions
        named = OrderedDict(self.lin.named_parameters())

        def prog(tape, ps, xin):
            W, b = ps["weight"], ps["bias"]
            out = engine.Var(xin.data @ W.f32.t() + b.f32)

            def bwd():
                g = out.take_grad()
                W.g.add_(g.t() @ xin.data)
                b.g.add_(g.sum(0))

            tape.record(bwd)
            return [out], None

        (y,) = functions.run_program(prog, named, [x], cache=self._cache, training=True)
        return self.plain(y).sum()


def _ddp_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from toist_amd import parallel
    torch.manual_seed(0)
    ref = _Toy()
    xs = [torch.arange(6, dtype=torch.float32).reshape(2, 3) * (r + 1) for r in range(world)]
    # expected: mean over ranks of the local gradients (what DDP delivers)
    want = {}
    for r in range(world):
        ref.zero_grad(set_to_none=True)
        ref(xs[r]).backward()
        for n, p in ref.named_parameters():
            if p.grad is not None:
                want[n] = want.get(n, 0) + p.grad.detach().clone() / world
    res = {}

    def grads(m):
        return {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}

    def close(a, b):
        return set(a) == set(b) and all(torch.allclose(a[k], b[k], atol=1e-6) for k in a)

    # 1. torch's own wrapper, exactly as main.py:336 builds it; the engine.py call sequence around it
    torch.manual_seed(0)
    m1 = torch.nn.parallel.DistributedDataParallel(_Toy(), find_unused_parameters=True)
    opt = torch.optim.SGD(m1.parameters(), lr=0.0)
    opt.zero_grad()
    m1(xs[rank]).backward()
    torch.nn.utils.clip_grad_norm_(m1.parameters(), 0.1 * 1e9)
    opt.step()
    res["torch_ddp"] = close(grads(m1.module), want)
    # 2. the flat-buffer wrapper with the same surface
    torch.manual_seed(0)
    m2 = parallel.DistributedDataParallel(_Toy(), device_ids=None, find_unused_parameters=True)
    m2(xs[rank]).backward()
    res["toist_ddp"] = close(grads(m2.module), want)
    with m2.no_sync():
        m2.module.zero_grad(set_to_none=True)
        m2(xs[rank]).backward()
    res["no_sync_local"] = not close(grads(m2.module), want)
    # 3. GradSync(model): flat buffers + the parameters no flat buffer carries
    torch.manual_seed(0)
    m3 = _Toy()
    sync = parallel.GradSync(m3)
    with sync:
        m3(xs[rank]).backward()
        sync.finish()
    res["gradsync"] = close(grads(m3), want)
    # 4. gradient accumulation: only the last micro-step is wrapped
    m3.zero_grad(set_to_none=True)
    m3(xs[rank]).backward()
    with sync:
        m3(xs[rank]).backward()
        sync.finish()
    res["accumulate"] = close(grads(m3), {k: 2 * v for k, v in want.items()})
    # 5. a parameter with a gradient on rank 0 only (conditionally used branch): the ranks agree on the set to reduce (the
    # collective sequences used to differ and hang); the rank without a gradient contributes zeros, like DDP(find_unused_parameters=True)
    torch.manual_seed(0)
    m4 = _Toy()
    extra = torch.nn.Parameter(torch.ones(3))
    m4.register_parameter("extra", extra)
    sync4 = parallel.GradSync(m4)
    with sync4:
        loss = m4(xs[rank])
        if rank == 0:
            loss = loss + (extra * torch.tensor([1.0, 2.0, 3.0])).sum()
        loss.backward()
        sync4.finish()
    g4 = grads(m4)
    res["unused_on_one_rank"] = "extra" in g4 and torch.allclose(g4["extra"], torch.tensor([0.5, 1.0, 1.5])) and \
        close({k: v for k, v in g4.items() if k != "extra"}, want)
    # 6. the same gradient in a flat program buffer on rank 0 and stand-alone on rank 1 (the ranks ran different schedules): EVERY rank raises
    # (round 3 / 4: rank 0 skipped the stand-alone reduction rank 1 entered -- a hang)
    w = torch.nn.Parameter(torch.ones(4))
    holder = torch.nn.Module()
    holder.register_parameter("w", w)
    flat = torch.full((4,), float(rank + 1))
    w.grad = flat if rank == 0 else torch.full((4,), 5.0)
    sync6 = parallel.GradSync(holder)
    sync6._launch(flat)                      # both ranks issue the flat collective; only rank 0's covers w.grad
    try:
        sync6.finish()
        res["conflict_raises_everywhere"] = False
    except RuntimeError as e:
        res["conflict_raises_everywhere"] = "flat program buffer" in str(e)
    dist.barrier()
    # 7. static_set=True: the agreement of the first step is kept -- the second finish() issues no agreement collective
    torch.manual_seed(0)
    m7 = _Toy()
    extra7 = torch.nn.Parameter(torch.ones(3))
    m7.register_parameter("extra", extra7)
    sync7 = parallel.GradSync(m7, static_set=True)
    calls = []
    real = dist.all_reduce

    def counting(t, *a, **kw):
        calls.append(tuple(t.shape))
        return real(t, *a, **kw)

    counts = []
    for step in range(2):
        m7.zero_grad(set_to_none=True)
        calls.clear()
        dist.all_reduce = counting
        try:
            with sync7:
                (m7(xs[rank]) + (extra7 * torch.tensor([1.0, 2.0, 3.0])).sum()).backward()
                sync7.finish()
        finally:
            dist.all_reduce = real
        counts.append(sum(1 for sh in calls if len(sh) == 2 and sh[1] == 3))        # the [n_params, 3] one-hot agreement
    g7 = grads(m7)
    res["static_set"] = counts == [1, 0] and torch.allclose(g7["extra"], torch.tensor([1.0, 2.0, 3.0])) and close({k: v for k, v in g7.items() if k != "extra"}, want)
    out[rank] = res
    dist.barrier()
    dist.destroy_process_group()


def test_ddp_wrapping_delivers_averaged_gradients():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_ddp_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    for r in range(world):
        assert out[r] == {"torch_ddp": True, "toist_ddp": True, "no_sync_local": True, "gradsync": True, "accumulate": True, "unused_on_one_rank": True,
                          "conflict_raises_everywhere": True, "static_set": True}, out[r]
