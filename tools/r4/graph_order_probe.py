"""Does the ORDER in which two parallel branches were captured decide how a replayed hipGraph overlaps them?

Two independent branches inside one captured graph: A = a chain of NA small kernels (a few workgroups each, the shape of the
text encoder's launches) on a side stream, B = a chain of NB chip-filling kernels on the capturing stream.  Captured three ways --
A first, B first, interleaved (rA kernels of A, then one of B, ...) -- and replayed.  Perfect overlap = max(A alone, B alone);
none = the sum.  GPU only."""
import sys

import torch

dev = torch.device("cuda")
NA, NB = int(sys.argv[1]) if len(sys.argv) > 1 else 120, int(sys.argv[2]) if len(sys.argv) > 2 else 40

small = [torch.randn(128, 768, device=dev, dtype=torch.bfloat16) for _ in range(2)]
wsmall = torch.randn(768, 768, device=dev, dtype=torch.bfloat16) * 0.03
big = [torch.randn(12800, 1024, device=dev, dtype=torch.bfloat16) for _ in range(2)]
wbig = torch.randn(1024, 1024, device=dev, dtype=torch.bfloat16) * 0.03


def a_step(i):
    torch.mm(small[i & 1], wsmall, out=small[(i + 1) & 1])


def b_step(i):
    torch.mm(big[i & 1], wbig, out=big[(i + 1) & 1])


def capture(order, pre=0):
    side = torch.cuda.Stream()
    cap = torch.cuda.Stream()
    cap.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(cap):
        for i in range(2):
            a_step(i)
            b_step(i)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=cap):
            if order == "A only":
                for i in range(NA):
                    a_step(i)
            elif order == "B only":
                for i in range(NB):
                    b_step(i)
            else:
                for i in range(pre):          # nodes BEFORE the fork: both branches then hang off a common predecessor
                    a_step(i)
                fork = torch.cuda.Event()
                fork.record(cap)
                side.wait_event(fork)
                if order == "A first":
                    with torch.cuda.stream(side):
                        for i in range(NA):
                            a_step(i)
                    for i in range(NB):
                        b_step(i)
                elif order == "B first":
                    for i in range(NB):
                        b_step(i)
                    with torch.cuda.stream(side):
                        for i in range(NA):
                            a_step(i)
                else:       # interleaved
                    ia = 0
                    per = (NA + NB - 1) // NB
                    for i in range(NB):
                        with torch.cuda.stream(side):
                            for _ in range(per):
                                if ia < NA:
                                    a_step(ia)
                                    ia += 1
                        b_step(i)
                cap.wait_stream(side)
    torch.cuda.current_stream().wait_stream(cap)
    return g


def timed(g, reps=20):
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return 1000.0 * e0.elapsed_time(e1) / reps


print(f"A = {NA} x mm(128 x 768 x 768) on a side stream, B = {NB} x mm(12800 x 1024 x 1024) on the capturing stream; us per replay")
for pre in (0, 2):
    for order in ("A only", "B only", "A first", "B first", "interleaved"):
        if pre and "only" in order:
            continue
        g = capture(order, pre)
        print(f"  {pre} kernels before the fork, captured {order:12s} {timed(g):9.1f} us", flush=True)

# ---- is a replay's enqueue allowed to run ahead of the GPU?  host time per replay() call (no synchronisation inside the loop) ----
import time


def host_times(gs, reps=12):
    torch.cuda.synchronize()
    ts = [time.perf_counter()]
    for i in range(reps):
        gs[i % len(gs)].replay()
        ts.append(time.perf_counter())
    torch.cuda.synchronize()
    end = time.perf_counter()
    per = [1e6 * (b - a) for a, b in zip(ts, ts[1:])]
    return per, 1e6 * (end - ts[0]) / reps


print("host microseconds spent inside each replay() call (12 calls back to back, then one synchronize); last column = wall per replay")
for name, order in (("B only (40 long kernels)", "B only"), ("A only (120 short kernels)", "A only"), ("A first (fork)", "A first")):
    g1, g2 = capture(order), capture(order)
    per, wall = host_times([g1])
    print(f"  one exec,  {name:28s}", " ".join(f"{p:6.0f}" for p in per), f" | {wall:7.1f}")
    per, wall = host_times([g1, g2])
    print(f"  two execs, {name:28s}", " ".join(f"{p:6.0f}" for p in per), f" | {wall:7.1f}")
