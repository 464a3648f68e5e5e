"""Does a parallel branch of a replayed hipGraph start late when its queue has been idle for a while?

Graph: PRE chip-filling kernels on the capturing stream, then a fork: A (NA short kernels, side stream) beside B (NB long kernels,
capturing stream), join.  If both branches start at the fork the replay takes PRE + max(A, B); a branch that starts late shows as
more.  Variants: who continues on the capturing stream, and a `keep-warm` variant in which the side stream also runs one tiny kernel
per PRE kernel before the fork (so its queue never goes idle).  GPU only."""
import sys

import torch

dev = torch.device("cuda")
NA, NB = 120, 40
small = [torch.randn(128, 768, device=dev, dtype=torch.bfloat16) for _ in range(2)]
wsmall = torch.randn(768, 768, device=dev, dtype=torch.bfloat16) * 0.03
big = [torch.randn(12800, 1024, device=dev, dtype=torch.bfloat16) for _ in range(2)]
wbig = torch.randn(1024, 1024, device=dev, dtype=torch.bfloat16) * 0.03
tiny = torch.zeros(64, device=dev)


def a_step(i):
    torch.mm(small[i & 1], wsmall, out=small[(i + 1) & 1])


def b_step(i):
    torch.mm(big[i & 1], wbig, out=big[(i + 1) & 1])


def capture(pre, small_on_side=True, warm=False, fork=True):
    side = torch.cuda.Stream()
    cap = torch.cuda.Stream()
    cap.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(cap):
        a_step(0)
        b_step(0)
        tiny.add_(1)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=cap):
            if warm:
                side.wait_stream(cap)
            for i in range(pre):
                b_step(i)
                if warm:
                    with torch.cuda.stream(side):
                        tiny.add_(1)
            if fork:
                side.wait_stream(cap)
                first, second = (a_step, NA), (b_step, NB)
                if not small_on_side:
                    first, second = second, first
                with torch.cuda.stream(side):
                    for i in range(first[1]):
                        first[0](i)
                for i in range(second[1]):
                    second[0](i)
                cap.wait_stream(side)
            else:
                for i in range(NA):
                    a_step(i)
                for i in range(NB):
                    b_step(i)
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


for pre in (0, 40, 160, 320):
    base = timed(capture(pre, fork=False))
    print(f"PRE = {pre:3d} long kernels | everything on one stream {base:8.1f} us", flush=True)
    for label, kw in (("A (short) on the side stream", dict(small_on_side=True)), ("B (long) on the side stream", dict(small_on_side=False)),
                      ("A on the side stream, kept warm", dict(small_on_side=True, warm=True))):
        print(f"              {label:34s} {timed(capture(pre, **kw)):8.1f} us", flush=True)
