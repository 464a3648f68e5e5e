"""Round 6: toist_lsap on the distillation step's problem sets: 24 softkd problems of ~95 x 95 (one launch) and the 1 x 1024
memory-bank replacements; us per launch (hipGraph replay)."""
import os
import sys
sys.path.insert(0, os.getcwd())
import torch
from toist_amd import matcher as tm
from tools.bench_gemm import timeit
dev = torch.device("cuda")
g = torch.Generator().manual_seed(1)
for name, shapes in (("24 softkd problems (Q - c)^2, c = 0 .. 10", [(100 - (i * 7) % 11,) * 2 for i in range(24)]), ("24 x 100 x 100", [(100, 100)] * 24),
                     ("4 bank updates 1 x 1024", [(1, 1024)] * 4), ("1 x (3 x 1024)", [(3, 1024)]), ("48 matcher-like 10 x 100", [(10, 100)] * 48)):
    costs = [torch.rand(r, c, generator=g) * 3 - 1 for r, c in shapes]
    sizes = [r * c for r, c in shapes]
    flat = torch.cat([m.reshape(-1) for m in costs]).to(dev)
    offs = [sum(sizes[:i]) for i in range(len(sizes))]
    f = lambda: tm.lsap_blocks(flat, shapes, offs, 0)
    f()
    print(f"{name:45s} {timeit(f, 10) * 1000:9.1f} us per launch", flush=True)
