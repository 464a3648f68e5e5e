"""Round 3 experiment (KNOBS=1 build): ablations of gemm128w_kernel's k-loop on the grouped layer-3 weight gradients -- flags bits 12..14:
1 = no MFMAs, 2 = no fragment reads, 4 = no DMA inside the loop (results are then garbage; only the time is read)."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
import torch
from toist_amd import kernels as k, ops
from tools.bench_gemm import timeit
BF = torch.bfloat16
dev = torch.device("cuda")
k.DEBUG_WS = torch.zeros(1 << 20, device=dev)
for Nb, H, W, C, Co, R, n in ((8, 40, 40, 1024, 256, 1, 22), (8, 40, 40, 256, 256, 3, 22), (8, 80, 80, 128, 128, 3, 3)):
    items = [(torch.randn(Nb, H, W, Co, device=dev).to(BF), torch.randn(Nb, H, W, C, device=dev).to(BF), torch.zeros(Co, R, R, C, device=dev), torch.ones(Co, device=dev)) for _ in range(n)]
    row = []
    for abl in (0, 1, 2, 4, 3, 6, 5, 7):
        k.DEBUG_FLAGS = abl << 12
        t = timeit(lambda: ops.conv2d_wgrad_group(items, items[0][2].shape, pad=R // 2), 10) * 1000
        row.append(f"{abl}:{t:6.1f}")
    print(f"{n} x [{Nb}x{H}x{W}] C{C}->{Co} {R}x{R}  us by ablation  " + "  ".join(row), flush=True)
