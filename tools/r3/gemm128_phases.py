"""Round 3: shader cycles of prologue / k-loop / epilogue of gemm128_kernel (per wave, mean over the grid)."""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from toist_amd import kernels as k, ops
BF = torch.bfloat16
dev = torch.device("cuda")
ws = torch.zeros(1 << 18, device=dev)
for M, N, K in ((12800, 256, 1024), (4096, 4096, 4096), (3200, 1024, 2048)):
    x, w = torch.randn(M, K, device=dev).to(BF), (torch.randn(N, K, device=dev) / K ** 0.5).to(BF)
    wt = w.t().contiguous()
    res = torch.randn(M, N, device=dev).to(BF); aux = torch.randn(M, N, device=dev).to(BF)
    shift = torch.randn(N, device=dev)
    out = torch.empty(M, N, dtype=BF, device=dev)
    for name in ("fwd", "dgrad"):
        k.DEBUG_WS = ws
        ws.zero_()
        for _ in range(3):
            if name == "fwd":
                ops.linear(x, w, shift, out=out, act=k.ACT_RELU, tile=136, split_k=1)
            else:
                k.gemm(M, N, K, k.A_ROWK, k.operand(x, K), k.B_KROW, k.operand(wt, N), out, N, res=res, ldr=N, act=k.ACT_MASK_POS, aux=aux, ldaux=N, tile=136)
        torch.cuda.synchronize()
        k.DEBUG_WS = None
        r = ws.view(-1, 8)
        r = r[r[:, 4] > 0]
        print(f"{M} {N} {K} {name}: waves {len(r)} k-tiles {int(r[0,4])}  cycles: prologue {r[:,5].mean():7.0f}  loop {r[:,6].mean():8.0f} ({r[:,6].mean()/r[0,4]:5.0f} per k-tile)  epilogue {r[:,7].mean():7.0f} (max {r[:,7].max():7.0f})", flush=True)
