"""Round 3: shader cycles per tile phase of panel2_kernel (wave 0 of every workgroup): tile-top wait + barrier | DMA issue | LDS reads + MFMA | epilogue."""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from toist_amd import kernels as k, ops
BF = torch.bfloat16
dev = torch.device("cuda")
ws = torch.zeros(1 << 16, device=dev)
for M, N, K in ((12800, 1024, 256), (51200, 512, 128), (204800, 256, 64), (3328, 2048, 256)):
    x, w = torch.randn(M, K, device=dev).to(BF), (torch.randn(N, K, device=dev) * 0.05).to(BF)
    res = torch.randn(M, N, device=dev).to(BF)
    shift = torch.randn(N, device=dev)
    out = torch.empty(M, N, dtype=BF, device=dev)
    for with_res in (True, False):
        for v in (2, 3, 18, 19):
            k.DEBUG_WS = ws
            for per_cu in ("2", "3"):
                pass
            ws.zero_()
            for _ in range(3):
                ops.linear(x, w, shift, out=out, res=res if with_res else None, act=k.ACT_RELU, tile=135 | (v << 8))
            torch.cuda.synchronize()
            k.DEBUG_WS = None
            r = ws.view(-1, 8)
            r = r[r[:, 4] > 0]
            if len(r) == 0:
                print(M, N, K, with_res, v, "no data"); continue
            T = r[:, 4].mean().item()
            ph = (r[:, :4].sum(0) / r[:, 4].sum()).tolist()
            print(f"{M:7d} {N:5d} {K:4d} res={int(with_res)} v{v:2d}  wgs {len(r):4d} tiles/wg {T:5.1f}  cycles/tile: wait {ph[0]:7.0f} issue {ph[1]:7.0f} mfma {ph[2]:7.0f} epi {ph[3]:7.0f}  sum {sum(ph):7.0f}", flush=True)
