"""us per launch (hipGraph replay of 20 back-to-back launches) of the row-complete sub-layer kernels (csrc/tlayer.hip) against the
launches they replace (GEMM with the complete epilogue + stand-alone LayerNorm), at the encoder (3328 rows) and decoder (800 rows) shapes."""
import math
import sys

import torch

sys.path.insert(0, ".")
from toist_amd import kernels as k, ops  # noqa: E402

dev = torch.device("cuda")
BF = torch.bfloat16
REP = 20


def timed(fn, label):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(REP):
                fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1000 / REP)
    print("%-72s %7.2f us" % (label, best), flush=True)
    return best


def main():
    g = torch.Generator().manual_seed(0)
    gamma, beta, bias = torch.rand(256, device=dev) + 0.5, torch.randn(256, device=dev), torch.randn(256, device=dev)
    for M in (3328, 800):
        res = torch.randn(M, 256, device=dev).to(BF)
        add = torch.randn(M, 256, device=dev).to(BF)
        z, y, y2 = (torch.empty(M, 256, dtype=BF, device=dev) for _ in range(3))
        mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
        dgamma, dbeta = torch.zeros(256, device=dev), torch.zeros(256, device=dev)
        for K in (256, 768, 2048):
            a = torch.randn(M, K, device=dev).to(BF)
            wf = (torch.randn(256, K, device=dev) / math.sqrt(K)).to(BF)      # forward weight [256][K]
            wb = (torch.randn(K, 256, device=dev) / math.sqrt(K)).to(BF)      # dgrad reads [K][256]
            print(f"--- M = {M}, K = {K}")
            timed(lambda: k.rowgemm(a, wf, y, b_kind=k.B_ROWK, epi=k.ROW_LN_FWD, bias=bias, res=res, gamma=gamma, beta=beta, z=z, mean=mean, rstd=rstd,
                                    add=add, out2=y2, drop_p=0.1, drop_seed=5), "fused  linear + dropout + residual + LayerNorm (+ add)")

            def unfused_fwd():
                zz = ops.linear(a, wf, bias, res=res, drop_where=1, drop_p=0.1, drop_seed=5, out=z)
                k.layernorm_fwd(zz, gamma, beta, 1e-5, y, mean, rstd, add=add, y2=y2)
            timed(unfused_fwd, "       ops.linear(res, dropout) + layernorm_fwd")
            timed(lambda: ops.linear(a, wf, bias, res=res, drop_where=1, drop_p=0.1, drop_seed=5, out=z), "       ops.linear(res, dropout) alone")
            timed(lambda: k.rowgemm(a, wf, y, b_kind=k.B_ROWK, epi=k.ROW_PLAIN, bias=bias, res=res), "fused  plain epilogue (forward weights)")
            dz, dzd = torch.empty(M, 256, dtype=BF, device=dev), torch.empty(M, 256, dtype=BF, device=dev)

            def fused_bwd():
                k.rowgemm(a, wb, dz, b_kind=k.B_KROW, epi=k.ROW_LN_BWD, res=res, gamma=gamma, z=z, mean=mean, rstd=rstd, out2=dzd, drop_p=0.1, drop_seed=5,
                          dgamma=dgamma, dbeta=dbeta)
            timed(fused_bwd, "fused  dgrad + residual + LayerNorm backward + mask")
            gbuf = torch.empty(M, 256, dtype=BF, device=dev)

            def unfused_bwd():
                ops.linear_dgrad(a, wb, out=gbuf, res=res)
                k.layernorm_bwd(gbuf, z, mean, rstd, gamma, dz, dgamma, dbeta, dx_drop=dzd, drop_p=0.1, seed=5, defer=True)
            timed(unfused_bwd, "       ops.linear_dgrad(res) + layernorm_bwd")
            timed(lambda: ops.linear_dgrad(a, wb, out=gbuf, res=res), "       ops.linear_dgrad(res) alone")
            timed(lambda: k.rowgemm(a, wb, dz, b_kind=k.B_KROW, epi=k.ROW_PLAIN, res=res), "fused  plain epilogue (k-major weights)")
            k.flush_reductions()


if __name__ == "__main__":
    main()
