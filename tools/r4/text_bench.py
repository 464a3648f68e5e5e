"""us per launch (hipGraph replay) of the text encoder's operators at the benchmark's shape (8 captions x 16 tokens = 128 rows, d = 768,
FFN 3072): the general GEMM (ops.linear / ops.linear_dgrad, split-K where the dispatcher picks it) and the LayerNorm launches.
(A column-block variant of csrc/tlayer.hip's row kernel was tried for these shapes and dropped: 8.7 / 9.2 / 12.9 us against 7.9 / 8.1 /
11.4 us of the general GEMM for QKV / out-proj / FFN1, 17.3 and 27.5 us against 12.5 and 12.9 for the deep data gradients.)"""
import math
import sys

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tools/r4")
from toist_amd import kernels as k, ops  # noqa: E402
from rowgemm_bench import timed  # noqa: E402

dev = torch.device("cuda")
BF = torch.bfloat16


def check(name, got, ref, tol=2e-2):
    rel = float((got.float() - ref.float()).norm() / (ref.float().norm() + 1e-20))
    print(f"    check {name}: rel {rel:.2e}", "OK" if rel < tol else "MISMATCH", flush=True)


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    D, F = 768, 3072
    rn = lambda *s: torch.randn(*s, device=dev)
    x = rn(M, D).to(BF)
    res = rn(M, D).to(BF)
    for name, N, K, act in (("qkv 2304 <- 768", 3 * D, D, k.ACT_NONE), ("out-proj 768 <- 768 (+res, dropout)", D, D, k.ACT_NONE), ("ffn1 3072 <- 768 GELU", F, D, k.ACT_GELU),
                            ("ffn2 768 <- 3072 (+res, dropout)", D, F, k.ACT_NONE)):
        a = rn(M, K).to(BF)
        w = (rn(N, K) / math.sqrt(K)).to(BF)
        b = rn(N)
        out, out2, pre, pre2 = (torch.empty(M, N, dtype=BF, device=dev) for _ in range(4))
        r = res if N == D else None
        dp = 0.1 if N == D else 0.0
        print(f"--- forward {name}")
        timed(lambda: ops.linear(a, w, b, res=r, act=act, pre_out=pre if act else None, drop_where=1 if dp else 0, drop_p=dp, drop_seed=7, out=out), "ops.linear")
        # data gradient: dx[M, K] = dy[M, N] @ w[N, K]
        dy = rn(M, N).to(BF)
        dx, dx2 = torch.empty(M, K, dtype=BF, device=dev), torch.empty(M, K, dtype=BF, device=dev)
        aux = rn(M, K).to(BF)
        bact = k.ACT_GELU_BWD if K == F else k.ACT_NONE
        print(f"--- dgrad of {name}: {K} <- {N}")
        timed(lambda: ops.linear_dgrad(dy, w, out=dx, act=bact, aux=aux if bact else None), "ops.linear_dgrad")
    print("--- LayerNorm 768")
    gamma, beta = torch.rand(D, device=dev) + 0.5, rn(D)
    y, dxx, dxd = (torch.empty(M, D, dtype=BF, device=dev) for _ in range(3))
    mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
    dg, db = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
    timed(lambda: k.layernorm_fwd(x, gamma, beta, 1e-5, y, mean, rstd), "layernorm_fwd")
    timed(lambda: k.layernorm_bwd(res, x, mean, rstd, gamma, dxx, dg, db, defer=True), "layernorm_bwd (deferred parameter gradients)")
    k.flush_reductions()
    timed(lambda: k.layernorm_bwd(res, x, mean, rstd, gamma, dxx, dg, db, defer=False), "layernorm_bwd (atomics)")
    timed(lambda: k.layernorm_bwd(res, x, mean, rstd, gamma, dxx, None, None), "layernorm_bwd (no parameter gradients)")
    timed(lambda: k.layernorm_bwd(res, x, mean, rstd, gamma, dxx, dg, db, dx_drop=dxd, drop_p=0.1, seed=3, defer=False), "layernorm_bwd (atomics, masked copy)")
    timed(lambda: k.flush_reductions(), "flush_reductions (empty)")


if __name__ == "__main__":
    main()
