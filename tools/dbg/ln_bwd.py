import sys, os
sys.path.insert(0, os.getcwd())
import torch
from toist_amd import kernels as k
from tools.bench_gemm import timeit
import inspect
BF = torch.bfloat16
dev = torch.device("cuda")
print(inspect.signature(k.layernorm_bwd))
for rows, D in ((3328, 256), (800, 256), (128, 768), (13312, 256)):
    x = torch.randn(rows, D, device=dev).to(BF); dy = torch.randn(rows, D, device=dev).to(BF)
    g = torch.randn(D, device=dev); b = torch.randn(D, device=dev)
    y = torch.empty_like(x); mean = torch.empty(rows, device=dev); rstd = torch.empty(rows, device=dev)
    k.layernorm_fwd(x, g, b, 1e-5, y, mean, rstd)
    dx = torch.empty_like(x); dg = torch.zeros(D, device=dev); db = torch.zeros(D, device=dev)
    t = timeit(lambda: k.layernorm_bwd(dy, x, mean, rstd, g, dx, dg, db), 20) * 1000
    t2 = timeit(lambda: k.layernorm_bwd(dy, x, mean, rstd, g, dx, None, None), 20) * 1000
    def deferred():
        for _ in range(6):
            k.layernorm_bwd(dy, x, mean, rstd, g, dx, dg, db, defer=True)
        k.flush_reductions()
    t3 = timeit(deferred, 5) * 1000 / 6
    dg.zero_(); db.zero_(); k.layernorm_bwd(dy, x, mean, rstd, g, dx, dg, db); a = (dg.clone(), db.clone(), dx.clone())
    dg.zero_(); db.zero_(); k.layernorm_bwd(dy, x, mean, rstd, g, dx, dg, db, defer=True); k.flush_reductions()
    err = max(float((dg - a[0]).abs().max() / a[0].abs().max()), float((db - a[1]).abs().max() / a[1].abs().max()), float((dx.float() - a[2].float()).abs().max()))
    print(f"{rows}x{D}: atomics {t:.1f} us | without dgamma/dbeta {t2:.1f} us | partials + batched fold (6 calls per fold) {t3:.1f} us | rel diff {err:.1e}", flush=True)
