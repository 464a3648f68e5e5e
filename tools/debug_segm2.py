import sys, math, torch
sys.path.insert(0, "/root/repo")
from toist_amd import kernels as k, ops
dev = torch.device("cuda"); BF = torch.bfloat16
g = torch.Generator().manual_seed(0)
def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu(); return float((a - b).norm() / (b.norm() + 1e-12))
B, Q, d, H, h, w = 2, 6, 256, 8, 5, 6
HW, dh, ld = h * w, d // H, 32
scale = dh ** -0.5
q = torch.randn(B * Q, d, generator=g).to(BF); kk = torch.randn(B * HW, d, generator=g).to(BF)
qd, kd = q.to(dev), kk.to(dev)
scores = torch.empty(B, Q, H, ld, dtype=BF, device=dev)
k.gemm(Q, HW, dh, k.A_ROWK, k.operand(qd, d, bs_outer=Q * d, bs_inner=dh), k.B_ROWK, k.operand(kd, d, bs_outer=HW * d, bs_inner=dh),
       scores, H * ld, batch=B * H, batch_inner=H, cs_outer=Q * H * ld, cs_inner=ld, alpha=scale, tile=64)
qr = q.float().view(B, Q, H, dh).requires_grad_(True); kr = kk.float().view(B, HW, H, dh).requires_grad_(True)
sref = torch.einsum("bqnc,bpnc->bqnp", qr * scale, kr)
print("scores", rel(scores[..., :HW], sref.detach()))
ds = torch.zeros(B, Q, H, ld); ds[..., :HW] = torch.randn(B, Q, H, HW, generator=g)
ds = ds.to(BF)
sref.backward(ds[..., :HW].float())
dsd = ds.to(dev)
dq = torch.empty(B * Q, d, dtype=BF, device=dev); dk = torch.empty(B * HW, d, dtype=BF, device=dev)
k.gemm(Q, dh, HW, k.A_ROWK, k.operand(dsd, H * ld, bs_outer=Q * H * ld, bs_inner=ld), k.B_KROW,
       k.operand(kd, d, bs_outer=HW * d, bs_inner=dh), dq, d, batch=B * H, batch_inner=H, cs_outer=Q * d, cs_inner=dh, alpha=scale, tile=64)
k.gemm(HW, dh, Q, k.A_KROW, k.operand(dsd, H * ld, bs_outer=Q * H * ld, bs_inner=ld), k.B_KROW,
       k.operand(qd, d, bs_outer=Q * d, bs_inner=dh), dk, d, batch=B * H, batch_inner=H, cs_outer=HW * d, cs_inner=dh, alpha=scale, tile=64)
print("dq", rel(dq, qr.grad.reshape(B * Q, d)), "dk", rel(dk, kr.grad.reshape(B * HW, d)))
