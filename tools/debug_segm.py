import sys, math, torch
sys.path.insert(0, "/root/repo")
import torch.nn.functional as F
from toist_amd import kernels as k, ops
dev = torch.device("cuda"); BF = torch.bfloat16
g = torch.Generator().manual_seed(0)
def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu(); return float((a - b).norm() / (b.norm() + 1e-12))
nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous()
# 1. conv dgrad/wgrad with C=8, Co=264 (kin % 64 != 0)
for (C, Co, H, W) in [(8, 264, 5, 6), (264, 128, 5, 6), (16, 8, 12, 12)]:
    Nb = 3
    x = torch.randn(Nb, C, H, W, generator=g).to(BF); w = (torch.randn(Co, C, 3, 3, generator=g) * 0.1).to(BF)
    xr, wr = x.float().requires_grad_(True), w.float().requires_grad_(True)
    y = F.conv2d(xr, wr, padding=1); dy = torch.randn(y.shape, generator=g).to(BF); y.backward(dy.float())
    out = ops.conv2d(nhwc(x).to(dev), nhwc(w).to(dev), pad=1)
    dx = ops.conv2d_dgrad(nhwc(dy).to(dev), nhwc(w).to(dev), (H, W), pad=1)
    dw = ops.conv2d_wgrad(nhwc(dy).to(dev), nhwc(x).to(dev), (Co, 3, 3, C), pad=1)
    print(f"C={C} Co={Co}: fwd {rel(out, nhwc(y.detach())):.4f} dgrad {rel(dx, nhwc(xr.grad)):.4f} wgrad {rel(dw, nhwc(wr.grad)):.4f}")
# 2. attention-map softmax fwd/bwd
B, Q, Hh, HW = 2, 6, 8, 30; ld = 32
s = (torch.randn(B, Q, Hh, ld, generator=g)).to(BF)
kp = torch.zeros(B, HW, dtype=torch.uint8); kp[1, 25:] = 1
prob = torch.empty(B * Q, HW, Hh, dtype=BF, device=dev)
k.attnmap_softmax_fwd(s.to(dev), kp.to(dev), B, Q, Hh, HW, ld, prob)
sr = s.float()[..., :HW].clone().requires_grad_(True)
pr = torch.softmax(sr.masked_fill(kp.bool()[:, None, None, :], float("-inf")), -1)
print("softmax fwd", rel(prob.view(B, Q, HW, Hh).permute(0, 1, 3, 2), pr.detach()))
gp = torch.randn(B, Q, Hh, HW, generator=g).to(BF)
pr.backward(gp.float())
ds = torch.empty(B, Q, Hh, ld, dtype=BF, device=dev)
k.attnmap_softmax_bwd(prob, gp.permute(0, 1, 3, 2).contiguous().view(B * Q, HW, Hh).to(dev), B * Q, Hh, HW, ld, ds)
print("softmax bwd", rel(ds[..., :HW], sr.grad), "pad", float(ds[..., HW:].abs().max()))
