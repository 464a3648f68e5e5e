import sys, torch
sys.path.insert(0, "/root/repo")
from oracle import model_ref
from toist_amd.segmentation import DETRsegm
dev = torch.device("cuda"); BF = torch.bfloat16
def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu(); return float((a - b).norm() / (b.norm() + 1e-12))
torch.manual_seed(0)
B, Q, d, H, h, w = 2, 6, 256, 8, 5, 6
class Stub(torch.nn.Module):
    def __init__(self):
        super().__init__(); self.transformer = type("T", (), {"d_model": d, "nhead": H})()
seg = DETRsegm(Stub(), "smallconv")
g = torch.Generator().manual_seed(1)
sd = {k_: v.detach().clone().float().requires_grad_(True) for k_, v in seg.state_dict().items()}
seg.to(dev); seg._debug = {}
hs = torch.randn(B, Q, d, generator=g).to(BF); mem = torch.randn(B, h*w, d, generator=g).to(BF); src = torch.randn(B, h*w, d, generator=g).to(BF)
c4 = torch.randn(B, 2*h, 2*w, 1024, generator=g).clamp(min=0).to(BF); c3 = torch.randn(B, 4*h, 4*w, 512, generator=g).clamp(min=0).to(BF); c2 = torch.randn(B, 8*h, 8*w, 256, generator=g).clamp(min=0).to(BF)
fmask = torch.zeros(B, h, w, dtype=torch.bool); fmask[1, :, 4:] = True
ins = [t.to(dev).requires_grad_(True) for t in (hs.view(B*Q, d), mem.view(B*h*w, d), src.view(B*h*w, d), c4, c3, c2)]
masks = seg._masks(*ins, fmask.to(dev), B, Q, h, w)
gout = torch.randn(masks.shape, generator=g) * 0.1
masks.backward(gout.to(dev)); torch.cuda.synchronize()
nchw = lambda t: t.float().permute(0, 3, 1, 2)
bm = model_ref.attention_map(sd, "bbox_attention.", hs.float(), mem.float().transpose(1, 2).reshape(B, d, h, w), fmask, H)
bm.retain_grad()
ref = model_ref.mask_head(sd, "mask_head.", src.float().transpose(1, 2).reshape(B, d, h, w), bm, [nchw(c4), nchw(c3), nchw(c2)]).view(B, Q, 8*h, 8*w)
ref.backward(gout)
import torch.nn.functional as F
qr = F.linear(hs.float(), sd["bbox_attention.q_linear.weight"], sd["bbox_attention.q_linear.bias"]).detach()
kr = F.linear(mem.float(), sd["bbox_attention.k_linear.weight"], sd["bbox_attention.k_linear.bias"]).detach()
print("q", rel(seg._debug["q"].view(B, Q, d), qr), "kk", rel(seg._debug["kk"].view(B, h*w, d), kr))
sref = torch.einsum("bqnc,bpnc->bqnp", qr.view(B, Q, H, d // H) * (d // H) ** -0.5, kr.view(B, h*w, H, d // H))
print("scores", rel(seg._debug["scores"][..., :h*w], sref))
prob = seg._debug["pv"].data.view(B, Q, h, w, H).permute(0, 1, 4, 2, 3)
print("prob", rel(prob, bm.detach()), "masks", rel(masks, ref))
dprob = seg._debug["dprob"].view(B, Q, h, w, H).permute(0, 1, 4, 2, 3)
print("dprob", rel(dprob, bm.grad), float(dprob.float().norm()), float(bm.grad.norm()))
for n in ["bbox_attention.q_linear.weight", "mask_head.lay1.weight"]:
    gp = dict(seg.named_parameters())[n].grad
    print(n, rel(gp, sd[n].grad))
w1 = dict(seg.named_parameters())["mask_head.lay1.weight"].grad
print("lay1 slice q-part", rel(w1[:, 256:], sd["mask_head.lay1.weight"].grad[:, 256:]), "img-part", rel(w1[:, :256], sd["mask_head.lay1.weight"].grad[:, :256]))
