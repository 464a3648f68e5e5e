"""Round 3: the fused stem (csrc/stem.hip) against pack_image + implicit-GEMM conv + maxpool at the bench shape, us (hipGraph replay)."""
import sys, os, math
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
import torch
from toist_amd import kernels as k, ops
from tools.bench_gemm import timeit
BF = torch.bfloat16
dev = torch.device("cuda")
N, H, W = 8, 640, 640
img = torch.randn(N, 3, H, W, device=dev)
w8 = torch.nn.functional.pad(torch.randn(64, 7, 7, 3, device=dev) / math.sqrt(147), (0, 5)).to(BF).contiguous()
shift = torch.randn(64, device=dev) * 0.3
out = torch.empty(N, 160, 160, 64, dtype=BF, device=dev)
xin = torch.empty(N, H, W, 8, dtype=BF, device=dev)
ref = torch.empty_like(out)
def old():
    k.pack_image(img, xin)
    y = ops.conv2d(xin, w8, stride=2, pad=3, shift=shift, act=k.ACT_RELU, cin_real=3)
    k.maxpool3x3s2(y, ref)
def new():
    k.stem_fwd(img, w8, shift, out)
old(); new()
print("max diff", float((out.float() - ref.float()).abs().max()))
print("three launches %.1f us | fused %.1f us" % (timeit(old, 10) * 1000, timeit(new, 10) * 1000))
