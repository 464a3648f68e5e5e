"""Per-launch time of the three fused mask-head stages (csrc/maskstage.hip) at configs[2]'s shapes (B = 8, Q = 100 -> 800 maps, 640 x 640 images:
lay4 at 80 x 80, lay5 / out_lay at 160 x 160) against their algorithmic HBM bytes (one read of the source + the FPN term, one write of the output).
usage (GPU box): python tools/r5/maskstage_bench.py [--maps 800] [--iters 20]"""
import argparse
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from toist_amd import kernels as k

ap = argparse.ArgumentParser()
ap.add_argument("--maps", type=int, default=800)
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda:0")
BF = torch.bfloat16
N, Q, B = a.maps, 100, max(1, a.maps // 100)
g = torch.Generator(device="cpu").manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g)


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.iters * 1e3   # us


rows = []
# lay4: a3 [N,40,40,64] (normalised) + fpn [B,80,80,64] -> pre4 [N,80,80,32]
a3 = rnd(N, 40, 40, 64).clamp(min=0).to(BF).to(dev)
f3 = rnd(B, 80, 80, 32).to(BF).to(dev)      # lay4(adapter2(fpn)) of the image, bias included
w4 = (rnd(32, 3, 3, 64) * 0.05).to(BF).to(dev)
b4 = rnd(32).to(dev)
pre4 = torch.empty(N, 80, 80, 32, dtype=BF, device=dev)
st4 = torch.empty(N, 8, 2, dtype=torch.float32, device=dev)
us = timed(lambda: k.mask_stage_fwd(a3, None, None, None, f3, w4, None, pre4, st4, N, Q, 80, 80, 64, 32, 32, False, True))
rows.append(("lay4  64->32 @ 80x80   (up, + FPN conv)", us, a3.numel() * 2 + f3.numel() * 2 + pre4.numel() * 2, 2 * N * 6400 * 64 * 32 * 9))
# lay5: pre4 (+ GN) + fpn [B,160,160,32] -> pre5 [N,160,160,16]
f2 = rnd(B, 160, 160, 16).to(BF).to(dev)
w5 = (rnd(16, 3, 3, 32) * 0.05).to(BF).to(dev)
b5 = rnd(16).to(dev)
g4w, g4b = (1 + 0.1 * rnd(32)).to(dev), (0.1 * rnd(32)).to(dev)
pre5 = torch.empty(N, 160, 160, 16, dtype=BF, device=dev)
st5 = torch.empty(N, 8, 2, dtype=torch.float32, device=dev)
us = timed(lambda: k.mask_stage_fwd(pre4, st4, g4w, g4b, f2, w5, None, pre5, st5, N, Q, 160, 160, 32, 16, 16, True, True))
rows.append(("lay5  32->16 @ 160x160 (GN + up, + FPN conv)", us, pre4.numel() * 2 + f2.numel() * 2 + pre5.numel() * 2, 2 * N * 25600 * 32 * 16 * 9))
# out_lay: pre5 (+ GN) -> logits f32 [N,160,160]
wo = (rnd(1, 3, 3, 16) * 0.05).to(BF).to(dev)
bo = rnd(1).to(dev)
g5w, g5b = (1 + 0.1 * rnd(16)).to(dev), (0.1 * rnd(16)).to(dev)
logits = torch.empty(N, 160, 160, dtype=torch.float32, device=dev)
us = timed(lambda: k.mask_stage_fwd(pre5, st5, g5w, g5b, None, wo, bo, logits, None, N, Q, 160, 160, 16, 1, 1, True, False))
rows.append(("out_lay 16->1 @ 160x160 (GN)", us, pre5.numel() * 2 + logits.numel() * 4, 2 * N * 25600 * 16 * 1 * 9))
print(f"# mask_stage_kernel, {N} maps; algorithmic bytes = source + FPN term + output, once each; HBM peak 8 TB/s")
for name, us, byt, fl in rows:
    print(f"{name:42s} {us:8.1f} us  {byt / 1e6:8.1f} MB  {byt / us / 1e6:6.2f} TB/s = {byt / us / 1e6 / 8 * 100:4.1f} % of the HBM peak   ({fl / us / 1e6:6.1f} TFLOP/s)")
