"""us per launch (hipGraph replay of 20 back-to-back launches) of the attention cores: (round 4 also timed the first generation, csrc/attn.hip, removed in round 5: forward, query-major
backward with 4 query splits + fold) against second generation (csrc/attn2.hip) at the three shapes of the bench model, dropout 0.1."""
import math
import sys

import torch

sys.path.insert(0, ".")
from toist_amd import kernels as k  # noqa: E402
from tools.r4.rowgemm_bench import timed  # noqa: E402

dev = torch.device("cuda")
BF = torch.bfloat16
B, H, dh, d = 8, 8, 32, 256
k.SEED_DEV = torch.zeros(1, dtype=torch.int64, device=dev)
drop = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
for Sq, Sk in [(416, 416), (100, 416), (100, 100), (1066, 1066)]:
    q = torch.randn(B * Sq, 3 * d, device=dev).to(BF)
    kv = torch.randn(B * Sk, 3 * d, device=dev).to(BF)
    qs, ks, vs = q[:, :d], kv[:, d:2 * d], kv[:, 2 * d:]
    pad = torch.zeros(B, Sk, dtype=torch.uint8, device=dev)
    ctx = torch.empty(B * Sq, d, dtype=BF, device=dev)
    dctx = torch.randn(B * Sq, d, device=dev).to(BF)
    lse = torch.empty(B * H, Sq, 2, device=dev)
    dqkv = torch.empty(B * Sq, 3 * d, dtype=BF, device=dev)
    dkv = torch.empty(B * Sk, 3 * d, dtype=BF, device=dev)
    sc = 1 / math.sqrt(dh)
    print(f"--- Sq = {Sq}, Sk = {Sk}, dropout {drop}")
    timed(lambda: k.attn2_fwd(qs, ks, vs, pad, B, H, Sq, Sk, dh, sc, drop, 7, ctx, lse), "attn2 forward")
    splits = k.attn2_splits(Sk)
    part = torch.empty(splits, B * Sq, d, dtype=BF, device=dev) if splits > 1 else None
    timed(lambda: k.attn2_bwd(qs, ks, vs, ctx, dctx, lse, pad, B, H, Sq, Sk, dh, sc, drop, 7, dqkv[:, :d] if splits == 1 else None, dkv[:, d:2 * d], dkv[:, 2 * d:],
                              dq_part=part), f"attn2 backward ({splits} key splits, dQ shares folded by the consumer)")
    flop = B * 4.0 * Sq * Sk * d
    print(f"    core FLOP forward {flop / 1e9:.2f} G, backward {2.5 * flop / 1e9:.2f} G")
