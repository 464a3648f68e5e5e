"""What the memory system gives elementwise traffic of the size of one 1x1-conv epilogue (12800 x 1024 bf16 = 26 MB per stream)."""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from toist_amd import kernels as k
from tools.bench_gemm import timeit
BF = torch.bfloat16
dev = torch.device("cuda")
for M, N in ((12800, 1024), (51200, 512), (204800, 256), (12800, 256)):
    a, b, c = (torch.randn(M, N, device=dev).to(BF) for _ in range(3))
    bufs = [torch.randn(M, N, device=dev).to(BF) for _ in range(12)]     # 12 x 26 MB > the 256 MB Infinity Cache when rotated
    t_add = timeit(lambda: k.add(a, b, c), 20) * 1000
    t_copy = timeit(lambda: c.copy_(a), 20) * 1000
    t_tadd = timeit(lambda: torch.add(a, b, out=c), 20) * 1000
    i = [0]
    def rot():
        j = i[0]; i[0] = (j + 3) % 12
        k.add(bufs[j], bufs[j + 1], bufs[j + 2])
    t_rot = timeit(rot, 24) * 1000
    mb = M * N * 2 / 1e6
    print(f"{M}x{N} ({mb:.0f} MB/stream): k.add {t_add:.1f} us = {3 * mb / t_add:.2f} TB/s | rotating buffers {t_rot:.1f} us = {3 * mb / t_rot:.2f} TB/s | "
          f"torch.add {t_tadd:.1f} us | copy_ {t_copy:.1f} us = {2 * mb / t_copy:.2f} TB/s", flush=True)
