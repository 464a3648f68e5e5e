"""toist_amd -- MI355X-native (gfx950) implementation of the TOIST/MDETR forward/backward hot path.

Host code is Python on PyTorch-ROCm (memory, streams, torch.distributed only); all compute runs in
hand-written HIP kernels behind the C ABI of include/toist_hip.h.
"""
