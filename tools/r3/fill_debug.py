import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
import torch
from toist_amd import kernels as k, ops
BF = torch.bfloat16
dev = torch.device("cuda")
items = [(torch.randn(8, 16, 16, 256, device=dev).to(BF), torch.randn(8, 16, 16, 1024, device=dev).to(BF), torch.zeros(256, 1, 1, 1024, device=dev), torch.ones(256, device=dev)) for _ in range(9)]
def fn():
    ops.conv2d_wgrad_group(items, items[0][2].shape, accumulate=False)
    k.flush_reductions()
fn(); torch.cuda.synchronize()
ref = [it[2].clone() for it in items]
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(side), k.tables_beside_graph():
    fn()
    k._GROUP_TABLES.clear()
    with torch.cuda.graph(g, stream=side):
        fn()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
print("arena rows used:", [v[1] for v in k._TABLE_ARENAS.values()])
print("expected a ptrs", [hex(it[0].data_ptr()) for it in items[:2]])
for it in items: it[2].zero_()
g.replay(); torch.cuda.synchronize()
print("max diff after replay", max(float((it[2] - r).abs().max()) for it, r in zip(items, ref)))
