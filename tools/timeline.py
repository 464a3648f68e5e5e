"""Timeline analysis of one graph-replayed training step from a rocprofv3 kernel trace.

usage: python tools/timeline.py <kernel_trace.csv> [out.txt]
Steps are delimited by toist::matcher_kernel (one launch per step); the last complete interval is analysed:
wall time, GPU-busy time (union of kernel intervals, so parallel graph branches are not double counted), idle
gaps, and per-kernel totals."""
import collections
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"void toist::gemm_kernel<([^>]*)>", name)
    if m:
        return "gemm<" + m.group(1).replace(" ", "") + ">"
    m = re.search(r"toist::(\w+)", name)
    if m:
        return m.group(1)
    m = re.search(r"multi_tensor_apply_kernel<at::native::(\w+)<(\d+)>, at::native::(\w+)", name)
    if m:
        return f"multi_tensor[{m.group(3)}]"
    m = re.search(r"at::native::(\w+)<[^,]*, at::native::(\w+)", name)
    if m:
        return f"{m.group(1)}[{m.group(2)}]"
    return name[:60]


rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
marks = [i for i, r in enumerate(rows) if "matcher_kernel" in r[2]]
assert len(marks) >= 3, "need at least 3 steps in the trace"
lo, hi = marks[-3], marks[-2]
step = rows[lo:hi]
t0, t1 = step[0][0], rows[hi][0]
busy, cur_s, cur_e = 0, None, None
for s, e, _ in step:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - s for s, e, _ in step)
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
print(f"step wall {1e-6 * (t1 - t0):.3f} ms | kernels {len(step)} | sum of kernel durations {1e-6 * tot:.3f} ms | "
      f"GPU busy (union) {1e-6 * busy:.3f} ms | idle {1e-6 * (t1 - t0 - busy):.3f} ms", file=out)
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n in step:
    a = agg[short(n)]
    a[0] += e - s
    a[1] += 1
print(f"{'kernel':70s} {'calls':>6s} {'total ms':>9s} {'avg us':>8s} {'%':>6s}", file=out)
for n, (d, c) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print(f"{n:70s} {c:6d} {1e-6 * d:9.3f} {1e-3 * d / c:8.1f} {100.0 * d / tot:6.1f}", file=out)
bk = collections.Counter()
for s, e, _ in step:
    us = (e - s) / 1e3
    bk["<4us" if us < 4 else "4-8us" if us < 8 else "8-16us" if us < 16 else "16-32us" if us < 32 else "32-64us" if us < 64 else ">=64us"] += 1
print("duration histogram:", dict(bk), file=out)
