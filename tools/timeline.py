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
    # every template family is ONE row (its instantiations are summed), like gemm128_kernel / panel2_kernel: round 4 listed the 15 instantiations
    # of the generic tile separately, which hid that the family was the largest of the step
    m = re.match(r"void toist::gemm_kernel<(\d+), (\d+), (\d+),", name)
    if m:
        return "gemm_kernel<%s,%s,%s,...>" % m.groups()
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
META = {}
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
        META[(int(r["Start_Timestamp"]), r["Kernel_Name"])] = (r.get("Queue_Id", "?"), r.get("Stream_Id", "?"), r.get("Grid_Size_X", "?"), r.get("Workgroup_Size_X", "?"))
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

# phases of the replayed step (matcher -> ... -> next matcher), delimited by kernels that occur once per step
def first(pred, start=0):
    for i in range(start, len(step)):
        if pred(step[i][2]):
            return i
    return None


conv_like = lambda n: "conv3_kernel" in n or re.search(r"gemm_kernel<\d+, \d+, \d+, [23],", n) is not None or re.search(r"gemm_kernel<\d+, \d+, \d+, \d, 2,", n) is not None
i_bb_bwd = first(conv_like)
i_opt = first(lambda n: "sqnorm_kernel" in n)
i_opt_end = first(lambda n: "adamw_ema_kernel" in n)
i_pos = first(lambda n: "sine_pos" in n, i_opt_end or 0)
if None not in (i_bb_bwd, i_opt, i_opt_end, i_pos):
    cuts = [("criterion + transformer backward", 0, i_bb_bwd), ("backbone backward (+ text branch)", i_bb_bwd, i_opt),
            ("optimizer tail", i_opt, i_opt_end + 1), ("stem + backbone forward (+ text branch)", i_opt_end + 1, i_pos),
            ("transformer forward + heads", i_pos, len(step))]
    print("phases (wall between first kernels):", file=out)
    for name, a, b in cuts:
        if b <= a:
            continue
        w0 = step[a][0]
        w1 = step[b][0] if b < len(step) else t1
        print(f"  {name:45s} {1e-6 * (w1 - w0):7.3f} ms  {b - a:5d} kernels  sum {1e-6 * sum(e - s for s, e, _ in step[a:b]):7.3f} ms", file=out)
if len(sys.argv) > 3:                      # optional: dump the kernel sequence (name, us) of the analysed step
    with open(sys.argv[3], "w") as f:
        for s, e, n in step:
            qid, sid, gx, wx = META.get((s, n), ("?", "?", "?", "?"))
            f.write(f"{1e-3 * (s - t0):9.1f} {1e-3 * (e - s):7.1f} q{qid} s{sid} g{gx}/{wx} {short(n)}\n")
