"""Combine the FETCH_SIZE / WRITE_SIZE per-kernel summaries (tools/pmc_summary.py) into per-launch HBM traffic.

usage: python tools/pmc_traffic.py <fetch_summary.txt> <write_summary.txt> <out.json>
Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): rocprofv3 reports both counters in
KiB; on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes, so wide coalesced reads (all of these kernels load
16 B per lane) are doubled; WRITE_SIZE is taken as reported (uncalibrated)."""
import json
import sys


def read(path):
    out = {}
    for line in open(path):
        if line.startswith("#") or not line.strip():
            continue
        name, n, total, mean = line.rstrip("\n").split("\t")
        out[name] = (int(n), float(mean))
    return out


fetch, write = read(sys.argv[1]), read(sys.argv[2])
res = {}
for name, (n, f_kib) in fetch.items():
    w_kib = write.get(name, (0, 0.0))[1]
    res[name] = {"dispatches": n, "fetch_kib_raw_mean": round(f_kib, 1), "write_kib_raw_mean": round(w_kib, 1),
                 "hbm_bytes_per_launch": int(2 * f_kib * 1024 + w_kib * 1024)}
import subprocess
try:
    commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
except Exception:
    commit = None
json.dump({"commit": sys.argv[4] if len(sys.argv) > 4 else commit, "note": "per-launch means over an eager (no-graph) run of bench.py; hbm_bytes = 2*FETCH_SIZE + WRITE_SIZE (KiB -> bytes), "
                   "FETCH doubled per the gfx950 correction of MI355X_MICROARCH.md", "kernels": res}, open(sys.argv[3], "w"), indent=1)
print("wrote", sys.argv[3], len(res), "kernels")
