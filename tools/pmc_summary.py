"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes per kernel family.
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x
(MI355X_MICROARCH.md section HBM), so reads are doubled.  usage: pmc_summary.py fetch.csv write.csv"""
import csv
import sys
from collections import defaultdict


def load(path, name):
    tot = defaultdict(lambda: [0.0, 0])
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != name:
                continue
            k = r["Kernel_Name"]
            fam = ("gemm" if ("gemm_kernel" in k or "gemm_big" in k) else "attn_fwd" if "attn_fwd" in k else "attn_bwd" if "attn_bwd" in k
                   else "splitk_reduce" if "splitk_reduce" in k else "other")
            tot[fam][0] += float(r["Counter_Value"])
            tot[fam][1] += 1
    return tot


fetch = load(sys.argv[1], "FETCH_SIZE")
write = load(sys.argv[2], "WRITE_SIZE")
import json
summary = {"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python bench.py --steps 2 --warmup 1 "
                      "--no-cpu-baseline --no-kernel-timing (two separate passes)",
           "correction": "FETCH_SIZE, WRITE_SIZE in KiB; FETCH_SIZE doubled (gfx950 reports 1/2 of wide coalesced reads, "
                         "MI355X_MICROARCH.md section HBM)", "families": {}}
for fam in sorted(set(fetch) | set(write)):
    fkb, n = fetch.get(fam, [0.0, 0])
    wkb, n2 = write.get(fam, [0.0, 0])
    n = max(n, n2, 1)
    rd = 2.0 * fkb * 1024
    wr = wkb * 1024
    print("%-14s launches %6d  read %.3f GB (x2-corrected)  write %.3f GB  per-launch %.2f MB" %
          (fam, n, rd / 1e9, wr / 1e9, (rd + wr) / n / 1e6))
    summary["families"][fam] = {"launches": n, "read_bytes": rd, "write_bytes": wr, "traffic_bytes_per_launch": (rd + wr) / n}
if len(sys.argv) > 3:
    json.dump(summary, open(sys.argv[3], "w"), indent=1)
