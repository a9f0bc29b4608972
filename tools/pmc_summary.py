"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes per kernel family.
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x
(MI355X_MICROARCH.md section HBM), so reads are doubled.

    pmc_summary.py fetch.csv write.csv [out.json]                      one pass of each counter
    pmc_summary.py --passes f1.csv,f2.csv,f3.csv w1.csv,w2.csv,w3.csv out.json
                                                                       several passes: per family the MEDIAN pass is the figure
                                                                       (`traffic_bytes_per_launch`), min / max are reported next to it
The summary carries the content hash of the kernel sources it was taken at (druglamp_amd.build.csrc_hash): bench.py uses a
summary only when the hash matches the tree being benched."""
import csv
import json
import os
import statistics
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


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


args = sys.argv[1:]
multi = bool(args) and args[0] == "--passes"
if multi:
    args = args[1:]
fetch_files, write_files = args[0].split(","), args[1].split(",")
fetches = [load(f, "FETCH_SIZE") for f in fetch_files if f]
writes = [load(f, "WRITE_SIZE") for f in write_files if f]
from druglamp_amd.build import csrc_hash
summary = {"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python bench.py --steps 2 --warmup 1 "
                      "--no-cpu-baseline --no-kernel-timing (separate passes per counter; %d + %d passes)" % (len(fetches), len(writes)),
           "correction": "FETCH_SIZE, WRITE_SIZE in KiB; FETCH_SIZE doubled (gfx950 reports 1/2 of wide coalesced reads, "
                         "MI355X_MICROARCH.md section HBM)",
           "csrc_sha1": csrc_hash(), "passes": {"fetch": len(fetches), "write": len(writes)}, "families": {}}
fams = set()
for t in fetches + writes:
    fams |= set(t)
for fam in sorted(fams):
    rds = [2.0 * t.get(fam, [0.0, 0])[0] * 1024 for t in fetches]
    wrs = [t.get(fam, [0.0, 0])[0] * 1024 for t in writes]
    n = max([t.get(fam, [0.0, 0])[1] for t in fetches + writes] + [1])
    rd, wr = statistics.median(rds or [0.0]), statistics.median(wrs or [0.0])
    print("%-14s launches %6d  read %.3f GB (x2-corrected; passes %s)  write %.3f GB (passes %s)  per-launch %.2f MB" %
          (fam, n, rd / 1e9, "/".join("%.3f" % (v / 1e9) for v in rds), wr / 1e9, "/".join("%.3f" % (v / 1e9) for v in wrs),
           (rd + wr) / n / 1e6))
    summary["families"][fam] = {"launches": n, "read_bytes": rd, "write_bytes": wr, "traffic_bytes_per_launch": (rd + wr) / n,
                                "traffic_bytes_per_launch_min": (min(rds or [0.0]) + min(wrs or [0.0])) / n,
                                "traffic_bytes_per_launch_max": (max(rds or [0.0]) + max(wrs or [0.0])) / n,
                                "read_bytes_passes": rds, "write_bytes_passes": wrs}
if len(args) > 2:
    json.dump(summary, open(args[2], "w"), indent=1)
