"""Summarise a rocprofv3 kernel trace: per-kernel totals inside the LAST `frac` of the run (the timed
steps of bench.py, skipping warm-up / MIOpen find-mode kernels).  usage: prof_summary.py trace.csv [frac]"""
import csv
import re
import sys
from collections import defaultdict

path = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
t0, t1 = rows[0][0], rows[-1][1]
cut = t1 - (t1 - t0) * frac
tot = defaultdict(lambda: [0, 0])
for s, e, n in rows:
    if s < cut:
        continue
    n = n.replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", "")
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\((?!anonymous).*", "", n)
    n = n[:120]
    tot[n][0] += e - s
    tot[n][1] += 1
busy = sum(v[0] for v in tot.values())
print("window %.1f ms, kernel-busy %.1f ms, %d launches" % ((t1 - cut) / 1e6, busy / 1e6, sum(v[1] for v in tot.values())))
for n, (d, c) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:int(sys.argv[3]) if len(sys.argv) > 3 else 45]:
    print("%8.2f ms %6d x %8.1f us  %5.1f%%  %s" % (d / 1e6, c, d / c / 1e3, 100.0 * d / busy, n))
