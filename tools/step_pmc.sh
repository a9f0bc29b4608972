#!/bin/bash
# SQ counters per kernel over a few training steps: LDS bank-conflict share and where wave cycles go.  (GPU box)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/step_pmc; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES \
  -d "$OUT/p" -o p --output-format csv -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > "$OUT/p.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(out + "/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"])[:70]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
rows = []
for k, d in acc.items():
    wc = d.get("SQ_WAVE_CYCLES", 0.0)
    if wc <= 0: continue
    rows.append((wc, k, d))
rows.sort(reverse=True)
print("%-70s %10s %7s %7s %7s %7s" % ("kernel", "wave-cyc M", "confl%", "park%", "stall%", "issue%"))
for wc, k, d in rows[:40]:
    idx = d.get("SQ_LDS_IDX_ACTIVE", 0.0)
    print("%-70s %10.1f %7.1f %7.1f %7.1f %7.1f" % (k, wc / 1e6, 100 * d.get("SQ_LDS_BANK_CONFLICT", 0) / idx if idx else 0.0,
          100 * d.get("SQ_WAIT_ANY", 0) / wc, 100 * d.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * d.get("SQ_ACTIVE_INST_ANY", 0) / wc))
PY
