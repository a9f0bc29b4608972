#!/bin/bash
# SQ counters of the attention kernels at one of the path's shapes.  usage: tools/attn_pmc.sh self|paired|pgca
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=${1:-self}
OUT=$ROOT/gpurun_out/attn_pmc_$W; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES \
  -d "$OUT/a" -o a --output-format csv -- python3 "$ROOT/tools/attn_one.py" $W > "$OUT/a.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE \
  -d "$OUT/b" -o b --output-format csv -- python3 "$ROOT/tools/attn_one.py" $W > "$OUT/b.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for p in ("a", "b"):
    for f in glob.glob(out + "/%s/**/*counter_collection.csv" % p, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "attn" not in k: continue
            k = k[k.find("attn"):][:44]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k, d in acc.items():
    print(k)
    g = d["GRBM_GUI_ACTIVE"] / max(n[(k, "GRBM_GUI_ACTIVE")], 1) / 8
    for c, v in sorted(d.items()):
        print("    %-28s %14.0f per launch" % (c, v / max(n[(k, c)], 1)))
    w = d["SQ_WAVES"] / n[(k, "SQ_WAVES")]
    print("    -> launch %.0f k cycles; per SIMD: matrix pipe busy %.0f %%, VALU issuing %.0f %%; LDS wait share of wave time %.0f %%" % (
        g / 1e3, 100 * d["SQ_VALU_MFMA_BUSY_CYCLES"] / n[(k, "SQ_VALU_MFMA_BUSY_CYCLES")] / 1024 / g,
        100 * 4 * d["SQ_ACTIVE_INST_VALU"] / n[(k, "SQ_ACTIVE_INST_VALU")] / 1024 / g,
        100 * d["SQ_WAIT_INST_LDS"] / n[(k, "SQ_WAIT_INST_LDS")] / (d["SQ_WAVE_CYCLES"] / n[(k, "SQ_WAVE_CYCLES")])))
PY
