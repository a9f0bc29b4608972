#!/bin/bash
# SQ counter passes over one GEMM shape: where the waves' cycles go.  usage: tools/gemm_pmc.sh M N K  (on the GPU box)
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/gemm_pmc_$1_$2_$3_${5:-nn}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_BF16 \
  -d "$OUT/p1" -o p1 --output-format csv -- python3 "$ROOT/tools/gemm_one.py" "$@" > "$OUT/p1.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_MISC SQ_INSTS_LDS SQ_WAIT_INST_ANY \
  -d "$OUT/p2" -o p2 --output-format csv -- python3 "$ROOT/tools/gemm_one.py" "$@" > "$OUT/p2.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for p in ("p1", "p2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob(out + "/" + p + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for k, d in acc.items():
        if "gemm" not in k: continue
        print(p, k)
        for c, v in sorted(d.items()):
            print("    %-36s %16.0f per launch" % (c, v / max(n[(k, c)], 1)))
PY
