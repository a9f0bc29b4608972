#!/bin/bash
# SQ counters of the GELU-epilogue GEMM on the 256x256 tile (one workgroup per CU) and on 256x128 tiles with two workgroups
# per CU (study library): do the second workgroup's VALU / store phases overlap the first one's matrix phases?
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/gemm_epi_pmc; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export DL_USE_STUDY_LIB=1
for cfg in 0 1; do
  export DL_GEMM_BIGCFG=$cfg
  rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES \
    -d "$OUT/c${cfg}a" -o a --output-format csv -- python3 "$ROOT/tools/gemm_epi_one.py" > "$OUT/c${cfg}a.log" 2>&1
  rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE \
    -d "$OUT/c${cfg}b" -o b --output-format csv -- python3 "$ROOT/tools/gemm_epi_one.py" > "$OUT/c${cfg}b.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for cfg in ("0", "1"):
    print("DL_GEMM_BIGCFG=%s" % cfg)
    for p in ("a", "b"):
        acc = collections.defaultdict(float); n = collections.Counter()
        for f in glob.glob(out + "/c%s%s/**/*counter_collection.csv" % (cfg, p), recursive=True):
            for r in csv.DictReader(open(f)):
                if "gemm_big" not in r["Kernel_Name"]: continue
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
        for c, v in sorted(acc.items()):
            print("    %-30s %16.0f per launch" % (c, v / max(n[c], 1)))
PY
