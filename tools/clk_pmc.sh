#!/bin/bash
# Shader clock under load: GRBM_GUI_ACTIVE cycles / kernel duration for one GEMM shape.  usage: tools/clk_pmc.sh M N K
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/clk_pmc; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT -d "$OUT/p" -o p --output-format csv -- python3 "$ROOT/tools/gemm_one.py" "$@" > "$OUT/p.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm" in r["Kernel_Name"]:
            us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            print(r["Counter_Name"], r["Counter_Value"], "%.1f us" % us, "cycles/us = %.0f" % (float(r["Counter_Value"]) / us))
PY
