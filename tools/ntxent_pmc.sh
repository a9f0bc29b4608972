#!/bin/bash
# NT-Xent kernels: timing table + rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE passes (separate PMC runs, as the
# microarchitecture guide prescribes; FETCH_SIZE x2 on gfx950).  Output: gpurun_out/ntxent/r3_ntxent.txt
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/ntxent; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
{
echo "== tools/ntxent_bench.py (wall clock over repeated launches) =="
python3 "$ROOT/tools/ntxent_bench.py" 2>/dev/null | grep -v amdgpu.ids
echo
echo "== rocprofv3 --kernel-trace --stats -- python3 tools/ntxent_bench.py --once =="
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- python3 "$ROOT/tools/ntxent_bench.py" --once > "$OUT/trace.log" 2>&1
S=$(find "$OUT/trace" -name '*kernel_stats.csv' | head -1)
grep -i "ntxent\|Name" "$S" | cut -c1-220
T=$(find "$OUT/trace" -name '*kernel_trace.csv' | head -1)
python3 - "$T" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "ntxent_kernel" in r["Kernel_Name"]]
print("per dispatch (ns): kernel, grid, duration")
for r in rows:
    print("  %-70s grid %8s  %10d" % (r["Kernel_Name"][:70], r.get("Grid_Size", r.get("Grid_Size_X", "?")), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
PY
echo
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/pmc_$C" -o p -- python3 "$ROOT/tools/ntxent_bench.py" --once > "$OUT/pmc_$C.log" 2>&1
  F=$(find "$OUT/pmc_$C" -name '*counter_collection.csv' | head -1)
  echo "== --pmc $C (per dispatch of ntxent_kernel; FETCH_SIZE in KB, x2 on gfx950 for 16-byte streaming reads) =="
  python3 - "$F" $C <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "ntxent_kernel" in r["Kernel_Name"] and r["Counter_Name"] == sys.argv[2]:
        print("  %-70s grid %8s  %s = %.1f KB" % (r["Kernel_Name"][:70], r.get("Grid_Size", "?"), sys.argv[2], float(r["Counter_Value"])))
PY
done
} > "$OUT/r3_ntxent.txt" 2>&1
rm -rf "$OUT/trace" "$OUT"/pmc_*
cat "$OUT/r3_ntxent.txt"
