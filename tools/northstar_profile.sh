#!/bin/bash
# North-star block (tools/northstar_block.py): host-timed numbers, rocprofv3 per-kernel durations, FETCH_SIZE / WRITE_SIZE passes.
# (GPU box)  usage: tools/northstar_profile.sh  -> gpurun_out/northstar/summary.txt
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/northstar; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python3 "$ROOT/tools/northstar_block.py" > "$OUT/host_timed.txt" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- python3 "$ROOT/tools/northstar_block.py" --once > "$OUT/trace.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_f" -o f -- python3 "$ROOT/tools/northstar_block.py" --once > "$OUT/pmc_f.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_w" -o w -- python3 "$ROOT/tools/northstar_block.py" --once > "$OUT/pmc_w.log" 2>&1
python3 - "$OUT" <<'PY' > "$OUT/summary.txt" 2>&1
import csv, glob, sys, collections, re
out = sys.argv[1]
print(open(out + "/host_timed.txt").read())
st = glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True)
print("# rocprofv3 --kernel-trace --stats -- python3 tools/northstar_block.py --once  (B = 256 passes + the 8-pair parity passes; per-kernel average ns)")
if st:
    rows = list(csv.DictReader(open(st[0])))
    for r in rows[:24]:
        print("%-100s calls %5s  avg %10.0f ns  total %8.3f ms" % (re.sub(r"\(anonymous namespace\)::|void ", "", r["Name"])[:100], r["Calls"], float(r["AverageNs"]), float(r["TotalDurationNs"]) / 1e6))
def load(pat, name):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(out + pat, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != name: continue
            k = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"])[:60]
            acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    return acc
fe, wr = load("/pmc_f/**/*counter_collection.csv", "FETCH_SIZE"), load("/pmc_w/**/*counter_collection.csv", "WRITE_SIZE")
print("# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes; KiB -> bytes, FETCH_SIZE x2 per MI355X_MICROARCH.md section HBM), all launches of the run:")
tr = tw = 0.0
for k in sorted(set(fe) | set(wr), key=lambda k: -(2 * fe.get(k, [0, 0])[0] + wr.get(k, [0, 0])[0])):
    r, w = 2 * fe.get(k, [0, 0])[0] * 1024, wr.get(k, [0, 0])[0] * 1024
    tr += r; tw += w
    print("%-62s launches %4d  read %8.1f MB  write %8.1f MB" % (k, max(fe.get(k, [0, 0])[1], wr.get(k, [0, 0])[1]), r / 1e6, w / 1e6))
print("total read %.1f MB, write %.1f MB over the run (parity passes on 8 pairs + %s timed fwd / bwd passes at B = 256)" % (tr / 1e6, tw / 1e6, "1 + 3"))
PY
cat "$OUT/summary.txt"
