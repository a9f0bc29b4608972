#!/bin/bash
# Same-box comparison of this tree against another commit of it (round 5: how profiles/r5_same_box_r4_vs_r5.txt was made).
#   here (no GPU):   tools/same_box_vs_commit.sh prepare <commit>     # git archive of <commit> into _other/ + its library build
#   on the GPU box:  gpurun -- 'bash tools/same_box_vs_commit.sh run [bench.py arguments]'
# `run` alternates the two trees three times (a throwaway run first: a fresh box runs its first process ~2 % slower) and prints
# ms per step; outputs under gpurun_out/same_box/.  _other/ is git-ignored but travels with the gpurun snapshot; delete it afterwards.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
case "$1" in
  prepare)
    rm -rf "$ROOT/_other" && mkdir -p "$ROOT/_other" || exit 1
    (cd "$ROOT" && git archive "$2" druglamp_amd include bench.py oracle profiles | tar -x -C "$ROOT/_other") || exit 1
    (cd "$ROOT/_other" && python -m druglamp_amd.build | tail -1)
    ;;
  run)
    shift
    ARGS=${*:---steps 100 --no-cpu-baseline --no-kernel-timing}
    OUT=$ROOT/gpurun_out/same_box; mkdir -p "$OUT"
    (cd "$ROOT" && python bench.py --steps 30 --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1)
    for rep in 1 2 3; do
      (cd "$ROOT/_other" && python bench.py $ARGS 2>/dev/null | tail -1 > "$OUT/other_$rep.json")
      (cd "$ROOT" && python bench.py $ARGS 2>/dev/null | tail -1 > "$OUT/this_$rep.json")
    done
    for f in "$OUT"/other_*.json "$OUT"/this_*.json; do
      python -c "import json; d=json.loads(open('$f').read()); print('$f'.split('/')[-1], d['ms_per_step'])"
    done
    ;;
  *) echo "usage: $0 prepare <commit> | run [bench.py arguments]"; exit 2;;
esac
