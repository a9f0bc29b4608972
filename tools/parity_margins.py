"""Per-test parity margins from a DL_PARITY_LOG file (tests/helpers.py logs every relerr / elemerr evaluation):

    DL_PARITY_LOG=/tmp/margins.jsonl python -m pytest tests/test_parity_gpu.py tests/test_model_gpu.py -m gpu -q
    python tools/parity_margins.py /tmp/margins.jsonl > profiles/r6_parity_margins.txt

One row per (test, parametrisation): the worst max-norm error (`relerr`: max |a - b| / max |b|) and the worst element-wise error
with a 1 % floor (`elemerr`: max |a - b| / (|b| + 0.01 max |b|)) over all comparisons the test made."""
import collections
import json
import sys

rows = collections.OrderedDict()
for line in open(sys.argv[1]):
    r = json.loads(line)
    k = r["test"]
    cur = rows.setdefault(k, {"n": 0, "relerr": 0.0, "elemerr": 0.0, "line_rel": 0, "line_elem": 0})
    cur["n"] += 1
    if r["relerr"] > cur["relerr"]:
        cur["relerr"], cur["line_rel"] = r["relerr"], r["line"]
    if r["elemerr"] > cur["elemerr"]:
        cur["elemerr"], cur["line_elem"] = r["elemerr"], r["line"]
print("%-110s %6s %11s %6s %11s %6s" % ("test", "cmps", "max relerr", "line", "max elemerr", "line"))
for k, v in rows.items():
    print("%-110s %6d %11.3e %6d %11.3e %6d" % (k[-110:], v["n"], v["relerr"], v["line_rel"], v["elemerr"], v["line_elem"]))
