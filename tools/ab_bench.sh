#!/bin/bash
# A/B of builds of libd3dp_hip on the same box: tools/ab_bench.sh <libA> <libB> [bench args...], or LIBS="a b c ..."
# tools/ab_bench.sh [bench args...]; "default" = the in-tree library.
if [ -n "$LIBS" ]; then set -- $LIBS -- "$@"; else A=$1; B=$2; shift 2; set -- "$A" "$B" -- "$@"; fi
L=(); while [ "$1" != "--" ]; do L+=("$1"); shift; done; shift
for lib in "${L[@]}"; do
  python - "$lib" "$@" <<'PY'
import json, runpy, sys, io, contextlib
lib, args = sys.argv[1], sys.argv[2:]
import d3dp_amd._lib as l
if lib != "default":
    l.LIB_PATH = lib
sys.argv = ["bench.py"] + args
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    runpy.run_path("bench.py", run_name="__main__")
d = json.loads([x for x in buf.getvalue().splitlines() if x.startswith("{")][-1])
k = d.get("kernel_ms_per_step", {})
print(lib.split("/")[-1], round(d["value"], 2), round(d["ms_per_step"], 1), {n: round(v) for n, v in k.items() if v > 100})
PY
done
