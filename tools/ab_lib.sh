#!/bin/bash
# Same-box A/B of builds of the library (boxes differ by several per cent under MFMA load, so numbers from different gpurun
# calls do not compare):  bash tools/ab_lib.sh <rounds> <name=lib.so> ...   ("cur" = the product library) -> serial layer totals
# and the bench line per build, alternating
R=$1; shift
mkdir -p gpurun_out
for r in $(seq 1 $R); do
  for spec in "$@" cur=; do
    which=${spec%%=*}; lib=${spec#*=}
    if [ -n "$lib" ]; then export Y3_LIB=$PWD/$lib; else unset Y3_LIB; fi
    python tools/layer_times.py > gpurun_out/ab_${which}_${r}_layers.txt 2>&1
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-tiled --no-inference > gpurun_out/ab_${which}_${r}.json 2>/dev/null
    python - <<PY
import json, ast
d=json.loads(open("gpurun_out/ab_${which}_${r}.json").read().strip().splitlines()[-1])
t=[l for l in open("gpurun_out/ab_${which}_${r}_layers.txt") if l.startswith("totals")][0]
tt=ast.literal_eval(t.split(":",1)[1].strip())
print("${which} ${r}: %.1f img/s %.3f ms frac %.3f | fwd %.0f dgradb %.0f wgrad %.0f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], tt["fwd"], tt["dgradb"], tt["wgrad"]), flush=True)
PY
  done
done
