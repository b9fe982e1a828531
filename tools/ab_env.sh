#!/bin/bash
# Same-box sweep of development switches (needs make DEV=1):  bash tools/ab_env.sh <rounds> "<VAR=val ...>" ...   ("" = defaults)
export Y3_LIB=$PWD/object-detection-yolov3_amd/yolo3/_lib/libyolo3hip_dev.so
R=$1; shift
for r in $(seq 1 $R); do
  for cfg in "$@"; do
    env $cfg python bench.py --no-tiled --no-cpu-baseline --no-inference --steps 20 --warmup 5 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-50s %.1f img/s %.3f ms  conv busy %.3f ms' % ('[$cfg]', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step']), flush=True)"
  done
done
