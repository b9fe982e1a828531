#!/bin/bash
# The two PMC passes behind roofline.traffic, alone:  bash tools/traffic_pass.sh r04b   ->  gpurun_out/<tag>/traffic.json
set -e -o pipefail
TAG=${1:-r04b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o f --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-inference --no-f32-reference > $OUT/bench_fetch.json 2> $OUT/rocprof_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o w --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-inference --no-f32-reference > $OUT/bench_write.json 2> $OUT/rocprof_write.err
python tools/traffic_summary.py $(find $OUT/fetch -name "*counter_collection.csv") $(find $OUT/write -name "*counter_collection.csv") > $OUT/traffic.json
find $OUT -name "*kernel_trace.csv" -delete
cat $OUT/traffic.json
