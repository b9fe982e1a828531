#!/bin/bash
# rocprofv3 passes over the bf16 patch kernels on one planned batch of the tiled path:  bash tools/profile_bf16_patch.sh r04
# -> gpurun_out/<tag>/bf16_patch.md (copy to profiles/<tag>_bf16_patch.md) and the kernel-trace stats CSV
set -e -o pipefail
TAG=${1:-r04}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/bp_trace -o t --output-format csv -- python3 tools/bf16_ab.py 45 608 > $OUT/bp_trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/bp_fetch -o f --output-format csv -- python3 tools/bf16_ab.py 45 608 > $OUT/bp_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/bp_write -o w --output-format csv -- python3 tools/bf16_ab.py 45 608 > $OUT/bp_write.log 2>&1
python tools/bf16_patch_summary.py $(find $OUT/bp_trace -name "*kernel_trace.csv") $(find $OUT/bp_fetch -name "*counter_collection.csv") $(find $OUT/bp_write -name "*counter_collection.csv") > $OUT/bf16_patch.md
cp $(find $OUT/bp_trace -name "*kernel_stats.csv") $OUT/bf16_patch_kernel_stats.csv
find $OUT/bp_* -name "*.csv" -size +30M -delete
cat $OUT/bf16_patch.md
