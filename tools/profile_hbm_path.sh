#!/bin/bash
# rocprofv3 passes over the HBM-bound kernels of the path (decode, NMS, z-score, Adam):  bash tools/profile_hbm_path.sh r03
# -> gpurun_out/<tag>/hbm_path.md (copy to profiles/<tag>_hbm_path.md) and the kernel-trace stats CSV
set -e -o pipefail
TAG=${1:-r03}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/hp_trace -o t --output-format csv -- python3 tools/hbm_path_driver.py > $OUT/hp_trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/hp_fetch -o f --output-format csv -- python3 tools/hbm_path_driver.py > $OUT/hp_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/hp_write -o w --output-format csv -- python3 tools/hbm_path_driver.py > $OUT/hp_write.log 2>&1
python tools/hbm_path_summary.py $(find $OUT/hp_trace -name "*kernel_trace.csv") $(find $OUT/hp_fetch -name "*counter_collection.csv") $(find $OUT/hp_write -name "*counter_collection.csv") > $OUT/hbm_path.md
cp $(find $OUT/hp_trace -name "*kernel_stats.csv") $OUT/hbm_path_kernel_stats.csv
find $OUT/hp_* -name "*.csv" -size +30M -delete
cat $OUT/hbm_path.md
