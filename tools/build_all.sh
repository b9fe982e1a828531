#!/bin/bash
# development: product library, DEV library (tuning switches compiled in) and the conv timing probe, from any directory
R=$(cd "$(dirname "$0")/.." && pwd)
make -C $R/object-detection-yolov3_amd/csrc 2>&1 | grep -E " error|Error " 
make -C $R/object-detection-yolov3_amd/csrc DEV=1 2>&1 | grep -E " error|Error "
if [ "$1" != "nopro" ]; then
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -Wno-unused-value -DY3_TIMING -DY3_DEV -I $R/include -I $R/object-detection-yolov3_amd/csrc \
    $R/tools/probe/conv_timing.hip $R/object-detection-yolov3_amd/csrc/core.hip -o $R/tools/probe/conv_timing 2>&1 | grep -E " error" -A5 | head
fi
ls -la $R/tools/probe/conv_timing $R/object-detection-yolov3_amd/yolo3/_lib/*.so | awk '{print $5, $6, $7, $8, $9}'
