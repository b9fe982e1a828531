#!/bin/bash
# A/B two builds of the library on ONE box (box-to-box spread is ~1 %):  bash tools/ab.sh [repeats]
# _lib/libyolo3hip_prev.so is the reference build (make -C <copy of csrc at the old commit> OUT=.../libyolo3hip_prev.so)
P=$PWD/object-detection-yolov3_amd/yolo3/_lib
for i in $(seq 1 ${1:-2}); do
  echo -n "prev: "; Y3_LIB=$P/libyolo3hip_prev.so python tools/fwd_time.py 2>&1 | tail -1 | sed 's/.*| inference/inference/'
  echo -n "new:  "; Y3_LIB=$P/libyolo3hip.so python tools/fwd_time.py 2>&1 | tail -1 | sed 's/.*| inference/inference/'
done
