export PYTHONPATH=object-detection-yolov3_amd
L=gpurun_out/r04_x3_pad.log
: > $L
export Y3_LIB=object-detection-yolov3_amd/yolo3/_lib/libyolo3hip_dev.so
for pad in 0 16 32; do
for cfg in "Y3_X3_RSPLIT=2" "Y3_X3_NO_PATCH=1"; do
echo "=== pad $pad $cfg" >> $L
env $cfg timeout -k 10 200 python tools/x3_check.py --no-ref --pad $pad >> $L 2>&1 || exit 1
done
done
grep -E "^===|^M=" $L
