# development: forced K slice counts (Y3_X3_KS) over the three 3x3 shapes
export PYTHONPATH=object-detection-yolov3_amd
L=gpurun_out/r04_x3_ks.log
: > $L
export Y3_LIB=object-detection-yolov3_amd/yolo3/_lib/libyolo3hip_dev.so
for ks in 0 1 2 4 8 16; do
echo "=== Y3_X3_KS=$ks" >> $L
Y3_X3_KS=$ks timeout -k 10 200 python tools/x3_check.py --no-ref --x3-only >> $L 2>&1 || exit 1
done
for cfg in "Y3_X3_RSPLIT=1 Y3_X3_KS=1" "Y3_X3_RSPLIT=1 Y3_X3_KS=2"; do
echo "=== $cfg" >> $L
env $cfg timeout -k 10 200 python tools/x3_check.py --no-ref --x3-only >> $L 2>&1 || exit 1
done
grep -E "^===|^M=" $L
