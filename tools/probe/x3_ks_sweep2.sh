# development: forced K slice counts (Y3_X3_KS) over the three 3x3 shapes
export PYTHONPATH=object-detection-yolov3_amd
export Y3_LIB=object-detection-yolov3_amd/yolo3/_lib/libyolo3hip_dev.so
L=gpurun_out/r04_x3_ks2.log
: > $L
for ks in 0 3 5 6 11 12; do
echo "=== Y3_X3_KS=$ks" >> $L
Y3_X3_KS=$ks timeout -k 10 200 python tools/x3_check.py --x3-only --no-ref >> $L 2>&1
done
grep -E "^===|^M=" $L
