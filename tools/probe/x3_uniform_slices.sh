# development: uniform slice counts (overflow of the 512 slots by SHORT last slices, which the K-slice-major deal places last in every XCD)
export PYTHONPATH=object-detection-yolov3_amd
export Y3_LIB=object-detection-yolov3_amd/yolo3/_lib/libyolo3hip_dev.so
L=gpurun_out/r04_x3_ks3.log
: > $L
for ks in 0 6 0 6 3 12; do
echo "=== Y3_X3_KS=$ks" >> $L
Y3_X3_KS=$ks timeout -k 10 200 python tools/x3_check.py --x3-only --no-ref >> $L 2>&1
done
grep -E "^===|^M=" $L
