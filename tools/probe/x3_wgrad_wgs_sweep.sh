# development: workgroups an x3 kernel gradient aims at (Y3_WGX3_WGS)
export PYTHONPATH=object-detection-yolov3_amd
export Y3_LIB=object-detection-yolov3_amd/yolo3/_lib/libyolo3hip_dev.so
L=gpurun_out/r04_wgx3_wgs.log
: > $L
for w in 480 600 700 480 600 1000; do
echo "=== Y3_WGX3_WGS=$w" >> $L
Y3_WGX3_WGS=$w timeout -k 10 200 python tools/x3_check.py --wgrad >> $L 2>&1
done
grep -E "^===|^wgrad" $L
