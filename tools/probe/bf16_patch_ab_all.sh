# development: patch kernels on / off, same box: bf16 forward at 8x416^2, 8x608^2, 25x608^2 (graph replay) and the tiled 4k image end to end
export Y3_LIB=object-detection-yolov3_amd/yolo3/_lib/libyolo3hip_dev.so
for rep in 1 2; do
for t in 1 0; do
echo "=== Y3_BF16_PATCH=$t (round $rep)"
Y3_BF16_PATCH=$t timeout -k 10 300 python tools/bf16_ab.py 2>&1 | grep "bf16 forward"
Y3_BF16_PATCH=$t timeout -k 10 300 python tools/infer_bench.py 2>&1 | grep "bf16:\|bf16 "
done
done
