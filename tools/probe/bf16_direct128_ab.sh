# development: staged (product) / direct epilogue of the 128 x 128 ring-kernel launches without a residual, same box, all batch sizes
export Y3_LIB=object-detection-yolov3_amd/yolo3/_lib/libyolo3hip_dev.so
for rep in 1 2; do
for t in 0 1; do
echo "=== Y3_BF16_DIRECT128=$t (round $rep)"
Y3_BF16_DIRECT128=$t timeout -k 10 300 python tools/bf16_ab.py 2>&1 | grep "bf16 forward"
Y3_BF16_DIRECT128=$t timeout -k 10 300 python tools/bf16_ab.py 45 608 --layers 2>&1 | grep "bf16 forward\| 256   128 1 1\| 512   128 1 1"
Y3_BF16_DIRECT128=$t timeout -k 10 300 python tools/infer_bench.py 2>&1 | grep "bf16: "
done
done
