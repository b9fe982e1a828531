# A/B of the 256x128 no-residual epilogue (Y3_BF16_DIRECT256=1: direct, 0: staged, the default); writes to stdout
export Y3_LIB=object-detection-yolov3_amd/yolo3/_lib/libyolo3hip_dev.so
for t in 1 0 1 0; do
echo "=== Y3_BF16_DIRECT256=$t"
Y3_BF16_DIRECT256=$t timeout -k 10 100 python tools/bf16_ab.py 2>&1 | grep "bf16 forward"
done
