# development: the patch kernel of the 32 -> 64 3x3 layers (conv_bf16_c32_kernel) on / off on one planned batch of the tiled path
export Y3_LIB=object-detection-yolov3_amd/yolo3/_lib/libyolo3hip_dev.so
for t in 1 0; do
echo "=== Y3_BF16_PATCH=$t"
Y3_BF16_PATCH=$t timeout -k 10 300 python tools/bf16_ab.py 45 608 --layers 2>&1 | grep -v amdgpu.ids
done
