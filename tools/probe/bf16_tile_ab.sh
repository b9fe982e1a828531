# development: forced tile of the bf16 ring kernel (Y3_BF16_TILE 0 = planner, 2 = 128x128, 3 = 256x128) on one planned batch of the tiled path
export Y3_LIB=object-detection-yolov3_amd/yolo3/_lib/libyolo3hip_dev.so
for t in 0 3 2; do
echo "=== Y3_BF16_TILE=$t"
Y3_BF16_TILE=$t timeout -k 10 300 python tools/bf16_ab.py 45 608 --layers 2>&1 | grep -v amdgpu.ids
done
