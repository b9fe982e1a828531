export PYTHONPATH=object-detection-yolov3_amd
export Y3_LIB=object-detection-yolov3_amd/yolo3/_lib/libyolo3hip_dev.so
L=gpurun_out/r04_x3_sweep1.log
: > $L
for cfg in "Y3_X3_WGS=700" "Y3_X3_WGS=512" "Y3_X3_WGS=350" "Y3_X3_WGS=1024" "Y3_X3_BN=64 Y3_X3_WGS=700" "Y3_X3_BN=64 Y3_X3_WGS=1024" "Y3_X3_BN=64 Y3_X3_WGS=1400"; do
  echo "=== $cfg" >> $L
  env $cfg timeout -k 10 120 python tools/x3_check.py --no-ref --x3-only >> $L 2>&1 || exit 1
done
P=tools/probe/conv_timing
for shape in "8 52 128 256 3" "8 26 256 512 3" "8 13 512 1024 3"; do
  for cfg in "Y3_X3_WGS=700" "Y3_X3_BN=64 Y3_X3_WGS=700"; do
  echo "=== x3 $shape | $cfg" >> $L; env $cfg $P $shape 1 >> $L 2>&1 || exit 1
  done
done
