export PYTHONPATH=object-detection-yolov3_amd
L=gpurun_out/r04_x3p_2.log
: > $L
timeout -k 10 500 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv" > gpurun_out/r04_pytest_conv4.log 2>&1; tail -3 gpurun_out/r04_pytest_conv4.log
export Y3_LIB=object-detection-yolov3_amd/yolo3/_lib/libyolo3hip_dev.so
for cfg in "Y3_X3_RSPLIT=2" "Y3_X3_RSPLIT=1" "Y3_X3_NO_PATCH=1 Y3_X3_RSPLIT=1" "Y3_X3_NO_PATCH=1 Y3_X3_RSPLIT=2"; do
echo "=== $cfg" >> $L
env $cfg timeout -k 10 200 python tools/x3_check.py --x3-only --no-ref >> $L 2>&1 || exit 1
done
P=tools/probe/conv_timing
for shape in "8 52 128 256 3" "8 26 256 512 3" "8 13 512 1024 3"; do
  for abl in 0 1; do
  echo "=== x3p $shape | Y3_ABL=$abl" >> $L; Y3_ABL=$abl $P $shape 1 >> $L 2>&1 || exit 1
  done
done
grep -E "^===|^M=|under abl|shader clock|prologue|main loop|epilogue  |launched" $L
