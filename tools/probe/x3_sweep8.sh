export PYTHONPATH=object-detection-yolov3_amd
export Y3_LIB=object-detection-yolov3_amd/yolo3/_lib/libyolo3hip_dev.so
L=gpurun_out/r04_x3_nt.log
: > $L
for mode in 0 1 0 1; do
echo "=== Y3_X3_MODE=$mode" >> $L
Y3_X3_MODE=$mode timeout -k 10 200 python tools/x3_check.py --x3-only --no-ref >> $L 2>&1
done
P=tools/probe/conv_timing
for shape in "8 26 256 512 3"; do
  for mode in 0 1; do
  echo "=== x3p $shape | mode $mode" >> $L; Y3_X3_MODE=$mode $P $shape 1 >> $L 2>&1 || exit 1
  done
done
grep -E "^===|^M=|under abl|shader clock|prologue|main loop|epilogue  " $L
