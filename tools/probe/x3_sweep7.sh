export PYTHONPATH=object-detection-yolov3_amd
L=gpurun_out/r04_x3_planes.log
: > $L
timeout -k 10 500 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv" > gpurun_out/r04_pytest_conv5.log 2>&1; tail -3 gpurun_out/r04_pytest_conv5.log
timeout -k 10 200 python tools/x3_check.py --x3-only --no-ref >> $L 2>&1
timeout -k 10 200 python tools/x3_check.py --stride2 >> $L 2>&1
P=tools/probe/conv_timing
for shape in "8 26 256 512 3" "8 52 128 256 3" "8 13 512 1024 3"; do
  for abl in 0 1 2; do
  echo "=== x3p $shape | Y3_ABL=$abl" >> $L; Y3_ABL=$abl $P $shape 1 >> $L 2>&1 || exit 1
  done
done
grep -E "^===|^M=|stride-2|under abl|shader clock|prologue|main loop|epilogue  " $L
timeout -k 10 300 python bench.py --no-tiled --no-cpu-baseline --no-inference > gpurun_out/r04_bench_e.json 2> gpurun_out/r04_bench_e.err; python -c "
import json; d=json.load(open('gpurun_out/r04_bench_e.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms_per_step'], d['roofline']['by_entry_ms'], d['fp32_mfma_reference']['value'])"
