# development: CU-partner pairing of the x3 patch kernel's work items (Y3_X3_MODE bit 1 = off) and forced column-major ids (bit 2 = off)
export PYTHONPATH=object-detection-yolov3_amd
export Y3_LIB=object-detection-yolov3_amd/yolo3/_lib/libyolo3hip_dev.so
L=gpurun_out/r04_x3_pair.log
: > $L
for mode in 1 3 7 5 1; do
echo "=== Y3_X3_MODE=$mode" >> $L
Y3_X3_MODE=$mode timeout -k 10 200 python tools/x3_check.py --x3-only --no-ref >> $L 2>&1
done
grep -E "^===|^M=" $L
unset Y3_LIB
timeout -k 10 500 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv" > gpurun_out/r04_pytest_conv7.log 2>&1; tail -3 gpurun_out/r04_pytest_conv7.log
timeout -k 10 300 python bench.py --no-tiled --no-cpu-baseline --no-inference > gpurun_out/r04_bench_g.json 2> gpurun_out/r04_bench_g.err; python -c "
import json; d=json.load(open('gpurun_out/r04_bench_g.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms_per_step'], d['roofline']['by_entry_ms'], d['fp32_mfma_reference']['value'])"
