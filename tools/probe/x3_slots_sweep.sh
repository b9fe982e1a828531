# development: workgroup slots a launch fills (Y3_X3_SLOTS 512 / 768) + conv tests + bench
export PYTHONPATH=object-detection-yolov3_amd
L=gpurun_out/r04_x3_slots.log
: > $L
timeout -k 10 200 python tools/x3_check.py --x3-only --no-ref >> $L 2>&1
Y3_LIB=object-detection-yolov3_amd/yolo3/_lib/libyolo3hip_dev.so Y3_X3_SLOTS=768 timeout -k 10 200 python tools/x3_check.py --x3-only --no-ref >> $L 2>&1
grep -E "^===|^M=" $L
timeout -k 10 500 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv" > gpurun_out/r04_pytest_conv6.log 2>&1; tail -3 gpurun_out/r04_pytest_conv6.log
timeout -k 10 300 python bench.py --no-tiled --no-cpu-baseline --no-inference > gpurun_out/r04_bench_f.json 2> gpurun_out/r04_bench_f.err; python -c "
import json; d=json.load(open('gpurun_out/r04_bench_f.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms_per_step'], d['roofline']['by_entry_ms'], d['fp32_mfma_reference']['value'])"
