export PYTHONPATH=object-detection-yolov3_amd
timeout -k 10 500 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv" > gpurun_out/r04_pytest_conv12.log 2>&1; tail -2 gpurun_out/r04_pytest_conv12.log
timeout -k 10 200 python tools/x3_check.py --x3-only --no-ref > gpurun_out/r04_x3_kzmajor.log 2>&1; grep -E "^M=" gpurun_out/r04_x3_kzmajor.log
timeout -k 10 300 python bench.py --no-tiled --no-cpu-baseline --no-inference > gpurun_out/r04_bench_l.json 2> gpurun_out/r04_bench_l.err; python -c "
import json; d=json.load(open('gpurun_out/r04_bench_l.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms_per_step'], d['roofline']['by_entry_ms'], d['fp32_mfma_reference']['value'])"
timeout -k 10 500 bash tools/traffic_pass.sh r04f > gpurun_out/r04f_traffic.log 2>&1; tail -7 gpurun_out/r04f_traffic.log
