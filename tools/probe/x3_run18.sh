export PYTHONPATH=object-detection-yolov3_amd
timeout -k 10 500 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv" > gpurun_out/r04_pytest_conv11.log 2>&1; tail -3 gpurun_out/r04_pytest_conv10.log
timeout -k 10 300 python tools/layer_times.py > gpurun_out/r04_layer_times_korder2.txt 2>&1
tail -2 gpurun_out/r04_layer_times_korder2.txt
timeout -k 10 300 python bench.py --no-tiled --no-cpu-baseline --no-inference > gpurun_out/r04_bench_k.json 2> gpurun_out/r04_bench_k.err; python -c "
import json; d=json.load(open('gpurun_out/r04_bench_k.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms_per_step'], d['roofline']['by_entry_ms'], d['fp32_mfma_reference']['value'])"
timeout -k 10 500 bash tools/traffic_pass.sh r04e > gpurun_out/r04e_traffic.log 2>&1; tail -8 gpurun_out/r04e_traffic.log
