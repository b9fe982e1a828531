# development: bench.py under Y3_KORDER 0 / 1 / 2 on one box (K order of the multi-tap launches)
export PYTHONPATH=object-detection-yolov3_amd
export Y3_LIB=object-detection-yolov3_amd/yolo3/_lib/libyolo3hip_dev.so
for ko in 2 0 1 2 0; do
Y3_KORDER=$ko timeout -k 10 300 python bench.py --no-tiled --no-cpu-baseline --no-inference --no-f32-reference > gpurun_out/r04_bench_ko$ko.json 2> gpurun_out/r04_bench_ko.err; python -c "
import json; d=json.load(open('gpurun_out/r04_bench_ko$ko.json')); print('korder $ko', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['roofline']['by_entry_ms'])"
done
