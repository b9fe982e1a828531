# A/B: dynamic-LDS pad of the kernel-gradient launches (caps their workgroups per CU so that the BatchNorm-backward
# finalize kernel of the compute stream finds LDS).  Needs the development library: make -C object-detection-yolov3_amd/csrc DEV=1
export Y3_LIB=$PWD/object-detection-yolov3_amd/yolo3/_lib/libyolo3hip_dev.so
B="python bench.py --no-tiled --no-cpu-baseline --no-inference --steps 20"
pick() { python -c "
import sys,json
d=json.loads([l for l in open('$1') if l.startswith('{')][-1])
print('$1', round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['kernel_ms_per_step'],3))"; }
for rep in 1 2; do
for cfg in "0 0" "8192 0" "8192 4096" "8192 8192" "0 4096"; do
set -- $cfg
Y3_WGRAD_PAD40=$1 Y3_WGRAD_PAD32=$2 $B > gpurun_out/ab_pad_$1_$2_$rep.json 2>/dev/null; pick gpurun_out/ab_pad_$1_$2_$rep.json
done
done
