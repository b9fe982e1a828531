# A/B of the data-parallel transports on ONE GPU (one-rank communicator, collectives forced): bench lines side by side.
B="python bench.py --no-tiled --no-cpu-baseline --no-inference --steps 20"
pick() { python -c "
import sys,json
d=json.loads([l for l in open('$1') if l.startswith('{')][-1]); c=d.get('comm') or {}
print('$1', round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['kernel_ms_per_step'],3), c.get('allreduce_ms_sum'), c.get('allreduce_ms_exposed'))"; }
for rep in 1 2; do
$B > gpurun_out/ab_plain_$rep.json 2>/dev/null; pick gpurun_out/ab_plain_$rep.json
$B --force-collective > gpurun_out/ab_torch_lean_$rep.json 2>/dev/null; pick gpurun_out/ab_torch_lean_$rep.json
Y3_DP_OWN_STREAM=1 $B --force-collective > gpurun_out/ab_torch_own_$rep.json 2>/dev/null; pick gpurun_out/ab_torch_own_$rep.json
$B --force-collective --transport native > gpurun_out/ab_native_$rep.json 2>/dev/null; pick gpurun_out/ab_native_$rep.json
$B --force-collective --backend gloo > gpurun_out/ab_gloo_$rep.json 2>/dev/null; pick gpurun_out/ab_gloo_$rep.json
done
