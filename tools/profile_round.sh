#!/bin/bash
# Refresh the judged measurements of a round on the GPU box:  bash tools/profile_round.sh r01
# Writes under gpurun_out/<tag>/ ; copy the summaries into profiles/ afterwards (tools/traffic_summary.py for the PMC passes).
# bench.py reads profiles/<tag>_traffic.json for roofline.traffic: after a conv change, copy the new traffic.json first and run
# bench.py once more for the committed bench line.
set -e -o pipefail
TAG=${1:-r04}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --steps 20 --warmup 5 > $OUT/bench_n1.json 2> $OUT/bench_n1.err
echo "bench done"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o s --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-inference --no-f32-reference > $OUT/bench_under_rocprof.json 2> $OUT/rocprof_stats.err
echo "stats done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o f --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-inference --no-f32-reference > $OUT/bench_fetch.json 2> $OUT/rocprof_fetch.err
echo "fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o w --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-inference --no-f32-reference > $OUT/bench_write.json 2> $OUT/rocprof_write.err
echo "write done"
python tools/traffic_summary.py $(find $OUT/fetch -name "*counter_collection.csv") $(find $OUT/write -name "*counter_collection.csv") > $OUT/traffic.json
python tools/layer_times.py > $OUT/layer_times.txt 2>&1
python tools/infer_bench.py --layers > $OUT/infer_bench.txt 2>&1
python tools/trace_union.py $(find $OUT/stats -name "*kernel_trace.csv") > $OUT/trace_union.json
# MFMA utilisation of the shipping conv kernels: one counter pass over the per-shape driver (program directly after --)
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $OUT/pmc -o p --output-format csv -- python3 tools/conv_tune.py > $OUT/pmc_conv_tune.txt 2> $OUT/rocprof_pmc.err
python tools/pmc_summary.py $(find $OUT/pmc -name "*counter_collection.csv") conv_ > $OUT/pmc_conv.txt
python tools/mfma_util.py $(find $OUT/pmc -name "*counter_collection.csv") conv_ > $OUT/mfma_util_conv.md
# the x3 kernels (round 4): forward / data gradient of the three 3x3 shapes and the kernel gradients, same counters
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $OUT/pmcx -o p --output-format csv -- python3 tools/x3_check.py --no-ref --x3-only > $OUT/pmc_x3_check.txt 2> $OUT/rocprof_pmcx.err
echo "pmc x3 fwd/dgrad done"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $OUT/pmcw -o p --output-format csv -- python3 tools/x3_check.py --wgrad > $OUT/pmc_x3_wgrad.txt 2> $OUT/rocprof_pmcw.err
echo "pmc x3 wgrad done"
(echo "## forward / data gradient (tools/x3_check.py --x3-only)"; python tools/mfma_util.py $(find $OUT/pmcx -name "*counter_collection.csv") conv_x3; echo; echo "## kernel gradient (tools/x3_check.py --wgrad; fp32-MFMA rows for comparison)"; python tools/mfma_util.py $(find $OUT/pmcw -name "*counter_collection.csv") conv_wgrad) > $OUT/mfma_util_x3.md
# bf16: the 8 x 608^2 forward and the 45-tile batch the tiled path plans for a 4k image (conv_bf16_pp_kernel layers)
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $OUT/pmcb -o p --output-format csv -- python3 tools/bf16_probe.py > $OUT/pmc_bf16_probe.txt 2> $OUT/rocprof_pmcb.err
python tools/pmc_summary.py $(find $OUT/pmcb -name "*counter_collection.csv") conv_bf16 > $OUT/pmc_bf16.txt
(echo "## 8 x 608^2 forward"; python tools/mfma_util.py $(find $OUT/pmcb -name "*counter_collection.csv") conv_bf16) > $OUT/mfma_util_bf16.md
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $OUT/pmcb45 -o p --output-format csv -- python3 tools/bf16_probe.py 45 > $OUT/pmc_bf16_probe45.txt 2> $OUT/rocprof_pmcb45.err
(echo; echo "## 45 x 608^2 forward (one planned batch of the tiled 4k path)"; python tools/mfma_util.py $(find $OUT/pmcb45 -name "*counter_collection.csv") conv_bf16) >> $OUT/mfma_util_bf16.md
python tools/bf16_ab.py 45 608 --layers > $OUT/bf16_layers_45x608.txt 2>&1
bash tools/profile_hbm_path.sh $TAG > $OUT/hbm_path.log 2>&1
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -size +20M -delete
ls -la $OUT $OUT/stats
