# development: cache / instruction-mix counters of the x3 patch kernel (one 26x26 3x3 launch shape, four counter sets, one rocprofv3 pass each)
cd $GRAFT_REPO_ROOT
export PYTHONPATH=object-detection-yolov3_amd TMPDIR=/tmp
O=gpurun_out/r04_pmc
mkdir -p $O
rocprofv3 -L > $O/counters.txt 2>&1
grep -o -E "\b(TCP|TCC|TA|TD|SQ)_[A-Z0-9_]+(sum|avr)?\b" $O/counters.txt | sort -u | tr '\n' ' ' | cut -c1-6000 > $O/counter_names.txt
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_MFMA"; do      # (a fifth set of TA_* counters hung the profiler on this pool and was removed: four sets)
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $O/p$i -o p --output-format csv -- python3 tools/x3_one.py 26 256 512 10 > $O/p$i.log 2>&1
  f=$(find $O/p$i -name "*counter_collection.csv" | head -1)
  echo "== set $i: $set" >> $O/summary.txt
  if [ -n "$f" ]; then python3 - "$f" >> $O/summary.txt <<'PY'
import csv, sys
from collections import defaultdict
acc = defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'conv_x3' in r.get('Kernel_Name', ''):
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in acc.items():
    print('  %-40s mean per dispatch %.4g  (n=%d)' % (k, sum(v) / len(v), len(v)))
PY
  else tail -3 $O/p$i.log >> $O/summary.txt; fi
done
find $O -name "*.csv" -size +5M -delete
cat $O/summary.txt
