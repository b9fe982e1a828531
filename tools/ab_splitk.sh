# A/B (development library): split-K policy for the 1x1 / mid-size launches
export Y3_LIB=$PWD/object-detection-yolov3_amd/yolo3/_lib/libyolo3hip_dev.so
for cfg in "" "Y3_SPLITK_MINK=128" "Y3_FEWTILES=700 Y3_SPLITK_MINK=128" "Y3_FEWTILES=700" "Y3_SPLITK_MINK=128 Y3_SPLITK_WGS=1400"; do
  echo "=== $cfg"
  env $cfg python tools/conv_tune.py 2>&1 | grep -E "k=1|cin= 128 cout= 256"
  env $cfg python bench.py --no-tiled --no-cpu-baseline --no-inference --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['kernel_ms_per_step'],3))"
done
