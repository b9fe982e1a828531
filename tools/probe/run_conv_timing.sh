# development: per-workgroup phase timing of the fp32 conv kernel (conv.hip built with -DY3_TIMING -DY3_DEV)
P=tools/probe/conv_timing
for shape in "8 52 128 256 3" "8 52 256 128 1" "8 26 512 256 1" "8 13 1024 512 1"; do
  for cfg in "Y3_RSPLIT=1" "Y3_RSPLIT=1 Y3_TILE=128,64,16" "Y3_RSPLIT=1 Y3_TILE=64,128,16"; do
    echo "=== $shape | $cfg"; env $cfg $P $shape || exit 1
  done
done
