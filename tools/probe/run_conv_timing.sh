# development: per-workgroup phase timing of the fp32 conv kernel under the tile / schedule switches
P=tools/probe/conv_timing
for shape in "8 52 128 256 3" "8 26 256 512 3" "8 13 512 1024 3" "8 52 256 128 1"; do
  for cfg in "Y3_PIPE=0 Y3_RSPLIT=0" "Y3_PIPE=1 Y3_RSPLIT=0" "Y3_PIPE=1 Y3_RSPLIT=1" "Y3_PIPE=0 Y3_RSPLIT=0 Y3_TILE=128,128,16" "Y3_PIPE=1 Y3_RSPLIT=0 Y3_TILE=128,128,16" "Y3_PIPE=1 Y3_RSPLIT=1 Y3_TILE=128,128,16"; do
    echo "=== $shape | $cfg"; env $cfg $P $shape || exit 1
  done
done
