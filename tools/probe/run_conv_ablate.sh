# development: lone-workgroup K-loop cost with parts of the step ablated (tools/probe/conv_timing, Y3_ABL)
P=tools/probe/conv_timing
export Y3_SPLITK_MINK=100000000 Y3_RSPLIT=0
for pipe in 0 2; do for abl in 0 1 2 3 4 7; do
  echo "=== 1x1 256->128 M=21632, tile 128x128 (169 WGs, 1 per CU), PIPE=$pipe ABL=$abl"; Y3_ABL=$abl Y3_PIPE=$pipe Y3_TILE=128,128,16 $P 8 52 256 128 1 || exit 1
done; done
for pipe in 0 2; do for abl in 0 1 2 3 4 7; do
  echo "=== 3x3 128->128 M=32768, tile 128x128 (256 WGs, 1 per CU), PIPE=$pipe ABL=$abl"; Y3_ABL=$abl Y3_PIPE=$pipe Y3_TILE=128,128,16 $P 2 128 128 128 3 || exit 1
done; done
