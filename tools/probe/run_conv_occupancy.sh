# development: main-loop cycles per K step of the fp32 conv kernel at 1 / 2 / 3 workgroups per CU (grid = 256 x r whole tiles,
# no split-K), for the two loop schedules.  MFMA floor per step: 64x128 tile = 16 MFMAs x 64 = 1024 cycles per wave,
# 128x128 = 2048; a CU with r workgroups needs r times that per step-round.
P=tools/probe/conv_timing
export Y3_SPLITK_MINK=100000000 Y3_RSPLIT=0
for pipe in 0 2; do
  for r in 1 2 3; do
    echo "=== tile 64x128, $r WG/CU, PIPE=$pipe"; Y3_PIPE=$pipe Y3_TILE=64,128,16 $P $r 128 128 128 3 || exit 1
  done
  for r in 2 4 6; do
    echo "=== tile 128x128, $((r/2)) WG/CU, PIPE=$pipe"; Y3_PIPE=$pipe Y3_TILE=128,128,16 $P $r 128 128 128 3 || exit 1
  done
done
for st in 0; do
  echo "=== stagger $st: 8x52x52 128->256 (64x128 tiles, 676 WGs), PIPE=0"; Y3_STAGGER=$st Y3_PIPE=0 $P 8 52 128 256 3 || exit 1
  echo "=== stagger $st: same, PIPE=2"; Y3_STAGGER=$st Y3_PIPE=2 $P 8 52 128 256 3 || exit 1
done
