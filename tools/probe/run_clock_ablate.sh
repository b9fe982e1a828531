# development: the clock the chip HOLDS under the fp32 conv kernel, per ablation of the K loop's memory
# activity (Y3_ABL: 1 no global loads, 2 no LDS stores, 4 no barrier, 7 all three, 8 split-K slabs with the default cache policy)
# and with split-K off; the same for the x3 kernel.  tools/probe/conv_timing: stamps of the last of 200 back-to-back launches.
P=tools/probe/conv_timing
for shape in "8 52 128 256 3" "8 26 256 512 3" "8 13 512 1024 3" "8 52 256 128 1"; do
  for abl in 0 1 2 4 7 8; do
    echo "=== fp32 $shape | Y3_ABL=$abl"; Y3_ABL=$abl $P $shape 0 | grep -E "^layer|under ablation|shader clock|main loop|whole workgroup" || exit 1
  done
  echo "=== fp32 $shape | split-K off"; Y3_SPLITK_MINK=100000000 Y3_RSPLIT=0 $P $shape 0 | grep -E "^layer|shader clock|main loop|whole workgroup" || exit 1
done
for shape in "8 52 128 256 3" "8 26 256 512 3" "8 13 512 1024 3"; do
  for abl in 0 1 2 4 7; do
    echo "=== x3 $shape | Y3_ABL=$abl"; Y3_ABL=$abl $P $shape 1 | grep -E "^layer|under ablation|shader clock|prologue|main loop|epilogue|whole workgroup|MFMA floor|per CU" || exit 1
  done
done
