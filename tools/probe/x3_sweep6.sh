P=tools/probe/conv_timing
L=gpurun_out/r04_x3p_3.log
: > $L
for shape in "8 26 256 512 3" "8 52 128 256 3"; do
  for abl in 0 1 16 2 18; do
  echo "=== x3p $shape | Y3_ABL=$abl" >> $L; Y3_ABL=$abl $P $shape 1 >> $L 2>&1 || exit 1
  done
done
grep -E "^===|under abl|shader clock|prologue|main loop|epilogue  " $L
