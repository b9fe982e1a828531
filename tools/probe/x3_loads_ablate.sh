# development: which global loads of the x3 patch kernel's K loop cost the time -- activations (first-touch L2 misses) or weights
P=tools/probe/conv_timing
for shape in "8 52 128 256 3" "8 26 256 512 3" "8 13 512 1024 3"; do
  for abl in 0 1 16 32; do
    echo "=== x3 $shape | Y3_ABL=$abl"; Y3_ABL=$abl $P $shape 1 | grep -E "^layer|under ablation|shader clock|prologue|main loop|epilogue|whole workgroup|MFMA floor|per CU" || exit 1
  done
done
