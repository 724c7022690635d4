mkdir -p gpurun_out
for lib in "" cuda-bundle-adjustment_amd/csrc/libexp_touch.so; do
  echo "== CUBA_HIP_LIB_F64=$lib" >> gpurun_out/r03k_block_touch.txt
  for shape in kitti00 g4m; do
    (CUBA_HIP_LIB_F64=$PWD/$lib; [ -z "$lib" ] && unset CUBA_HIP_LIB_F64 || export CUBA_HIP_LIB_F64; timeout 200 python scripts/kernel_times.py $shape 2>&1 | tail -1) >> gpurun_out/r03k_block_touch.txt
  done
done
cat gpurun_out/r03k_block_touch.txt
