mkdir -p gpurun_out; f=gpurun_out/r03v_schur_u.txt; rm -f $f
for shape in kitti00 g4m; do
  (timeout 200 python scripts/kernel_times.py $shape 2>&1 | tail -1) >> $f
  (timeout 200 python scripts/kernel_times.py $shape schur_u=1 2>&1 | tail -1) >> $f
  (CUBA_HIP_SCHUR_U_SEPARATE=1 timeout 200 python scripts/kernel_times.py $shape schur_u=1 2>&1 | tail -1) >> $f
  (timeout 200 python scripts/kernel_times.py $shape schur_u=1 mixed_precision=1 2>&1 | tail -1) >> $f
  (CUBA_HIP_SCHUR_U_SEPARATE=1 timeout 200 python scripts/kernel_times.py $shape schur_u=1 mixed_precision=1 2>&1 | tail -1) >> $f
done
cat $f
