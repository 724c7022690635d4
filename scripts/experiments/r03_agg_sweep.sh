mkdir -p gpurun_out
(timeout 250 python scripts/option_ab_seq.py g4m "" pcg_aggregate=40 pcg_aggregate=48 pcg_aggregate=64 2>&1 | tail -4) > gpurun_out/r03i_agg_sweep.txt
(timeout 200 python scripts/option_ab_seq.py s2m "" pcg_aggregate=28 pcg_aggregate=32 pcg_aggregate=36 pcg_aggregate=44 2>&1 | tail -5) >> gpurun_out/r03i_agg_sweep.txt
(timeout 100 python scripts/option_ab_seq.py kitti00 "" pcg_aggregate=16 pcg_aggregate=20 pcg_aggregate=28 2>&1 | tail -4) >> gpurun_out/r03i_agg_sweep.txt
(timeout 100 python scripts/option_ab_seq.py kitti07 "" pcg_aggregate=4 pcg_aggregate=6 pcg_aggregate=12 2>&1 | tail -4) >> gpurun_out/r03i_agg_sweep.txt
(timeout 100 python scripts/option_ab_seq.py kitti00 "" coarse_overlap_period=1 coarse_overlap_period=3 2>&1 | tail -3) >> gpurun_out/r03i_agg_sweep.txt
cat gpurun_out/r03i_agg_sweep.txt
