#!/bin/bash
out=gpurun_out; mkdir -p $out
( for cap in 64 128; do TILES_TRUNCATE=$cap timeout 600 python scripts/experiments/r04_block_pass_tiles.py kitti00; done ) > $out/r04j_truncate.txt 2>&1
grep -v amdgpu $out/r04j_truncate.txt | cut -c1-250
