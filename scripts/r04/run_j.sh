#!/bin/bash
out=gpurun_out; mkdir -p $out

( time timeout 1200 python scripts/experiments/r04_block_pass_tiles.py kitti00 ) > $out/r04j_block_pass_tiles.txt 2>&1
cat $out/r04j_block_pass_tiles.txt | cut -c1-330
