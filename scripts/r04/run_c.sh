#!/bin/bash
out=gpurun_out; mkdir -p $out
( time timeout 900 python scripts/r04/forcing_ab.py kitti07 kitti00 s2m g4m ) > $out/r04c_forcing.log 2>&1
cat $out/r04c_forcing.log | cut -c1-330
