#!/bin/bash
# round 5, run y: reduction in parts + rank uploads (tests/test_dist.py), then the whole dist file
cd /root/repo
timeout 900 python -m pytest tests/test_dist.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r05y_dist_tests.txt
cat gpurun_out/r05y_dist_tests.txt
