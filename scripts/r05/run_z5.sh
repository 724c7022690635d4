#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_ref_lm.py -q -s -m gpu -k "stage_times and s2m" 2>&1 | grep -v amdgpu.ids | tail -20
