#!/bin/bash
# round 5, run z4: the reference's own stage timers on the MI355X beside this library's (tests/test_ref_lm.py)
cd /root/repo
timeout 600 python -m pytest tests/test_ref_lm.py -q -s -m gpu -k "stage_times" 2>&1 | grep -v amdgpu.ids | tail -40
