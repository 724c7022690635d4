#!/bin/bash
cd /root/repo
timeout 600 python -m pytest tests/test_dist.py -q -m gpu -k "degenerate" 2>&1 | grep -E "passed|failed|Error|assert" | tail -12
