#!/bin/bash
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_configs.py -q -m gpu -k "g4m_eight" 2>&1 | tail -8
