#!/bin/bash
mkdir -p gpurun_out/r2c
VSC_DEBUG_SORT=2 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "test_sort_levels and scan and 256" -s > gpurun_out/r2c/sortlevels.log 2>&1
grep -E "vsc sort|passed|failed" gpurun_out/r2c/sortlevels.log | head -80
