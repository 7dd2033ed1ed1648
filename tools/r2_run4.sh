#!/bin/bash
mkdir -p gpurun_out/r2d
VSC_DEBUG_SORT=2 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "test_sort_levels" -s > gpurun_out/r2d/sortlevels.log 2>&1
grep -E "device .* host|passed|failed" gpurun_out/r2d/sortlevels.log | head -30
timeout -k 10 600 python -m pytest tests -m gpu -q > gpurun_out/r2d/pytest.log 2>&1
tail -5 gpurun_out/r2d/pytest.log
