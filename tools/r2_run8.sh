#!/bin/bash
mkdir -p gpurun_out/r2i
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "multi_device or several_passes" > gpurun_out/r2i/pytest.log 2>&1
tail -30 gpurun_out/r2i/pytest.log
