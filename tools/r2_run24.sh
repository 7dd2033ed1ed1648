#!/bin/bash
mkdir -p gpurun_out/r2y
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_tools.py tests/test_pipeline.py -m gpu -q -x -k "index or several" > gpurun_out/r2y/pytest.log 2>&1
tail -15 gpurun_out/r2y/pytest.log
