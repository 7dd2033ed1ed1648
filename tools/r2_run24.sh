#!/bin/bash
mkdir -p gpurun_out/r2y
timeout -k 10 900 python -m pytest tests/test_full_size.py -m gpu -q -x -k "limit" > gpurun_out/r2y/pytest.log 2>&1
tail -15 gpurun_out/r2y/pytest.log
