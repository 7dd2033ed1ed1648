#!/bin/bash
# the whole GPU suite, as the driver runs it at round end
mkdir -p gpurun_out/r2z
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > gpurun_out/r2z/pytest_gpu.log 2>&1
tail -5 gpurun_out/r2z/pytest_gpu.log
