#!/bin/bash
mkdir -p gpurun_out/r2fuzz
timeout -k 10 ${2:-500} python tests/fuzz_gpu.py ${1:-150} ${3:-20000} > gpurun_out/r2fuzz/fuzz_$3.log 2>&1
tail -6 gpurun_out/r2fuzz/fuzz_$3.log
