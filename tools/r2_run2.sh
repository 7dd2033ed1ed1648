#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2b
VSC_DEBUG_SORT=1 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "test_sort_levels" -s > gpurun_out/r2b/sortlevels.log 2>&1
grep -E "passed|failed|FAILED" gpurun_out/r2b/sortlevels.log | tail -12
timeout -k 10 600 python -m pytest tests -m gpu -q -k "not test_sort_levels" > gpurun_out/r2b/pytest.log 2>&1
tail -12 gpurun_out/r2b/pytest.log
timeout -k 10 300 python bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r2b/bench_c3.json 2> gpurun_out/r2b/bench_c3.err || tail -20 gpurun_out/r2b/bench_c3.err
cat gpurun_out/r2b/bench_c3.json
timeout -k 10 300 python bench.py --workload c2 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r2b/bench_c2.json 2> gpurun_out/r2b/bench_c2.err || tail -20 gpurun_out/r2b/bench_c2.err
cat gpurun_out/r2b/bench_c2.json
