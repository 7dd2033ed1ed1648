#!/bin/bash
# round 2, run 1: parity of the bin sort, then first timings
set -o pipefail
mkdir -p gpurun_out/r2a
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r2a/pytest.log 2>&1
rc=$?
tail -15 gpurun_out/r2a/pytest.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r2a/bench_c3.json 2> gpurun_out/r2a/bench_c3.err || { tail -20 gpurun_out/r2a/bench_c3.err; exit 1; }
cat gpurun_out/r2a/bench_c3.json
timeout -k 10 300 python bench.py --workload c2 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r2a/bench_c2.json 2> gpurun_out/r2a/bench_c2.err || { tail -20 gpurun_out/r2a/bench_c2.err; exit 1; }
cat gpurun_out/r2a/bench_c2.json
