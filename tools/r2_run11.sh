#!/bin/bash
mkdir -p gpurun_out/r2l
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r2l/pytest.log 2>&1
tail -6 gpurun_out/r2l/pytest.log
timeout -k 10 600 python bench.py --workload c5 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r2l/bench_c5.json 2> gpurun_out/r2l/bench_c5.err || tail -20 gpurun_out/r2l/bench_c5.err
cat gpurun_out/r2l/bench_c5.json | python -c "import json,sys; d=json.load(sys.stdin); print(d['ms_per_step'], d['value'], d['config']['candidate_sites_per_s'], d['kernels_ms'])"
