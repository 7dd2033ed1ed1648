#!/bin/bash
mkdir -p gpurun_out/r2h
timeout -k 10 1000 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/r2h/pytest.log 2>&1
tail -30 gpurun_out/r2h/pytest.log
timeout -k 10 600 python bench.py --workload c3 --steps 3 --warmup 1 > gpurun_out/r2h/bench_c3.json 2> gpurun_out/r2h/bench_c3.err || tail -20 gpurun_out/r2h/bench_c3.err
cat gpurun_out/r2h/bench_c3.json | python -c "import json,sys; d=json.load(sys.stdin); print(d['ms_per_step'], d['value'], d['kernels_ms']); print(json.dumps(d['cpu_baseline'], indent=1))"
