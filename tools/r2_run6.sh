#!/bin/bash
mkdir -p gpurun_out/r2g
timeout -k 10 900 python -m pytest tests -m gpu -q -x --durations=8 > gpurun_out/r2g/pytest.log 2>&1
tail -25 gpurun_out/r2g/pytest.log
VSC_DEBUG_TIMING=1 timeout -k 10 600 python bench.py --workload c4 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r2g/bench_c4.json 2> gpurun_out/r2g/bench_c4.err || tail -20 gpurun_out/r2g/bench_c4.err
grep "vsc windows" gpurun_out/r2g/bench_c4.err | head; cat gpurun_out/r2g/bench_c4.json | python -c "import json,sys; d=json.load(sys.stdin); print(d['ms_per_step'], d['value'], d['config']['variant_genome'], d['kernels_ms'])"
timeout -k 10 600 python bench.py --workload c5 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r2g/bench_c5.json 2> gpurun_out/r2g/bench_c5.err || tail -20 gpurun_out/r2g/bench_c5.err
cat gpurun_out/r2g/bench_c5.json | python -c "import json,sys; d=json.load(sys.stdin); print(d['ms_per_step'], d['value'], d['config']['candidate_sites_per_s'], d['kernels_ms'])"
