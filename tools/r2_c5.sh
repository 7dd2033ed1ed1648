#!/bin/bash
mkdir -p gpurun_out/r2u
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_full_size.py -m gpu -q -x -k "scor or packed or c5 or chunk or stream or feature" > gpurun_out/r2u/pytest_c5.log 2>&1
tail -3 gpurun_out/r2u/pytest_c5.log
timeout -k 10 500 python bench.py --workload c5 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r2u/c5.json 2> gpurun_out/r2u/c5.err || tail -5 gpurun_out/r2u/c5.err
python -c "
import json
d=json.load(open('gpurun_out/r2u/c5.json'))
print(round(d['value']), round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['kernels_ms'].items() if v})"
