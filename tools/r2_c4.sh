#!/bin/bash
mkdir -p gpurun_out/r2u
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_pipeline.py -m gpu -q -x > gpurun_out/r2u/pytest_c4.log 2>&1; tail -1 gpurun_out/r2u/pytest_c4.log
for W in c4 c2 c3; do
timeout -k 10 500 python bench.py --workload $W --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r2u/w_$W.json 2> gpurun_out/r2u/w_$W.err || tail -5 gpurun_out/r2u/w_$W.err
python -c "
import json
d=json.load(open('gpurun_out/r2u/w_$W.json'))
print('$W', round(d['value']), round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['kernels_ms'].items() if v})"
done
