#!/bin/bash
mkdir -p gpurun_out/r2u
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > gpurun_out/r2u/pytest.log 2>&1
tail -3 gpurun_out/r2u/pytest.log
run() {
timeout -k 10 600 python bench.py --workload c3 --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r2u/b_$1.json 2> gpurun_out/r2u/b_$1.err || tail -5 gpurun_out/r2u/b_$1.err
python -c "
import json
d=json.load(open('gpurun_out/r2u/b_$1.json'))
print('$1', round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['kernels_ms'].items() if v})"
}
run w6c
run w6cb
