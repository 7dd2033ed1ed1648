#!/bin/bash
mkdir -p gpurun_out/r2r
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > gpurun_out/r2r/pytest.log 2>&1
tail -3 gpurun_out/r2r/pytest.log
for W in c3 c2; do
timeout -k 10 600 python bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r2r/bench_$W.json 2> gpurun_out/r2r/bench_$W.err || tail -20 gpurun_out/r2r/bench_$W.err
python -c "
import json
d=json.load(open('gpurun_out/r2r/bench_$W.json'))
print('$W', d['ms_per_step'], d['value'], d['kernels_ms'])"
done
