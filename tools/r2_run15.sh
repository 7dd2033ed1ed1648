#!/bin/bash
mkdir -p gpurun_out/r2p
timeout -k 10 1000 python -m pytest tests -m gpu -q -x -s > gpurun_out/r2p/pytest.log 2>&1
grep -E "repeat-rich|passed|failed|Error|assert" gpurun_out/r2p/pytest.log | tail -8
timeout -k 10 600 python bench.py --workload c4 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r2p/bench_c4.json 2> gpurun_out/r2p/bench_c4.err || tail -20 gpurun_out/r2p/bench_c4.err
python -c "
import json
d=json.load(open('gpurun_out/r2p/bench_c4.json'))
print('c4', d['ms_per_step'], d['value'], d['kernels_ms'])"
