#!/bin/bash
mkdir -p gpurun_out/r2t
run() {
timeout -k 10 600 python bench.py --workload c3 --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r2t/b_$1.json 2> gpurun_out/r2t/b_$1.err || tail -5 gpurun_out/r2t/b_$1.err
python -c "
import json
d=json.load(open('gpurun_out/r2t/b_$1.json'))
print('$1', round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['kernels_ms'].items() if v})"
}
run base
VSC_INDEX_KEEP_TEMP=1 run keep
