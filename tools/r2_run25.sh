#!/bin/bash
# several c3 bench processes in a row (run-to-run spread)
mkdir -p gpurun_out/r2u
for i in 1 2 3 4; do
timeout -k 10 300 python bench.py --workload c3 --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r2u/rep_$i.json 2> gpurun_out/r2u/rep_$i.err || tail -5 gpurun_out/r2u/rep_$i.err
python -c "
import json
d=json.load(open('gpurun_out/r2u/rep_$i.json'))
print($i, round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['kernels_ms'].items() if v})"
done
