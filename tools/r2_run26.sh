#!/bin/bash
# the driver's command (default flags, CPU baseline included) several times: spread of the kernel times
mkdir -p gpurun_out/r2u
for i in 1 2 3; do
timeout -k 10 300 python bench.py > gpurun_out/r2u/def_$i.json 2> gpurun_out/r2u/def_$i.err || tail -5 gpurun_out/r2u/def_$i.err
python -c "
import json
d=json.load(open('gpurun_out/r2u/def_$i.json'))
print($i, round(d['value']), round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['kernels_ms'].items() if v})"
done
