#!/bin/bash
mkdir -p gpurun_out/r2s
run() {
timeout -k 10 600 python bench.py --workload c3 --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r2s/b_$1.json 2> gpurun_out/r2s/b_$1.err || tail -5 gpurun_out/r2s/b_$1.err
python -c "
import json
d=json.load(open('gpurun_out/r2s/b_$1.json'))
print('$1', round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['kernels_ms'].items() if v})"
}
run base1
run base2
VSC_SORT_XCD=0 run noxcd
VSC_SORT_PAD_KB=4 run pad4k
VSC_SORT_PAD_KB=68 run pad68k
VSC_SORT_PAD_KB=1028 run pad1m
VSC_SORT_PAD_KB=33000 run pad33m
