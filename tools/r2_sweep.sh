#!/bin/bash
# search-kernel constants, one variant library each (built out of tree): c3 search time
mkdir -p gpurun_out/r2x
for v in base g16 g4 r2 c64 c16; do
LIB=$PWD/varscot_amd/libvsc_p$v.so; [ $v == base ] && LIB=$PWD/varscot_amd/libvarscot_hip.so
VSC_LIB_PATH=$LIB timeout -k 10 300 python bench.py --workload c3 --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r2x/sw_$v.json 2> gpurun_out/r2x/sw_$v.err || tail -3 gpurun_out/r2x/sw_$v.err
python -c "
import json
d=json.load(open('gpurun_out/r2x/sw_$v.json'))
print('$v', round(d['ms_per_step'],2), {k: round(x,2) for k,x in d['kernels_ms'].items() if x})"
done
