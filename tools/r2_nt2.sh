#!/bin/bash
# nontemporal loads of the records in hist / partition / finalize (variant library), same box
mkdir -p gpurun_out/r2x
for rep in 1 2 3; do for v in base nt; do
LIB=$PWD/varscot_amd/libvsc_pnt.so; [ $v == base ] && LIB=$PWD/varscot_amd/libvarscot_hip.so
VSC_LIB_PATH=$LIB timeout -k 10 300 python bench.py --workload c3 --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r2x/nt_$v.json 2> gpurun_out/r2x/nt_$v.err || tail -3 gpurun_out/r2x/nt_$v.err
python -c "
import json
d=json.load(open('gpurun_out/r2x/nt_$v.json'))
print('$v c3', round(d['ms_per_step'],2), {k: round(x,2) for k,x in d['kernels_ms'].items() if x})"
done; done
