#!/bin/bash
# resolve threshold of the six-wave search kernel, variant libraries, one box
mkdir -p gpurun_out/r2x
for rep in 1 2; do
for v in base r3 r5 r6; do
LIB=$PWD/varscot_amd/libvsc_p$v.so; [ $v == base ] && LIB=$PWD/varscot_amd/libvarscot_hip.so
VSC_LIB_PATH=$LIB timeout -k 10 300 python bench.py --workload c3 --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r2x/sw2_$v.json 2> gpurun_out/r2x/sw2_$v.err || tail -3 gpurun_out/r2x/sw2_$v.err
python -c "
import json
d=json.load(open('gpurun_out/r2x/sw2_$v.json'))
print('$v', round(d['ms_per_step'],2), round(d['kernels_ms']['search'],2))"
done; done
