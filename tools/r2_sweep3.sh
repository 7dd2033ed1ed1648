#!/bin/bash
# comparison specialised for the mismatch limit on the six-wave kernel, same box, two rounds
mkdir -p gpurun_out/r2x
timeout -k 10 600 env VSC_LIB_PATH=$PWD/varscot_amd/libvsc_pspec.so python -m pytest tests/test_gpu_parity.py -m gpu -q -x > gpurun_out/r2x/pytest_spec.log 2>&1; tail -1 gpurun_out/r2x/pytest_spec.log
for rep in 1 2; do
for v in base spec; do
LIB=$PWD/varscot_amd/libvsc_p$v.so; [ $v == base ] && LIB=$PWD/varscot_amd/libvarscot_hip.so
for W in c3 c2; do
VSC_LIB_PATH=$LIB timeout -k 10 300 python bench.py --workload $W --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r2x/sw3_$v.json 2> gpurun_out/r2x/sw3_$v.err || tail -3 gpurun_out/r2x/sw3_$v.err
python -c "
import json
d=json.load(open('gpurun_out/r2x/sw3_$v.json'))
print('$v', '$W', round(d['ms_per_step'],2), round(d['kernels_ms']['search'],2))"
done; done; done
