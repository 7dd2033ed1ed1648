#!/bin/bash
# search kernel without the register prefetch of the next chunk's block (93 instead of 125 VGPRs) at 4 and 5 waves per SIMD
mkdir -p gpurun_out/r2x
run() {
LIB=$PWD/varscot_amd/$2; 
VSC_SEED_GROUPS_PER_CU=$3 VSC_LIB_PATH=$LIB timeout -k 10 300 python bench.py --workload $4 --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r2x/wv_$1.json 2> gpurun_out/r2x/wv_$1.err || tail -3 gpurun_out/r2x/wv_$1.err
python -c "
import json
d=json.load(open('gpurun_out/r2x/wv_$1.json'))
print('$1', '$4', round(d['ms_per_step'],2), {k: round(x,2) for k,x in d['kernels_ms'].items() if x}, d['config']['hits_per_step'])"
}
run base libvarscot_hip.so 4 c3
run nopf4 libvsc_wv1.so 4 c3
run nopf_r2_4 libvsc_wv3.so 4 c3
run nopf_r2_5 libvsc_wv2.so 5 c3
run base_c2 libvarscot_hip.so 4 c2
run nopf4_c2 libvsc_wv1.so 4 c2
run nopf_r2_5_c2 libvsc_wv2.so 5 c2
