#!/bin/bash
mkdir -p gpurun_out/r2n
for SH in 28 25 23 22; do
VSC_SCORE_SLICES=1 VSC_SCORE_SLICE_SHIFT=$SH timeout -k 10 600 python bench.py --workload c5 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r2n/bench_c5_$SH.json 2> gpurun_out/r2n/bench_c5_$SH.err || tail -20 gpurun_out/r2n/bench_c5_$SH.err
cat gpurun_out/r2n/bench_c5_$SH.json | python -c "import json,sys; d=json.load(sys.stdin); print('shift=$SH', d['ms_per_step'], d['value'], d['kernels_ms'])"
done
