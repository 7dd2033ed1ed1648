#!/bin/bash
mkdir -p gpurun_out/r2m
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_full_size.py -m gpu -q -x -k "several_passes or c5" > gpurun_out/r2m/pytest.log 2>&1
tail -6 gpurun_out/r2m/pytest.log
for S in 1 0; do
VSC_SCORE_SLICES=$S timeout -k 10 600 python bench.py --workload c5 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r2m/bench_c5_$S.json 2> gpurun_out/r2m/bench_c5_$S.err || tail -20 gpurun_out/r2m/bench_c5_$S.err
cat gpurun_out/r2m/bench_c5_$S.json | python -c "import json,sys; d=json.load(sys.stdin); print('slices=$S', d['ms_per_step'], d['value'], d['kernels_ms'])"
done
