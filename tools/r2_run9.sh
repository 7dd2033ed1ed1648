#!/bin/bash
mkdir -p gpurun_out/r2j
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_full_size.py -m gpu -q -x > gpurun_out/r2j/pytest.log 2>&1
tail -5 gpurun_out/r2j/pytest.log
for R in default 256; do
if [ $R = default ]; then unset VSC_SEED_RESERVE; else export VSC_SEED_RESERVE=$R; fi
timeout -k 10 300 python bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r2j/bench_c3_$R.json 2> gpurun_out/r2j/bench_c3_$R.err || tail -20 gpurun_out/r2j/bench_c3_$R.err
python - <<PY
import json
d=json.load(open("gpurun_out/r2j/bench_c3_$R.json"))
print("reserve=$R", d["ms_per_step"], d["kernels_ms"])
PY
done
unset VSC_SEED_RESERVE
timeout -k 10 300 python bench.py --workload c2 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r2j/bench_c2.json 2> gpurun_out/r2j/bench_c2.err
python -c "
import json
d=json.load(open('gpurun_out/r2j/bench_c2.json'))
print('c2', d['ms_per_step'], d['kernels_ms'])"
