set -o pipefail
python3 -c "import __graft_entry__ as g; g.smoke()" || exit 1
bash tools/gpu.sh tests || exit 1
mkdir -p gpurun_out/final
timeout -k 10 300 python3 bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err || exit 1
timeout -k 10 300 python3 bench.py --workload c5 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/final/bench_c5.json 2> gpurun_out/final/bench_c5.err || exit 1
timeout -k 10 400 python3 bench.py --workload c5 --steps 1 --warmup 1 --no-cpu-baseline --classify > gpurun_out/final/bench_c5_classify.json 2> gpurun_out/final/bench_c5_classify.err || exit 1
for f in default c5 c5_classify; do python3 -c "
import json
d=json.loads(open('gpurun_out/final/bench_$f.json').read().strip().splitlines()[-1]); print('$f', round(d['ms_per_step'],2), round(d['value'],1), {k: round(v,2) for k,v in d['kernels_ms'].items() if v}, d['roofline']['frac'], d['roofline']['traffic'])"; done
