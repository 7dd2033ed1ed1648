mkdir -p gpurun_out/sweep
for rep in 1 2; do
for v in varscot_hip vsc_nt vsc_grab16 vsc_grab4 vsc_res2 vsc_res5; do
  VSC_LIB_PATH=$PWD/varscot_amd/lib$v.so timeout -k 10 300 python3 bench.py --workload c3 --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/sweep/$v.json 2> gpurun_out/sweep/$v.err || { tail -3 gpurun_out/sweep/$v.err; exit 1; }
  python3 -c "
import json
d=json.loads(open('gpurun_out/sweep/$v.json').read().strip().splitlines()[-1]); print('$v', round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['kernels_ms'].items() if v})"
done; done
