#!/bin/bash
# the multi-rank bench paths on a one-GPU box: the RCCL exchange with one rank, and 2 / 4 ranks rehearsed on cuda:0 (gloo)
mkdir -p gpurun_out/r2d
timeout -k 10 300 python bench.py --workload c3 --force-dist --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2d/force.json 2> gpurun_out/r2d/force.err || tail -5 gpurun_out/r2d/force.err
python -c "
import json; d=json.load(open('gpurun_out/r2d/force.json')); print('force-dist', round(d['ms_per_step'],2), d['config']['multi_gpu_path'], d['config']['exchange'])"
for N in 2 4; do
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus $N --rehearse --workload c2 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2d/reh_$N.json 2> gpurun_out/r2d/reh_$N.err || tail -5 gpurun_out/r2d/reh_$N.err
python -c "
import json; d=json.loads(open('gpurun_out/r2d/reh_$N.json').read().strip().splitlines()[-1]); print('rehearse', $N, round(d['ms_per_step'],2), d['n_gpus'], d['config']['multi_gpu_path'], d['config']['hits_per_step'])"
done
