#!/bin/bash
# finalize time per record against the fill of its bins: the genome size sets the hits per (read, strand) and with
# them the records per bin (16 bins per group at these sizes)
mkdir -p gpurun_out/r2x
for B in 3000000000 3500000000 3900000000; do
timeout -k 10 300 python bench.py --workload c3 --bases $B --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r2x/fill_$B.json 2> gpurun_out/r2x/fill_$B.err || tail -3 gpurun_out/r2x/fill_$B.err
python -c "
import json
d=json.load(open('gpurun_out/r2x/fill_$B.json'))
h=d['config']['hits_per_step']; k=d['kernels_ms']
print($B, 'hits', h, 'per bin', round(h/(157*2048)), {a: round(b,2) for a,b in k.items() if b}, 'finalize ps/record', round(k['finalize']*1e9/h,2), 'sort ps/record', round(k['sort']*1e9/h,2), d.get('roofline_sort',{}).get('first_level_bin_bits'))"
done
