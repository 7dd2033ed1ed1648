#!/bin/bash
# Does the search kernel's over-fetch come from the L2 capacity?  Resident workgroups per CU 1 .. 6 (hook
# seed_groups_per_cu): kernel time and FETCH_SIZE per launch (one counter per pass).
for G in 1 2 4 6; do
  TAG=g$G bash tools/gpu.sh pmc c3 "FETCH_SIZE" seed_sliced --hook seed_groups_per_cu=$G || exit 1
  python3 -c "
import json; d=json.loads(open('gpurun_out/g$G/pmc_bench.json').read().strip().splitlines()[-1]); print('groups_per_cu $G search ms', d['kernels_ms']['search'])"
done
