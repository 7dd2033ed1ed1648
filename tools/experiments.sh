#!/bin/bash
# experiments.sh NAME - the round-3 experiments of tools/README.md, one subcommand each (run on the GPU box through
# gpurun from the repository root; every step goes through tools/gpu.sh, which bounds it with a timeout).
#   groups            resident workgroups per CU of the search kernel 1 / 2 / 4 / 6: time and FETCH_SIZE (is the over-fetch L2 capacity?)
#   shared-threshold  c3 with 2 000 .. 8 000 reads, chunk per wave vs per workgroup
#   reserve           record slots a wave / workgroup reserves per atomic
#   group-out         per-wave output blocks vs workgroup-shared ones of 256 / 512 / 1 024 records
#   variants LIB...   c3 with each variant library (tools/build_variant.sh), twice
#   forest LIB...     one c5 batch with the classifier on the path, with each variant library
set -o pipefail
NAME=${1:?experiment}; shift
line() { python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], 'ms/step', round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['kernels_ms'].items() if v})" "$1" "$2"; }
case $NAME in
groups)
    for G in 1 2 4 6; do
        TAG=g$G bash tools/gpu.sh pmc c3 "FETCH_SIZE" seed_sliced --hook seed_groups_per_cu=$G --hook seed_shared=0 || exit 1
        line gpurun_out/g$G/pmc_bench.json "groups_per_cu $G"
    done;;
shared-threshold)
    for G in 2000 4000 6000 8000; do for S in 0 1; do
        TAG=sh_${G}_$S bash tools/gpu.sh bench c3 --guides $G --steps 5 --warmup 1 --hook seed_shared=$S | sed "s/^/guides $G shared $S: /" || exit 1
    done; done;;
reserve)
    for R in 64 128 256 512 1024; do
        TAG=res$R bash tools/gpu.sh bench c3 --steps 4 --warmup 1 --hook seed_reserve=$R | sed "s/^/reserve $R: /" || exit 1
    done;;
group-out)
    for rep in 1 2; do
        TAG=go_base bash tools/gpu.sh bench c3 --steps 4 --warmup 1 --hook seed_group_out=0 | sed "s/^/per-wave 128: /" || exit 1
        for R in 256 512 1024; do
            TAG=go_$R bash tools/gpu.sh bench c3 --steps 4 --warmup 1 --hook seed_group_out=1 --hook seed_reserve=$R | sed "s/^/group $R: /" || exit 1
        done
    done;;
variants)
    mkdir -p gpurun_out/sweep
    for rep in 1 2; do for L in varscot_amd/libvarscot_hip.so "$@"; do
        VSC_LIB_PATH=$PWD/$L timeout -k 10 300 python3 bench.py --workload c3 --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/sweep/v.json 2> gpurun_out/sweep/v.err || { tail -3 gpurun_out/sweep/v.err; exit 1; }
        line gpurun_out/sweep/v.json "$L"
    done; done;;
forest)
    mkdir -p gpurun_out/sweep
    for L in varscot_amd/libvarscot_hip.so "$@"; do
        VSC_LIB_PATH=$PWD/$L timeout -k 10 300 python3 bench.py --workload c5 --guides 10000 --steps 1 --warmup 1 --no-cpu-baseline --classify > gpurun_out/sweep/f.json 2> gpurun_out/sweep/f.err || { tail -3 gpurun_out/sweep/f.err; exit 1; }
        line gpurun_out/sweep/f.json "$L"
    done;;
*) echo "experiments.sh: unknown experiment $NAME" >&2; exit 2;;
esac
