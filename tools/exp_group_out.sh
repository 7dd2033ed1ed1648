#!/bin/bash
# workgroup-shared output blocks (hook seed_group_out) at c3: default blocks (512 records) and other sizes, against the
# per-wave blocks of the default build
for rep in 1 2; do
  TAG=go_base bash tools/gpu.sh bench c3 --steps 4 --warmup 1 | sed "s/^/per-wave 128: /" || exit 1
  for R in 256 512 1024; do
    TAG=go_$R bash tools/gpu.sh bench c3 --steps 4 --warmup 1 --hook seed_group_out=1 --hook seed_reserve=$R | sed "s/^/group $R: /" || exit 1
  done
done
