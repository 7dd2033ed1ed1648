#!/bin/bash
# Where does sharing a chunk between the four waves of a workgroup start to pay?  c3 with fewer reads (reads per bucket
# = reads x 633 / 49152), both modes forced by the hook.
for G in 2000 4000 6000 8000; do for S in 0 1; do
  TAG=sh_${G}_$S bash tools/gpu.sh bench c3 --guides $G --steps 5 --warmup 1 --hook seed_shared=$S | sed "s/^/guides $G shared $S: /" || exit 1
done; done
