#!/bin/bash
mkdir -p gpurun_out/r2x
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/cmp_probe.hip -o gpurun_out/r2x/cmp_probe && timeout -k 10 120 gpurun_out/r2x/cmp_probe
