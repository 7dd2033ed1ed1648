#!/bin/bash
# counters of the packed scoring kernel on a c5 batch stream, two passes
set -o pipefail
mkdir -p gpurun_out/r2x
timeout -k 10 500 bash tools/pmc_probe2.sh c5 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" score_packed_kernel > gpurun_out/r2x/pmc_score_a.txt 2>&1 && \
timeout -k 10 500 bash tools/pmc_probe2.sh c5 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_WR" score_packed_kernel > gpurun_out/r2x/pmc_score_b.txt 2>&1
cat gpurun_out/r2x/pmc_score_a.txt gpurun_out/r2x/pmc_score_b.txt
