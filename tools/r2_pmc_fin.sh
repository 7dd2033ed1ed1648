#!/bin/bash
# counters of the finalize kernel, two passes
set -o pipefail
mkdir -p gpurun_out/r2x
timeout -k 10 400 bash tools/pmc_probe2.sh c3 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" bin_finalize_kernel > gpurun_out/r2x/pmc_fin_a.txt 2>&1 && \
timeout -k 10 400 bash tools/pmc_probe2.sh c3 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" bin_finalize_kernel > gpurun_out/r2x/pmc_fin_b.txt 2>&1
cat gpurun_out/r2x/pmc_fin_a.txt gpurun_out/r2x/pmc_fin_b.txt
