#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $R/gpurun_out/pmc_r01_dense16 -- python3 $R/tools/kbench.py --cfg f16:4096:4096:4096 --iters 5 > $R/gpurun_out/pmc_r01_dense16.log 2>&1
echo done $?
