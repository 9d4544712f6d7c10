#!/bin/bash
# usage (GPU box, through gpurun): bash tools/pmc_cfgs.sh <tag> <kbench cfg...>
# kernel-trace stats + PMC passes (each in its own run, --kernel-trace only) around tools/kbench.py for the given configs;
# raw output under gpurun_out/pmc_<tag>_*; summarise with tools/summarize_cfgs.py <tag>.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pmc_${tag}_stats -- python3 $R/tools/kbench.py --cfg "$@" --iters 20 --no-check > $R/gpurun_out/pmc_${tag}_stats.log 2>&1 || exit 2
i=0
for pass in "FETCH_SIZE" "WRITE_SIZE" \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA" \
  "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "GRBM_GUI_ACTIVE GRBM_COUNT SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $R/gpurun_out/pmc_${tag}_p$i -- python3 $R/tools/kbench.py --cfg "$@" --iters 3 --no-check > $R/gpurun_out/pmc_${tag}_p$i.log 2>&1 || exit 3
done
echo ok
