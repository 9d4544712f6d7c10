#!/bin/bash
# developer experiment (on the GPU box, through gpurun): PMC passes over the batched-decode form (gemm_qmx.hip K3s), 4096 x 4096 x 32,
# weights rotating over 32 copies -> gpurun_out/pmc_k3s_*/ ; summary: python tools/pmc_summary.py gpurun_out/pmc_k3s_*
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SALU" "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $R/gpurun_out/pmc_k3s_$tag -- python3 $R/tools/kbench.py --cfg q4_0:4096:4096:32:32 --iters 3 --no-check > $R/gpurun_out/pmc_k3s_$tag.log 2>&1 || exit 3
done
echo ok
