#!/bin/bash
# usage (on the GPU box, through gpurun): bash tools/profile_round.sh r02  -- bench line + rocprofv3 kernel stats + PMC passes
T=${1:-r04}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R && timeout -k 10 500 python3 bench.py --steps 100 --warmup 20 > $R/gpurun_out/bench_$T.json 2> $R/gpurun_out/bench_$T.err || exit 1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$T -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-side-configs > $R/gpurun_out/prof_$T.log 2>&1 || exit 2
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $R/gpurun_out/pmc_${T}_$tag -- python3 $R/tools/kbench.py --cfg q4_0:4096:4096:4096 q4_0:4096:4096:1:32 --iters 3 --no-check > $R/gpurun_out/pmc_${T}_$tag.log 2>&1 || exit 3
done
# dense f16 path: MFMA utilisation of dense16_kernel (SQ_VALU_MFMA_BUSY_CYCLES / 4 SIMD-cycles vs SQ_BUSY...)
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $R/gpurun_out/pmc_${T}_dense16 -- python3 $R/tools/kbench.py --cfg f16:4096:4096:4096 --iters 5 > $R/gpurun_out/pmc_${T}_dense16.log 2>&1 || exit 4
# dense f32 path (operands as three bf16 pieces each, six bf16 MFMAs per product): the same counters for dense32s_kernel
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $R/gpurun_out/pmc_${T}_dense32 -- python3 $R/tools/kbench.py --cfg f32:4096:4096:4096 --iters 5 > $R/gpurun_out/pmc_${T}_dense32.log 2>&1 || exit 4
echo ok
# round 3: the other BASELINE configs, one set of passes each (summaries: tools/summarize_cfgs.py <tag> <name> <kernel regex>)
bash $R/tools/pmc_cfgs.sh ${T}c3 q4_0:4096:4096:512 || exit 5
bash $R/tools/pmc_cfgs.sh ${T}c4q8 q8_0:4096:11008:512 || exit 6
bash $R/tools/pmc_cfgs.sh ${T}c4q5 q5_0:4096:11008:512 || exit 7
bash $R/tools/pmc_cfgs.sh ${T}c5 q4_0:32000:4096:512 || exit 8
# round 4: the min-term form of K3p (Q5_1 = the planar form a Q5_K weight lives in)
bash $R/tools/pmc_cfgs.sh ${T}c4q51 q5_1:4096:11008:512 || exit 9
echo ok-configs
