#!/bin/bash
# usage: tools/pmc2.sh <tag> "<counters>" <kbench cfg...>  -- one PMC pass on the GPU box
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; ctr=$2; shift; shift
cd /tmp
rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $R/gpurun_out/pmc_${tag}_a -- python3 $R/tools/kbench.py --cfg "$@" --iters 3 --no-check > $R/gpurun_out/pmc_${tag}_a.log 2>&1
echo done $?
