#!/bin/bash
# a second validation campaign with other seeds than validate_round.sh (GPU box): randomised sweeps only
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out && rm -f gpurun_out/val2_*.log
timeout -k 10 500 python tools/sweep_parity.py 31 400 > gpurun_out/val2_sweep_parity_a.log 2>&1 || exit 2
timeout -k 10 500 python tools/sweep_parity.py 32 400 > gpurun_out/val2_sweep_parity_b.log 2>&1 || exit 3
timeout -k 10 300 python tools/sweep_push.py 9 200 > gpurun_out/val2_sweep_push.log 2>&1 || exit 4
timeout -k 10 300 python tools/sweep_epilogue.py 9 200 > gpurun_out/val2_sweep_epi.log 2>&1 || exit 5
timeout -k 10 900 python tools/sweep_k3p.py 41 200 > gpurun_out/val2_sweep_k3p.log 2>&1 || exit 6
timeout -k 10 600 python tools/stress_k3p.py > gpurun_out/val2_stress.log 2>&1 || exit 7
echo ok
