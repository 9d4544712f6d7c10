#!/bin/bash
# the validation campaign behind a kernel change (GPU box): the gpu test suite, then the randomised sweeps; every step must pass
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/val_tests.log 2>&1 || { echo "tests FAILED" >> gpurun_out/val_tests.log; exit 1; }
timeout -k 10 400 python tools/sweep_parity.py 11 300 > gpurun_out/val_sweep_parity_a.log 2>&1 || exit 2
timeout -k 10 400 python tools/sweep_parity.py 12 300 > gpurun_out/val_sweep_parity_b.log 2>&1 || exit 3
timeout -k 10 300 python tools/sweep_push.py 5 200 > gpurun_out/val_sweep_push.log 2>&1 || exit 4
timeout -k 10 300 python tools/sweep_epilogue.py 5 200 > gpurun_out/val_sweep_epi.log 2>&1 || exit 5
timeout -k 10 600 python tools/sweep_k3p.py 7 150 > gpurun_out/val_sweep_k3p.log 2>&1 || exit 6
echo ok
