#!/bin/bash
# usage: tools/ab_env.sh VAR "v1 v2 ..." cfg...   -- tools/ab_kernels.py under each value of an A/B environment switch
V=$1; VALS=$2; shift 2
for v in $VALS; do echo "== $V=$v"; env $V=$v python tools/ab_kernels.py "$@" || exit 1; done
