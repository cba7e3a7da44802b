#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
MIDM_KERNELS=2,3,6,7 MIDM_SHAPES=4096x11008,4096x28672,8192x8192,4096x6144,4096x4096,11008x4096 timeout 600 python tests/dev/midm.py 64,96,128,192,256 > gpurun_out/r4c5_midm.txt 2>&1
cat gpurun_out/r4c5_midm.txt
