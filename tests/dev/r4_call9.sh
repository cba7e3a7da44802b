#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
MIDM_KERNELS=2,3,41,42,81 MIDM_SHAPES=4096x4096,4096x11008,11008x4096,4096x28672 timeout 600 python tests/dev/midm.py 64,128,192,256 > gpurun_out/r4c9_midm.txt 2>&1
cat gpurun_out/r4c9_midm.txt
