#!/bin/bash
# round 4, GPU call 1: stripe-kernel parity + first timings
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_stripe.py -x -q 2>&1 | tail -25 > gpurun_out/r4c1_pytest.txt
cat gpurun_out/r4c1_pytest.txt
MIDM_KERNELS=1,2,3,4,5 MIDM_SHAPES=4096x4096,4096x11008,11008x4096,4096x6144,4096x28672,14336x4096 timeout 600 python tests/dev/midm.py 64,96,128,160,192,256 > gpurun_out/r4c1_midm.txt 2>&1
cat gpurun_out/r4c1_midm.txt
