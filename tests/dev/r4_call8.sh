#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_stripe.py -x -q 2>&1 | tail -5 > gpurun_out/r4c8_pytest.txt
cat gpurun_out/r4c8_pytest.txt
timeout 300 python tests/dev/stripe_stamps.py 4096x4096,11008x4096 128,128 > gpurun_out/r4c8_stamps.txt 2>&1
cat gpurun_out/r4c8_stamps.txt
MIDM_KERNELS=1,2,3 MIDM_SHAPES=4096x4096,4096x11008,11008x4096,4096x6144,14336x4096,4096x28672,8192x8192 timeout 600 python tests/dev/midm.py 48,64,80,96,128,160,192,256 > gpurun_out/r4c8_midm.txt 2>&1
cat gpurun_out/r4c8_midm.txt
