#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_stripe.py -x -q 2>&1 | tail -15 > gpurun_out/r4c2_pytest.txt
cat gpurun_out/r4c2_pytest.txt
timeout 300 python tests/dev/stripe_stamps.py 4096x4096,4096x11008,11008x4096 64,128 > gpurun_out/r4c2_stamps.txt 2>&1
STRIPE_WT=1 timeout 300 python tests/dev/stripe_stamps.py 4096x4096 128 >> gpurun_out/r4c2_stamps.txt 2>&1
cat gpurun_out/r4c2_stamps.txt
