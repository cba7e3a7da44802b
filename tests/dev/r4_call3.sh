#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_stripe.py -x -q 2>&1 | tail -5 > gpurun_out/r4c3_pytest.txt
cat gpurun_out/r4c3_pytest.txt
{
timeout 300 python tests/dev/stripe_stamps.py 4096x4096,4096x11008,11008x4096 128
echo "##### ablation: every second A fragment read dropped"
GPTQHIP_LIB=$PWD/tests/dev/ablate/libgptqhip_stamps_ablreads.so timeout 300 python tests/dev/stripe_stamps.py 4096x4096,4096x11008 128
echo "##### ablation: no dequant VALU"
GPTQHIP_LIB=$PWD/tests/dev/ablate/libgptqhip_stamps_abldeq.so timeout 300 python tests/dev/stripe_stamps.py 4096x4096,4096x11008 128
} > gpurun_out/r4c3_stamps.txt 2>&1
cat gpurun_out/r4c3_stamps.txt
MIDM_KERNELS=2,3 MIDM_SHAPES=4096x4096,4096x11008,11008x4096 timeout 600 python tests/dev/midm.py 64,128,192 > gpurun_out/r4c3_midm.txt 2>&1
cat gpurun_out/r4c3_midm.txt
