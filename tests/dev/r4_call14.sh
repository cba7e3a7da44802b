#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
{
echo "## prefill kernel: group constants expanded once per chunk (shipped) vs once per K-step (round 3: libgptqhip_nohoist), eager us per call"
timeout 300 python tests/dev/tiled_ablate.py
GPTQHIP_LIB=$R/tests/dev/ablate/libgptqhip_nohoist.so timeout 300 python tests/dev/tiled_ablate.py
ABLATE_CASES=midm timeout 300 python tests/dev/tiled_ablate.py
GPTQHIP_LIB=$R/tests/dev/ablate/libgptqhip_nohoist.so ABLATE_CASES=midm timeout 300 python tests/dev/tiled_ablate.py
ABLATE_CASES=midm2 timeout 300 python tests/dev/tiled_ablate.py
GPTQHIP_LIB=$R/tests/dev/ablate/libgptqhip_nohoist.so ABLATE_CASES=midm2 timeout 300 python tests/dev/tiled_ablate.py
} > gpurun_out/r4c14_hoist.txt 2>&1
cat gpurun_out/r4c14_hoist.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stripe.py tests/test_gpu_fullsize_prefill.py -x -q 2>&1 | tail -5
