#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 900 -rf -p no:cacheprovider -k "gemm_vs_oracle or random_shape or kernels_agree or lm_head" > gpurun_out/r3_pytest5.log 2>&1
echo "pytest rc=$?"; tail -6 gpurun_out/r3_pytest5.log
echo "== wide kernel on"; timeout 600 python tests/dev/midm.py 5,8,12,16,24,32,48,64 2>&1 | grep "N=28672" | tee gpurun_out/r3_wide_on.txt
echo "== GPTQHIP_NO_WIDE=1"; GPTQHIP_NO_WIDE=1 MIDM_KERNELS=1 timeout 600 python tests/dev/midm.py 5,8,12,16,24,32,48,64 2>&1 | grep "N=28672" | tee gpurun_out/r3_wide_off.txt
