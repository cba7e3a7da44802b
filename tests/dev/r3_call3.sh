#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py -m gpu -q --timeout 900 -rf -p no:cacheprovider -k "gemm_vs_oracle or rows_33 or kernels_agree or waves_per_block or act_order_and_w8 or random_shape" > gpurun_out/r3_pytest3.log 2>&1
echo "pytest rc=$?"; tail -8 gpurun_out/r3_pytest3.log
echo "== default (17..32 rows: up to 16 waves)"; timeout 600 python tests/dev/midm.py 17,24,32,40,48,56,64 2>&1 | grep "^K=" | tee gpurun_out/r3_midm_w16.txt
echo "== GPTQHIP_MT2_WAVES=8"; GPTQHIP_MT2_WAVES=8 timeout 600 python tests/dev/midm.py 17,24,32 2>&1 | grep "^K=" | tee gpurun_out/r3_midm_w8.txt
