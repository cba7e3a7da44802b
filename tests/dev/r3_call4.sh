#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 900 -rf -p no:cacheprovider -k "tiled or persistent or prefill or fullsize or kernels_agree" > gpurun_out/r3_pytest4.log 2>&1
echo "pytest rc=$?"; tail -6 gpurun_out/r3_pytest4.log
echo "== 64-row tiles, 5 stages (product)"; MIDM_KERNELS=2 timeout 600 python tests/dev/midm.py 17,32,64,96,128,256 2>&1 | grep "^K=" | tee gpurun_out/r3_tiled_d5.txt
echo "== 64-row tiles, 3 stages (round 2)"; GPTQHIP_LIB=$GRAFT_REPO_ROOT/tests/dev/ablate/libgptqhip_d3.so MIDM_KERNELS=2 timeout 600 python tests/dev/midm.py 17,32,64,96,128,256 2>&1 | grep "^K=" | tee gpurun_out/r3_tiled_d3.txt
