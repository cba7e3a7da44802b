#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py tests/test_gpu_modules.py -m gpu -q --timeout 900 -rf -p no:cacheprovider > gpurun_out/r3_pytest6.log 2>&1
echo "pytest rc=$?"; tail -6 gpurun_out/r3_pytest6.log
echo "== wide layers, product plan (skinny = decode kernel family, wide form where plannable)"
MIDM_SHAPES=4096x8192,8192x10240,8192x57344,4096x128256 timeout 900 python tests/dev/midm.py 8,16,24,32 2>&1 | grep "^K=" | tee gpurun_out/r3_wide2_on.txt
echo "== GPTQHIP_NO_WIDE=1"
GPTQHIP_NO_WIDE=1 MIDM_KERNELS=1 MIDM_SHAPES=4096x8192,8192x10240,8192x57344,4096x128256 timeout 900 python tests/dev/midm.py 8,16,24,32 2>&1 | grep "^K=" | tee gpurun_out/r3_wide2_off.txt
