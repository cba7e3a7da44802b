#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_gpu_round3.py tests/test_gpu_parity.py -m gpu -q --timeout 900 -rf -p no:cacheprovider -k "wide or gemm_vs_oracle or rows" > gpurun_out/r3_pytest8.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/r3_pytest8.log
echo "== decode op rows"; timeout 600 python tests/dev/rows_bench.py 5,8,12,16 2>&1 | grep "^M=" | tee gpurun_out/r3_rows_on2.txt
MIDM_KERNELS=1 MIDM_SHAPES=4096x28672,4096x8192 timeout 600 python tests/dev/midm.py 5,8,16 2>&1 | grep "^K=" | tee gpurun_out/r3_wide_on2.txt
