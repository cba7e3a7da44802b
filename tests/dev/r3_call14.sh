#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_gpu_round3.py tests/test_gpu_decode_chain.py tests/test_gpu_e2e_llama.py -m gpu -q --timeout 900 -rf -p no:cacheprovider -k "wide or rows or decoder_layers" > gpurun_out/r3_pytest14.log 2>&1
echo "pytest rc=$?"; tail -6 gpurun_out/r3_pytest14.log
timeout 600 python tests/dev/rows_bench.py 1,2,3,4,5,8,16 2>&1 | grep "^M=" | tee gpurun_out/r3_rows_final.txt
