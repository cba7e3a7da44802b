#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_gpu_round3.py tests/test_gpu_decode_chain.py tests/test_gpu_e2e_llama.py -m gpu -q --timeout 900 -rf -p no:cacheprovider > gpurun_out/r3_pytest7.log 2>&1
echo "pytest rc=$?"; tail -6 gpurun_out/r3_pytest7.log
echo "== decode op rows (wide form on)"; timeout 600 python tests/dev/rows_bench.py 1,4,5,8,12,16 2>&1 | grep "^M=" | tee gpurun_out/r3_rows_on.txt
echo "== GPTQHIP_NO_WIDE=1"; GPTQHIP_NO_WIDE=1 timeout 600 python tests/dev/rows_bench.py 5,8,12,16 2>&1 | grep "^M=" | tee gpurun_out/r3_rows_off.txt
for b in 8 16; do timeout 600 python examples/hf_llama_dropin.py --size 8b --new-tokens 32 --batch $b 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_e2e_batch$b.txt; done
