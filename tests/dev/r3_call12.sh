#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_gpu_round3.py tests/test_gpu_e2e_llama.py -m gpu -q --timeout 900 -rf -p no:cacheprovider -k "wide_form or sliding_window" > gpurun_out/r3_pytest12.log 2>&1
echo "pytest rc=$?"; tail -8 gpurun_out/r3_pytest12.log
timeout 600 python bench.py --model llama3-70b --steps 20 --warmup 3 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-900
