#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_gpu_round3.py tests/test_gpu_checkpoint.py tests/test_gpu_e2e_llama.py -m gpu -q --timeout 900 -rf -p no:cacheprovider > gpurun_out/r3_pytest2.log 2>&1
echo "pytest rc=$?"; tail -15 gpurun_out/r3_pytest2.log
bash tests/dev/collect_profiles_r03.sh
cat gpurun_out/r03_bench.json | head -c 6000
echo; cat gpurun_out/r03_gemm_tflops_bf16.txt gpurun_out/r03_gemm_tflops_bf16_bf16scales.txt | head -40
cat gpurun_out/r03_e2e_llama8b*.txt
