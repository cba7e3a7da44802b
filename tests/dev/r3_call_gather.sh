#!/bin/bash
# dev: A/B of the gather kernels (previous build vs this tree), the new decode-op / e2e tests, C3 bench entries
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
{
  echo "=== gather bench: previous build"; GPTQHIP_LIB=$PWD/tests/dev/ablate/libgptqhip_prev.so timeout 300 python tests/dev/gather_bench.py
  echo "=== gather bench: this tree"; timeout 300 python tests/dev/gather_bench.py
} > gpurun_out/gather_bench.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_decode_chain.py tests/test_gpu_e2e_llama.py tests/test_gpu_round3.py tests/test_gpu_checkpoint.py -x -q -m gpu > gpurun_out/pytest_sel.txt 2>&1
echo "pytest rc $?" >> gpurun_out/pytest_sel.txt
timeout 600 python bench.py > gpurun_out/bench_after_gather.json 2> gpurun_out/bench_after_gather.err
tail -5 gpurun_out/pytest_sel.txt
cat gpurun_out/gather_bench.txt
