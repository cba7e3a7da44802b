#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_tp8_shapes.py tests/test_gpu_fullsize_prefill.py tests/test_gpu_checkpoint.py tests/test_gpu_comm.py -x -q 2>&1 | tail -15 > gpurun_out/r4c10_pytest_a.txt
cat gpurun_out/r4c10_pytest_a.txt
timeout 1500 python -m pytest tests/test_gpu_round3.py -x -q -k "stress or lost_peer or tp_chain" 2>&1 | tail -15 > gpurun_out/r4c10_pytest_b.txt
cat gpurun_out/r4c10_pytest_b.txt
GPTQHIP_BENCH_SHARE_GPU=1 timeout 900 python3 bench.py --gpus 8 --steps 5 --warmup 2 > gpurun_out/r4c10_bench_share8.json 2> gpurun_out/r4c10_bench_share8.err
tail -c 3000 gpurun_out/r4c10_bench_share8.json; tail -5 gpurun_out/r4c10_bench_share8.err
