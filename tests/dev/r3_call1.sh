#!/bin/bash
# round-3 GPU call 1: the whole GPU suite (no -x: every failure in one pass), the driver's multi-GPU command form on the one-GPU box,
# the mid-M sweep after the 512-thread-bound change
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -m gpu -q --timeout 900 -rf -p no:cacheprovider > gpurun_out/r3_pytest1.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/r3_pytest1.log
tail -40 gpurun_out/r3_pytest1.log
GPTQHIP_BENCH_SHARE_GPU=1 timeout 1200 python3 bench.py --gpus 2 --steps 50 --warmup 5 > gpurun_out/r3_bench_share2.out 2> gpurun_out/r3_bench_share2.err
echo "bench share rc=$?"
tail -c 3000 gpurun_out/r3_bench_share2.out
tail -5 gpurun_out/r3_bench_share2.err
timeout 600 python tests/dev/midm.py 9,12,16,24,32,48,64 > gpurun_out/r3_midm.txt 2>&1
cat gpurun_out/r3_midm.txt
