#!/bin/bash
# round 4: the other bit widths on the GPU + the default bench line (live PMC on by default now)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_modules.py tests/test_gpu_checkpoint.py -q -m gpu -x 2>&1 | tail -15 > gpurun_out/r4a8_tests.txt
cat gpurun_out/r4a8_tests.txt
( time timeout 600 python bench.py ) > gpurun_out/r4a8_bench.json 2> gpurun_out/r4a8_bench.err
tail -c 3000 gpurun_out/r4a8_bench.json
tail -5 gpurun_out/r4a8_bench.err
