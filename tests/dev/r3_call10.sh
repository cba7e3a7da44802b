#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for mm in 5 2; do echo "== GPTQHIP_WIDE_MIN_M=$mm"; GPTQHIP_WIDE_MIN_M=$mm MIDM_KERNELS=1 MIDM_SHAPES=4096x28672,4096x6144 timeout 600 python tests/dev/midm.py 2,3,4 2>&1 | grep "^K="; GPTQHIP_WIDE_MIN_M=$mm timeout 600 python tests/dev/rows_bench.py 2,4 2>&1 | grep "^M="; done | tee gpurun_out/r3_wide_minm.txt
