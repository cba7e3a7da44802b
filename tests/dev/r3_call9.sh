#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for mb in 256 192 128 96; do echo "== GPTQHIP_WIDE_MIN_BLOCKS=$mb"; GPTQHIP_WIDE_MIN_BLOCKS=$mb MIDM_KERNELS=1 MIDM_SHAPES=4096x6144,4096x4096,14336x4096,8192x8192 timeout 600 python tests/dev/midm.py 8,16,32 2>&1 | grep "^K="; done | tee gpurun_out/r3_wide_minblocks.txt
