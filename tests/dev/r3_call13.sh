#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for nt in 4 2; do echo "== GPTQHIP_WIDE_NT_MT2=$nt"; GPTQHIP_WIDE_NT_MT2=$nt MIDM_KERNELS=1 MIDM_SHAPES=4096x28672,4096x12288,8192x57344 timeout 600 python tests/dev/midm.py 17,24,32 2>&1 | grep "^K="; done | tee gpurun_out/r3_wide_nt_mt2.txt
