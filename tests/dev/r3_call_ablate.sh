#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
timeout 120 python tests/dev/tiled_ablate.py
for l in tests/dev/ablate/*.so; do GPTQHIP_LIB=$PWD/$l timeout 120 python tests/dev/tiled_ablate.py; done
timeout 120 python tests/dev/tiled_ablate.py
} > gpurun_out/tiled_ablate.txt 2>&1
cat gpurun_out/tiled_ablate.txt | grep -v amdgpu.ids
