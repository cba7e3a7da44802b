#!/bin/bash
# same-box A/B: tests/dev/ablate/*.so vs the tree's library, alternating, twice
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
for rep in 1 2; do
  for l in tests/dev/ablate/*.so; do GPTQHIP_LIB=$PWD/$l timeout 120 python tests/dev/tiled_ablate.py; done
  timeout 120 python tests/dev/tiled_ablate.py
done
} > gpurun_out/tiled_ab.txt 2>&1
grep -v amdgpu.ids gpurun_out/tiled_ab.txt
