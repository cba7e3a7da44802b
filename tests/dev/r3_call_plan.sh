#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "tiled or split or prefill or random_shape or fullsize or lm_head or persistent or gemm_vs_oracle" -x > gpurun_out/pytest_tiled.txt 2>&1
tail -3 gpurun_out/pytest_tiled.txt
timeout 600 python tests/dev/tiled_plan_check.py 96,128,160,192,256,320,384,448,512,640,768,896,1024,1280,1536,2048 > gpurun_out/tiled_plan_check2.txt 2>&1
grep -v amdgpu.ids gpurun_out/tiled_plan_check2.txt | grep -c "planner loses"
grep -v amdgpu.ids gpurun_out/tiled_plan_check2.txt | grep "planner loses"
