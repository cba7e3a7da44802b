#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O
( timeout 900 python tests/dev/tiled_plan_sweep.py 2>&1 | grep "K=" ) | tee $O/c16_plan.txt
