#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O
( timeout 600 python tests/dev/tiled_variants_bench.py 2>&1 | grep "bits=" ) | tee $O/c14_tiled.txt
