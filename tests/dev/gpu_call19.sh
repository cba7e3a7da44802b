#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
( timeout 900 python tests/dev/midm.py 6,8,12,16,24,32,40,48,64,96,128 2>&1 | grep "K=" ) | tee $O/c19_midm.txt
