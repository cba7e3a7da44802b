#!/bin/bash
cd "$GRAFT_REPO_ROOT/tests/dev"; O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
( timeout 900 python skinny_m_sweep.py 2>&1 | grep "K=" ) | tee $O/c17_msweep.txt
