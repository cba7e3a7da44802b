#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
( cd tests/dev && timeout 900 python skinny_m_sweep.py 2>&1 | grep "K=" | cut -c1-40 ) | tee $O/c18_msweep.txt
( timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider --tb=short 2>&1 | tail -8 )
