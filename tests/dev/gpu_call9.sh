#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O
echo "--- product build (nt weight loads)" > $O/c9_l2.txt
( timeout 300 python tests/dev/l2_survival.py 2>&1 | grep "|" ) >> $O/c9_l2.txt
echo "--- plain-load build" >> $O/c9_l2.txt
( GPTQHIP_LIB=$GRAFT_REPO_ROOT/tests/dev/libgptqhip_plainloads.so timeout 300 python tests/dev/l2_survival.py 2>&1 | grep "|" ) >> $O/c9_l2.txt
cat $O/c9_l2.txt
