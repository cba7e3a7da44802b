#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_comm.py -m gpu -q --timeout 300 -p no:cacheprovider --tb=long 2>&1 | tail -60 ) > $O/c12_pytest.log 2>&1
tail -50 $O/c12_pytest.log
