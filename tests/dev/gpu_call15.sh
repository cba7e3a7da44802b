#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O
( timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider --tb=short 2>&1 | tail -12 ) > $O/c15_pytest.log 2>&1
tail -8 $O/c15_pytest.log
( timeout 300 python tests/dev/w8_awq_bench.py 2>&1 | tail -12 )
