#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O
( timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider --tb=short 2>&1 | tail -8 ) > $O/c11_pytest.log 2>&1
tail -6 $O/c11_pytest.log
( timeout 300 python bench.py --no-cpu-baseline --no-configs | cut -c1-140 )
