#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
O="$GRAFT_REPO_ROOT/gpurun_out"; mkdir -p $O
( timeout 120 python __graft_entry__.py smoke ) > $O/c5_smoke.log 2>&1
( timeout 600 python -m pytest tests/test_gpu_decode_chain.py -m gpu -q --timeout 600 -p no:cacheprovider --tb=short 2>&1 | tail -15 ) > $O/c5_pytest.log 2>&1
( timeout 300 python tests/dev/chain_ops_bench.py 2>&1 | grep "K=" ) > $O/c5_ops.txt
( timeout 300 python tests/dev/chain_ops_bench.py bf16 2>&1 | grep "K=" ) > $O/c5_ops_bf16.txt
( timeout 300 python bench.py --no-cpu-baseline --no-configs ) > $O/c5_bench.json 2>/dev/null
echo "=== smoke"; tail -3 $O/c5_smoke.log; echo "=== pytest"; tail -5 $O/c5_pytest.log
echo "=== ops"; cat $O/c5_ops.txt; echo "--- bf16"; cat $O/c5_ops_bf16.txt
echo "=== bench"; cut -c1-260 $O/c5_bench.json
