#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O="$GRAFT_REPO_ROOT/gpurun_out"
( time timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider --tb=short 2>&1 | tail -40 ) > $O/c4_pytest.log 2>&1
( timeout 120 python __graft_entry__.py smoke ) > $O/c4_smoke.log 2>&1
bash tests/dev/collect_profiles_r02.sh > $O/c4_collect.log 2>&1
echo "=== pytest"; tail -15 $O/c4_pytest.log; echo "=== smoke"; tail -3 $O/c4_smoke.log
echo "=== ops"; cat $O/r02_decode_ops.txt; cat $O/r02_eager_overhead.txt; cat $O/r02_e2e_llama8b.txt | tail -4
echo "=== bench"; cut -c1-300 $O/r02_bench.json; tail -3 $O/r02_bench.err; grep -o '"value": [0-9.]*' $O/r02_stats_bench.log | head -2
