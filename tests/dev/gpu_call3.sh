#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
O="$GRAFT_REPO_ROOT/gpurun_out"
( timeout 900 python -m pytest tests/test_gpu_decode_chain.py tests/test_gpu_modules.py tests/test_abi.py -m gpu -q --timeout 600 -p no:cacheprovider --tb=short 2>&1 | tail -60 ) > $O/c3_pytest.log 2>&1
( timeout 600 python tests/dev/chain_ops_bench.py ) > $O/c3_chain_ops.txt 2>&1
( GPTQHIP_SKINNY_PIPE=1 timeout 600 python tests/dev/chain_ops_bench.py ) > $O/c3_chain_ops_pipe.txt 2>&1
( timeout 600 python bench.py --no-cpu-baseline --no-configs ) > $O/c3_bench.json 2> $O/c3_bench.err
( GPTQHIP_SKINNY_PIPE=1 timeout 600 python bench.py --no-cpu-baseline --no-configs ) > $O/c3_bench_pipe.json 2> $O/c3_bench_pipe.err
( timeout 600 python bench.py --no-cpu-baseline --no-configs --dtype bf16 ) > $O/c3_bench_bf16.json 2> $O/c3_bench_bf16.err
echo "=== pytest"; tail -30 $O/c3_pytest.log
echo "=== ops"; cat $O/c3_chain_ops.txt; echo "--- pipe"; cat $O/c3_chain_ops_pipe.txt
echo "=== bench"; cut -c1-400 $O/c3_bench.json; tail -3 $O/c3_bench.err; echo; cut -c1-400 $O/c3_bench_pipe.json; echo; cut -c1-400 $O/c3_bench_bf16.json
