#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O
echo "--- depth2 allowed" > $O/c10.txt
( timeout 300 python tests/dev/chain_ops_bench.py 2>&1 | grep "K=" ) >> $O/c10.txt
( timeout 300 python bench.py --no-cpu-baseline --no-configs | cut -c1-140 ) >> $O/c10.txt
echo "--- GPTQHIP_NO_DEPTH2=1" >> $O/c10.txt
( GPTQHIP_NO_DEPTH2=1 timeout 300 python tests/dev/chain_ops_bench.py 2>&1 | grep "K=" ) >> $O/c10.txt
( GPTQHIP_NO_DEPTH2=1 timeout 300 python bench.py --no-cpu-baseline --no-configs | cut -c1-140 ) >> $O/c10.txt
cat $O/c10.txt
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_decode_chain.py tests/test_gpu_modules.py -m gpu -q --timeout 600 -p no:cacheprovider --tb=short -x 2>&1 | tail -5 )
