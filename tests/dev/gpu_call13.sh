#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O
( timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider --tb=short 2>&1 | tail -12 ) > $O/c13_pytest.log 2>&1
tail -8 $O/c13_pytest.log
( timeout 300 python bench.py --no-cpu-baseline --no-configs --dtype bf16 | cut -c1-140 )
( timeout 300 python bench.py --no-cpu-baseline --no-configs | cut -c1-140 )
( timeout 300 python tests/dev/chain_ops_bench.py bf16 2>&1 | grep "K=" )
( timeout 300 python tests/dev/gemm_tflops.py bf16 2>&1 | grep "^M=" | head -8 )
( timeout 600 python bench.py --model llama3-70b --steps 10 --warmup 2 --no-cpu-baseline | cut -c1-200 )
