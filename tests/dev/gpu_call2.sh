#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
O="$GRAFT_REPO_ROOT/gpurun_out"
( timeout 900 python -m pytest tests/test_gpu_decode_chain.py tests/test_gpu_modules.py -m gpu -q --timeout 600 -p no:cacheprovider --tb=long 2>&1 | tail -150 ) > $O/c2_pytest.log 2>&1
( timeout 600 python tests/dev/chain_ops_bench.py ) > $O/c2_chain_ops.txt 2>&1
( timeout 300 python tests/dev/eager_overhead.py ) > $O/c2_eager.txt 2>&1
( timeout 600 python bench.py --no-cpu-baseline --no-configs --mode chain-serial ) > $O/c2_bench_serial.json 2> $O/c2_bench_serial.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c2_prof_serial -o serial -- python "$GRAFT_REPO_ROOT/bench.py" --mode chain-serial --no-configs --no-cpu-baseline --steps 50 --warmup 5 ) > $O/c2_prof_serial.log 2>&1
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c2_prof_modules -o modules -- python "$GRAFT_REPO_ROOT/bench.py" --mode modules --no-configs --no-cpu-baseline --steps 20 --warmup 3 ) > $O/c2_prof_modules.log 2>&1
find $O/c2_prof_serial $O/c2_prof_modules -type f -name '*kernel_trace*' -size +8M -delete 2>/dev/null
( timeout 900 python examples/hf_llama_dropin.py --size 8b --new-tokens 64 ) > $O/c2_e2e_8b.txt 2>&1
echo "=== pytest"; tail -60 $O/c2_pytest.log
echo "=== ops"; cat $O/c2_chain_ops.txt
echo "=== eager"; cat $O/c2_eager.txt
echo "=== e2e"; tail -5 $O/c2_e2e_8b.txt
