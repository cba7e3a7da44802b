#!/bin/bash
# round-2 GPU call 1: full GPU test suite, bench (all configs), per-kernel profile of the serial chain
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | tail -40 ) > gpurun_out/c1_pytest.log 2>&1
( time timeout 900 python bench.py ) > gpurun_out/c1_bench.json 2> gpurun_out/c1_bench.err
tail -c 3000 gpurun_out/c1_bench.err > gpurun_out/c1_bench.err.tail
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/c1_prof_serial" -o serial -- python "$GRAFT_REPO_ROOT/bench.py" --mode chain-serial --no-configs --no-cpu-baseline --steps 50 --warmup 5 ) > gpurun_out/c1_prof_serial.log 2>&1
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/c1_prof_chain" -o chain -- python "$GRAFT_REPO_ROOT/bench.py" --mode chain --no-configs --no-cpu-baseline --steps 50 --warmup 5 ) > gpurun_out/c1_prof_chain.log 2>&1
# keep only the stats csvs (traces are big)
find gpurun_out/c1_prof_serial gpurun_out/c1_prof_chain -type f ! -name '*stats*' -size +2M -delete 2>/dev/null
echo "=== pytest"; cat gpurun_out/c1_pytest.log
echo "=== bench"; cat gpurun_out/c1_bench.json | cut -c1-6000; cat gpurun_out/c1_bench.err.tail | tail -20
