#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "tiled or split or prefill or random_shape or fullsize or lm_head or persistent" -x > gpurun_out/pytest_tiled.txt 2>&1
tail -3 gpurun_out/pytest_tiled.txt
rm -rf /tmp/midm_trace
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/midm_trace -o t -- python tests/dev/tiled_midm_trace.py > gpurun_out/midm_trace.log 2>&1
python tests/dev/tiled_midm_trace.py summarize /tmp/midm_trace gpurun_out/midm_trace.log > gpurun_out/midm_trace_summary2.txt 2>&1
grep "v=0" gpurun_out/midm_trace_summary2.txt
