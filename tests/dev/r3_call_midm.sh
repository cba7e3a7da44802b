#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf /tmp/midm_trace
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/midm_trace -o t -- python tests/dev/tiled_midm_trace.py > gpurun_out/midm_trace.log 2>&1
python tests/dev/tiled_midm_trace.py summarize /tmp/midm_trace gpurun_out/midm_trace.log > gpurun_out/midm_trace_summary.txt 2>&1
tail -5 gpurun_out/midm_trace.log
wc -l gpurun_out/midm_trace_summary.txt
