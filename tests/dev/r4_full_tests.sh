#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/ -q -m gpu --maxfail=8 --durations=14 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo\|Terminating process" | tail -45 > gpurun_out/r4_full_pytest.txt
cat gpurun_out/r4_full_pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee gpurun_out/r4_smoke.txt
