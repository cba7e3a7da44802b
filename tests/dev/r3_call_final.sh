#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -q -m gpu > gpurun_out/pytest_final3.txt 2>&1; echo "rc $?" >> gpurun_out/pytest_final3.txt
tail -4 gpurun_out/pytest_final3.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tests/dev/collect_profiles_r03.sh > gpurun_out/r3_collect.log 2>&1
tail -3 gpurun_out/r3_collect.log
timeout 300 python tests/dev/midm.py > gpurun_out/r3_midm_final.txt 2>&1
tail -5 gpurun_out/r3_midm_final.txt
