#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
bash tests/dev/collect_profiles_r04.sh > gpurun_out/r04_collect.log 2>&1
tail -3 gpurun_out/r04_collect.log
( time timeout 900 python -m pytest "tests/test_gpu_round3.py::test_tp_chain_two_layers_vs_oracle_composition" -q 2>&1 | tail -2 ) 2>&1 | tail -6
ls gpurun_out | head -40
