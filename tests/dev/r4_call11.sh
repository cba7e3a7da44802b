#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_tp8_shapes.py -x -q -k short_k 2>&1 | tail -15 > gpurun_out/r4c11_pytest_a.txt
cat gpurun_out/r4c11_pytest_a.txt
timeout 1500 python -m pytest "tests/test_gpu_comm.py::test_tp_decode_chain_processes_sharing_one_gpu" -x -q 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | tail -60 > gpurun_out/r4c11_pytest_b.txt
cat gpurun_out/r4c11_pytest_b.txt
timeout 1500 python -m pytest tests/test_gpu_round3.py -x -q -k "tp_chain" 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | tail -60 > gpurun_out/r4c11_pytest_c.txt
cat gpurun_out/r4c11_pytest_c.txt
