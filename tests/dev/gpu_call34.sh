#!/bin/bash
cd /root/repo
python tests/dev/mem_probe.py 2>&1 | grep -v amdgpu > gpurun_out/call34.txt
timeout 900 python -m pytest tests/test_gpu_e2e_llama.py -q --timeout 600 -p no:cacheprovider --tb=short -x 2>&1 | tail -8 >> gpurun_out/call34.txt
