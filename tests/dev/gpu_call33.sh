#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_e2e_llama.py -q --timeout 600 -p no:cacheprovider --tb=short -x 2>&1 | tail -30 > gpurun_out/call33.txt
timeout 600 python examples/hf_llama_dropin.py --size 8b --new-tokens 64 2>&1 | grep -v amdgpu.ids | tail -6 >> gpurun_out/call33.txt
timeout 600 python examples/hf_llama_dropin.py --size 8b --new-tokens 64 --siblings-only 2>&1 | grep -v amdgpu.ids | tail -4 >> gpurun_out/call33.txt
