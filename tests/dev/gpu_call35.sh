#!/bin/bash
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0
python tests/dev/mem_probe.py 2>&1 | grep -v amdgpu | grep -v buffers > gpurun_out/call35.txt
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider --tb=short -x 2>&1 | tail -8 >> gpurun_out/call35.txt
timeout 600 python examples/hf_llama_dropin.py --size 8b --new-tokens 64 2>&1 | grep -v amdgpu.ids | tail -4 >> gpurun_out/call35.txt
timeout 600 python examples/hf_llama_dropin.py --size 8b --new-tokens 64 --dtype bf16 2>&1 | grep -v amdgpu.ids | tail -2 >> gpurun_out/call35.txt
timeout 600 python examples/hf_llama_dropin.py --size 8b --new-tokens 64 --quant-lm-head 2>&1 | grep -v amdgpu.ids | tail -2 >> gpurun_out/call35.txt
