#!/bin/bash
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0
python tests/dev/ablate_bench.py - decode1 2>&1 | tail -1 > gpurun_out/call22.txt
GPTQHIP_NO_DECODE1=1 python tests/dev/ablate_bench.py - old 2>&1 | tail -1 >> gpurun_out/call22.txt
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider --tb=short -x 2>&1 | tail -15 >> gpurun_out/call22.txt
timeout 300 python bench.py --no-cpu-baseline --no-configs 2>&1 | cut -c1-300 | tail -1 >> gpurun_out/call22.txt
GPTQHIP_NO_DECODE1=1 timeout 300 python bench.py --no-cpu-baseline --no-configs 2>&1 | cut -c1-300 | tail -1 >> gpurun_out/call22.txt
timeout 300 python tests/dev/chain_ops_bench.py 2>&1 | tail -4 >> gpurun_out/call22.txt
