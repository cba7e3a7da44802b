#!/bin/bash
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider --tb=short -x 2>&1 | tail -25 > gpurun_out/call26.txt
timeout 300 python bench.py --no-cpu-baseline --no-configs 2>&1 | cut -c1-200 | tail -1 >> gpurun_out/call26.txt
