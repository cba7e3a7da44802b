#!/bin/bash
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider --tb=short -x 2>&1 | tail -15 > gpurun_out/call20_tests.txt
timeout 300 python bench.py --no-cpu-baseline --no-configs 2>&1 | cut -c1-400 > gpurun_out/call20_bench.txt
timeout 600 python bench.py --model llama3-70b --steps 20 --warmup 3 2>&1 | cut -c1-400 > gpurun_out/call20_70b.txt
