#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_parity.py -q --timeout 600 -p no:cacheprovider --tb=short -x -k "stress" 2>&1 | tail -15 > gpurun_out/call37.txt
timeout 600 python bench.py --no-cpu-baseline --no-e2e 2>gpurun_out/call37.err > gpurun_out/call37.json
