#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_e2e_llama.py -q --timeout 600 -p no:cacheprovider --tb=short -x 2>&1 | tail -25 > gpurun_out/call36.txt
