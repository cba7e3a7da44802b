#!/bin/bash
cd /root/repo
python tests/dev/irregular_bench.py pad 2>&1 | grep "K=" > gpurun_out/call30.txt
GPTQHIP_NO_PAD=1 python tests/dev/irregular_bench.py nopad 2>&1 | grep "K=" >> gpurun_out/call30.txt
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider --tb=short -x 2>&1 | tail -8 >> gpurun_out/call30.txt
timeout 300 python bench.py --no-cpu-baseline --no-configs 2>&1 | cut -c1-200 | tail -1 >> gpurun_out/call30.txt
