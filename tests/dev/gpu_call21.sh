#!/bin/bash
cd /root/repo
python tests/dev/ablate_bench.py - product > gpurun_out/call21.txt 2>&1
for a in 1 2 3 4 7 8 15; do
  python tests/dev/ablate_bench.py tests/dev/ablate/libgptqhip_abl$a.so abl$a 2>&1 | tail -1 >> gpurun_out/call21.txt
done
python tests/dev/ablate_bench.py - product 2>&1 | tail -1 >> gpurun_out/call21.txt
python tests/dev/run_probe.py 2>&1 | tail -6 >> gpurun_out/call21.txt
