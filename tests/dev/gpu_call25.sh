#!/bin/bash
cd /root/repo
export GPTQHIP_DEV_THIN=1
: > gpurun_out/call25.txt
for v in 0 1 2 4 8; do
  GPTQHIP_FORCE_VARIANT=$v python tests/dev/ablate_bench.py - d1_w$v 2>&1 | tail -1 >> gpurun_out/call25.txt
done
for v in 1 2; do
  GPTQHIP_DECODE1_DEEP=8 GPTQHIP_FORCE_VARIANT=$v python tests/dev/ablate_bench.py - d1deep_w$v 2>&1 | tail -1 >> gpurun_out/call25.txt
done
