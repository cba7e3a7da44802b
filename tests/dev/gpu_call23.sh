#!/bin/bash
cd /root/repo
python tests/dev/ablate_bench.py - decode1 2>&1 | tail -1 > gpurun_out/call23.txt
GPTQHIP_DECODE1_DEEP=8 python tests/dev/ablate_bench.py - deep8 2>&1 | tail -1 >> gpurun_out/call23.txt
GPTQHIP_DECODE1_DEEP=8 GPTQHIP_FORCE_WAVES=4 python tests/dev/ablate_bench.py - deep8w4 2>&1 | tail -1 >> gpurun_out/call23.txt
GPTQHIP_DECODE1_DEEP=8 GPTQHIP_FORCE_WAVES=7 python tests/dev/ablate_bench.py - deep8w7 2>&1 | tail -1 >> gpurun_out/call23.txt
GPTQHIP_FORCE_WAVES=8 python tests/dev/ablate_bench.py - w8 2>&1 | tail -1 >> gpurun_out/call23.txt
GPTQHIP_NO_DECODE1=1 python tests/dev/ablate_bench.py - old 2>&1 | tail -1 >> gpurun_out/call23.txt
