#!/bin/bash
cd /root/repo
python tests/dev/trace_blocks.py tests/dev/ablate/libgptqhip_abl16.so > gpurun_out/call24.txt 2>&1
