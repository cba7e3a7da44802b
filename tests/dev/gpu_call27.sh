#!/bin/bash
cd /root/repo
timeout 600 python bench.py --no-cpu-baseline 2>gpurun_out/call27.err > gpurun_out/call27.json
