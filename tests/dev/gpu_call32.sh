#!/bin/bash
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0
python tests/dev/g64_bench.py 2>&1 | grep "g=" > gpurun_out/call32.txt
echo "--- previous build (generic path for g32/g64)" >> gpurun_out/call32.txt
python - >> gpurun_out/call32.txt 2>&1 <<'PY'
import sys, runpy
sys.path.insert(0, "/root/repo")
import gptqmodel_amd._lib as L
L.LIB_PATH = "/root/repo/tests/dev/ablate/libgptqhip_prev.so"
sys.argv = ["g64_bench.py"]
runpy.run_path("/root/repo/tests/dev/g64_bench.py", run_name="__main__")
PY
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider --tb=short -x 2>&1 | tail -8 >> gpurun_out/call32.txt
