#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: collects the round's profile artefacts into gpurun_out/<tag>_*
# usage: bash tests/dev/collect_profiles.sh r01
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats -o bench -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/${TAG}_stats_bench.log 2>&1
timeout 180 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_pmc_fetch -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph > $O/${TAG}_pmc_fetch.log 2>&1
timeout 180 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_pmc_write -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph > $O/${TAG}_pmc_write.log 2>&1
timeout 180 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $O/${TAG}_pmc_sq -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph > $O/${TAG}_pmc_sq.log 2>&1
timeout 180 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/${TAG}_pmc_tiled -o p -- python $R/tests/dev/pmc_tiled.py > $O/${TAG}_pmc_tiled.log 2>&1
cd $R
timeout 240 python bench.py 2>&1 | tail -1 > $O/${TAG}_bench.json
timeout 240 python bench.py --model llama3-70b --steps 30 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > $O/${TAG}_bench_70b_tp1.json
timeout 240 python bench.py --dtype bf16 --steps 100 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 > $O/${TAG}_bench_bf16.json
timeout 240 python bench.py --dtype bf16 --exact-bf16 --steps 100 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 > $O/${TAG}_bench_bf16_exact_optin.json
timeout 240 python tests/dev/gemm_tflops.py 2>&1 | grep "^M=" > $O/${TAG}_gemm_tflops.txt
timeout 240 python tests/dev/gemm_tflops.py bf16 2>&1 | grep "^M=" > $O/${TAG}_gemm_tflops_bf16.txt
timeout 240 python tests/dev/midm.py 1,8,32,128,512,2048 2>&1 | grep "^K=" > $O/${TAG}_m_sweep.txt
timeout 240 python tests/dev/configs_bench.py 2>&1 | grep -E "^decode|^prefill" > $O/${TAG}_configs.txt
timeout 240 python tests/dev/torch_gpu_baseline.py 2>&1 | grep "^M=" > $O/${TAG}_torch_gpu_baseline.txt
timeout 240 python tests/dev/run_probe.py 2>&1 | grep "^K=" > $O/${TAG}_stream_probe.txt
echo done
