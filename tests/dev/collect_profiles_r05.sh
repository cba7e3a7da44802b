#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: round-5 profile artefacts into gpurun_out/r05_*
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# (1) the bench number and the per-kernel table from the SAME run
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r05_stats -o bench -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-configs --no-live-pmc > $O/r05_stats_bench.log 2>&1
# (2) HBM traffic: separate --pmc passes (no trace domains besides --kernel-trace), eager launches so every dispatch is counted
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/r05_pmc_fetch -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-graph --no-live-pmc > $O/r05_pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/r05_pmc_write -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-graph --no-live-pmc > $O/r05_pmc_write.log 2>&1
for d in r05_pmc_fetch r05_pmc_write; do python $R/tests/dev/pmc_agg.py $O/$d > /dev/null 2>&1; done
find $O/r05_stats $O/r05_pmc_fetch $O/r05_pmc_write -type f -size +12M -delete 2>/dev/null
cd $R
# (3) plain runs: the default command (what the driver records), then the variants
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05_bench_line.json 2> $O/r05_bench.err
cp $O/bench_detail.json $O/r05_bench_detail.json
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-configs --no-e2e > $O/r05_bench_line_200steps.json 2>/dev/null
timeout 300 python bench.py --dtype bf16 --no-cpu-baseline --no-configs > $O/r05_bench_bf16.json 2>/dev/null
timeout 300 python tests/dev/gemm_tflops.py bf16 2>&1 | grep "^M=" > $O/r05_gemm_tflops_bf16.txt
timeout 300 python tests/dev/gemm_tflops.py bf16 bf16s 2>&1 | grep "^M=" > $O/r05_gemm_tflops_bf16_bf16scales.txt
timeout 300 python tests/dev/gemm_tflops.py 2>&1 | grep "^M=" > $O/r05_gemm_tflops.txt
timeout 900 python examples/hf_llama_dropin.py --size 8b --new-tokens 64 2>&1 | grep -v amdgpu.ids > $O/r05_e2e_llama8b.txt
MIDM_KERNELS=1,2,0 MIDM_SHAPES=4096x4096,4096x11008,11008x4096,4096x6144,4096x28672,14336x4096 timeout 300 python tests/dev/midm.py 64,72,96,128,136,192,256 2>&1 | grep "^K=" > $O/r05_mid_m_sweep.txt
echo done
