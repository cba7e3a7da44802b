#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: round-2 profile artefacts into gpurun_out/r02_*
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# (1) the bench number and the per-kernel table from the SAME run
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r02_stats -o bench -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-configs > $O/r02_stats_bench.log 2>&1
# (2) HBM traffic: separate --pmc passes (no trace domains besides --kernel-trace), eager launches so every dispatch is counted
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/r02_pmc_fetch -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-graph > $O/r02_pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/r02_pmc_write -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-graph > $O/r02_pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $O/r02_pmc_sq -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-graph > $O/r02_pmc_sq.log 2>&1
for d in r02_pmc_fetch r02_pmc_write r02_pmc_sq; do python $R/tests/dev/pmc_agg.py $O/$d > /dev/null 2>&1; done
find $O/r02_stats $O/r02_pmc_fetch $O/r02_pmc_write $O/r02_pmc_sq -type f -size +12M -delete 2>/dev/null
cd $R
# (3) plain runs
timeout 900 python bench.py > $O/r02_bench.json 2> $O/r02_bench.err
timeout 300 python bench.py --dtype bf16 --no-cpu-baseline --no-configs > $O/r02_bench_bf16.json 2>/dev/null
timeout 300 python bench.py --mode modules --no-cpu-baseline --no-configs > $O/r02_bench_modules.json 2>/dev/null
timeout 300 python tests/dev/chain_ops_bench.py 2>&1 | grep "K=" > $O/r02_decode_ops.txt
timeout 300 python tests/dev/eager_overhead.py 2>&1 | grep "K=" > $O/r02_eager_overhead.txt
timeout 900 python examples/hf_llama_dropin.py --size 8b --new-tokens 64 2>&1 | grep -v amdgpu.ids > $O/r02_e2e_llama8b.txt
timeout 300 python tests/dev/configs_bench.py 2>&1 | grep -E "^decode|^prefill" > $O/r02_configs.txt
timeout 300 python tests/dev/torch_gpu_baseline.py 2>&1 | grep "^M=" > $O/r02_torch_gpu_baseline.txt
echo done
