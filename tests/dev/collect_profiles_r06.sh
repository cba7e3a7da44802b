#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: round-6 profile artefacts into gpurun_out/r06_*
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# (1) the bench number and the per-kernel table from the SAME run
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r06_stats -o bench -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-configs --no-e2e --no-live-pmc > $O/r06_stats_bench.log 2>&1
# (2) counters: one group per pass (tests/dev/pmc_passes.sh): SQ / GRBM / FETCH_SIZE / WRITE_SIZE for the decode ops and the prefill kernel
bash $R/tests/dev/pmc_passes.sh r06 > $O/r06_pmc_passes.log 2>&1
find $O/r06_stats -type f -size +12M -delete 2>/dev/null
cd $R
# (3) plain runs: the default command (what the driver records), then the variants
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06_bench_line.json 2> $O/r06_bench.err
cp $O/bench_detail.json $O/r06_bench_detail.json
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-configs --no-e2e > $O/r06_bench_line_200steps.json 2>/dev/null
timeout 300 python bench.py --dtype bf16 --no-cpu-baseline --no-configs --no-e2e > $O/r06_bench_bf16.json 2>/dev/null
GPTQHIP_DECODE_BITFAITHFUL=1 timeout 300 python bench.py --no-cpu-baseline --no-configs --no-e2e > $O/r06_bench_bitfaithful.json 2>/dev/null
echo done
