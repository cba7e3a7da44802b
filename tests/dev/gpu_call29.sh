#!/bin/bash
R=/root/repo; O=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $O/r02_pmc_sq -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-graph > $O/r02_pmc_sq.log 2>&1
python $R/tests/dev/pmc_agg.py $O/r02_pmc_sq > $O/r02_pmc_sq_agg.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $O/r02_pmc_sq2 -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-graph > $O/r02_pmc_sq2.log 2>&1
python $R/tests/dev/pmc_agg.py $O/r02_pmc_sq2 > $O/r02_pmc_sq2_agg.log 2>&1
find $O/r02_pmc_sq $O/r02_pmc_sq2 -type f -size +12M -delete
