#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for cfg in "2048 4096 4096" "8192 4096 4096" "2048 14336 4096"; do
  tag=$(echo $cfg | tr ' ' '_')
  timeout 90 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/gath_$tag -o p -- python $R/tests/dev/gather_prof.py $cfg > /dev/null 2>&1
  f=$(find $R/gpurun_out/gath_$tag -name "*kernel_stats.csv" | head -1)
  echo "== $cfg"; [ -n "$f" ] && grep -E "tiled_kernel|gather" $f | awk -F'",' '{n=$1; gsub(/"/,"",n); print substr(n,1,70), $2}'
done
