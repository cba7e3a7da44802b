#!/bin/bash
# Run ON THE GPU BOX: per-kernel durations (main + split-K reduce) of mid-M shapes.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for cfg in "16 4096 28672" "64 4096 28672" "16 4096 4096"; do
  tag=$(echo $cfg | tr ' ' '_')
  timeout 90 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/midm_$tag -o p -- python $R/tests/dev/midm_prof.py $cfg > /dev/null 2>&1
  f=$(find $R/gpurun_out/midm_$tag -name "*kernel_stats.csv" | head -1)
  echo "== $cfg"; [ -n "$f" ] && grep -E "tiled_kernel|splitk_reduce|skinny" $f | awk -F'",' '{n=$1; gsub(/"/,"",n); print substr(n,1,70), $2}'
done
