#!/bin/bash
# per-kernel times of the any-shape initialisation (rocprofv3 --kernel-trace --stats), one shape per run: B M N R
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for shape in "256 1536 256 51" "256 512 768 102" "512 256 384 26"; do
  tag=$(echo $shape | tr ' ' '_')
  timeout -k 10 120 rocprofv3 --output-format csv --kernel-trace --stats -d $R/gpurun_out/anyprof_$tag -o p -- python3 $R/tools/dev_anyshape_stages.py child $shape > /dev/null 2>&1 < /dev/null
  f=$(find $R/gpurun_out/anyprof_$tag -name '*kernel_stats.csv' | head -1)
  echo "== $shape"
  if [ -n "$f" ]; then head -8 "$f" | cut -d, -f1-4 | cut -c1-150; fi
done
