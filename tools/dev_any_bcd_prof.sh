#!/bin/bash
# per-kernel times of the any-shape encode (rocprofv3 --kernel-trace --stats), one patch size per run
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for ps in 4 16 32 none; do
  timeout -k 10 200 rocprofv3 --output-format csv --kernel-trace --stats -d $R/gpurun_out/anybcd_$ps -o p -- python3 $R/tools/bench_anyshape.py 256 20 $ps > /dev/null 2>&1 < /dev/null
  echo "== $ps"
done
