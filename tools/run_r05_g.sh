# round 5: kernel time line of one (26,13,13) / (16,8,8) step (where do the forked initialisation kernels sit?)
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_g
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for r in 26,13,13 16,8,8; do
  rocprofv3 --kernel-trace --output-format csv -d $OUT/tr_$r -o run -- python3 $GRAFT_REPO_ROOT/tools/dev_lib_rank.py liblrf_hip.so $r 256 > $OUT/tr_$r.log 2>&1
  f=$(find $OUT/tr_$r -name 'run_kernel_trace.csv' | head -1)
  python3 $GRAFT_REPO_ROOT/tools/dev_trace_timeline.py $f 10 > $OUT/timeline_$r.txt
  rm -rf $OUT/tr_$r
  head -12 $OUT/timeline_$r.txt
done
