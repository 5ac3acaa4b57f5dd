# round 5: step time lines with and without the first iteration inside k_bcd_p (dev library, LRF_NO_PERSIST_FIRST)
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_l
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for r in 7,3,3 16,8,8; do
  for v in 1 0; do
    export LRF_NO_PERSIST_FIRST=$v
    rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -o run -- python3 $GRAFT_REPO_ROOT/tools/dev_lib_rank.py liblrf_hip_dev.so $r 256 > $OUT/tr.log 2>&1
    f=$(find $OUT/tr -name 'run_kernel_trace.csv' | head -1)
    for s in 6 9 12 15 18; do python3 $GRAFT_REPO_ROOT/tools/dev_trace_timeline.py $f $s | head -1; done > $OUT/steps_${r}_nofirst$v.txt
    python3 $GRAFT_REPO_ROOT/tools/dev_trace_timeline.py $f 10 > $OUT/timeline_${r}_nofirst$v.txt
    rm -rf $OUT/tr
    echo "ranks $r LRF_NO_PERSIST_FIRST=$v"; cat $OUT/steps_${r}_nofirst$v.txt
  done
done
tail -8 $OUT/timeline_7,3,3_nofirst0.txt
