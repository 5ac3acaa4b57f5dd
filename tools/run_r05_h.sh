# round 5: k_init with the packed LDS layout: parity of the initialisation, step times by rank triple, one time line
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_h
mkdir -p $OUT
rm -f $OUT/rank.txt
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_parity.py tests/test_configs_at_size.py -x -q -m gpu > $OUT/t.log 2>&1 || { tail -30 $OUT/t.log; exit 1; }
tail -2 $OUT/t.log
for r in 7,3,3 10,5,5 16,8,8 20,10,10 26,13,13 12,12,12; do
  python tools/dev_lib_rank.py liblrf_hip.so $r 256 >> $OUT/rank.txt 2>&1
done
cat $OUT/rank.txt
cd /tmp && export TMPDIR=/tmp
for r in 26,13,13 16,8,8; do
  rocprofv3 --kernel-trace --output-format csv -d $OUT/tr_$r -o run -- python3 $GRAFT_REPO_ROOT/tools/dev_lib_rank.py liblrf_hip.so $r 256 > $OUT/tr_$r.log 2>&1
  f=$(find $OUT/tr_$r -name 'run_kernel_trace.csv' | head -1)
  python3 $GRAFT_REPO_ROOT/tools/dev_trace_timeline.py $f 10 > $OUT/timeline_$r.txt
  rm -rf $OUT/tr_$r
  head -7 $OUT/timeline_$r.txt
done
