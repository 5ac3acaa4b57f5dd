# round 5: the first iteration inside k_bcd_p (k_bcd_p<.., 0, true>): parity, then A/B against the same build with
# LRF_NO_PERSIST_FIRST=1 (dev library) on one box
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_k
mkdir -p $OUT
rm -f $OUT/ab.txt
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_configs_at_size.py tests/test_hip_parity.py tests/test_persist_error.py -x -q -m gpu > $OUT/t.log 2>&1 || { tail -30 $OUT/t.log; exit 1; }
tail -2 $OUT/t.log
for r in 7,3,3 16,8,8 4,2,2 12,12,12; do
  for rep in 1 2; do
    LRF_NO_PERSIST_FIRST=1 python tools/dev_lib_rank.py liblrf_hip_dev.so $r 256 >> $OUT/ab.txt 2>&1
    python tools/dev_lib_rank.py liblrf_hip_dev.so $r 256 >> $OUT/ab.txt 2>&1
  done
done
cat $OUT/ab.txt
