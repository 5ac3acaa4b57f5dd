# round 5: k_planes16_gram always on: the whole GPU suite, then step times at 48..128 images against the two-kernel form (dev build)
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_r
mkdir -p $OUT
rm -f $OUT/ab.txt
cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu > $OUT/t_all.log 2>&1 || { tail -40 $OUT/t_all.log; exit 1; }
tail -3 $OUT/t_all.log
for n in 48 64 96 128; do
  for r in 7,3,3 16,8,8; do
    LRF_NO_FUSED_GRAM=1 python tools/dev_lib_rank.py liblrf_hip_dev.so $r $n | sed 's/^/two kernels: /' >> $OUT/ab.txt 2>&1
    python tools/dev_lib_rank.py liblrf_hip_dev.so $r $n | sed 's/^/fused:       /' >> $OUT/ab.txt 2>&1
  done
done
cat $OUT/ab.txt
