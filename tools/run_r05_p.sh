# round 5: k_planes16_gram (patch matrices + the luma planes' exact Gram partials in one kernel) against the two-kernel form
# (dev build, LRF_NO_FUSED_GRAM=1): same factors, step times, alternating on one box
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_p
mkdir -p $OUT
rm -f $OUT/ab.txt
cd $GRAFT_REPO_ROOT
for r in 7,3,3:256 16,8,8:256 26,13,13:256 7,3,3:24 7,3,3:3; do
  rk=${r%%:*}; n=${r##*:}
  for rep in 1 2; do
    LRF_NO_FUSED_GRAM=1 timeout -k 10 200 python tools/dev_lib_rank.py liblrf_hip_dev.so $rk $n | sed 's/^/two kernels: /' >> $OUT/ab.txt 2>&1
    timeout -k 10 200 python tools/dev_lib_rank.py liblrf_hip_dev.so $rk $n | sed 's/^/fused:       /' >> $OUT/ab.txt 2>&1
  done
done
cat $OUT/ab.txt
