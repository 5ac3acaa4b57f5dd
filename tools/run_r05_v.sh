# round 5: lrf_encode8 built with -fno-slp-vectorize (variant noslp) against the shipped build on calls that use that unit's other
# kernels (launch-per-iteration k_bcd_w / k_bcd_w16, k_planes16, k_planes_strip), alternating on one box
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_v
mkdir -p $OUT
rm -f $OUT/ab.txt
cd $GRAFT_REPO_ROOT
for c in 7,3,3:64 7,3,3:24 16,8,8:64 7,3,3:128 7,3,3:256; do
  rk=${c%%:*}; n=${c##*:}
  for rep in 1 2; do
    for l in liblrf_hip.so liblrf_hip_noslp.so; do
      python tools/dev_lib_rank.py $l $rk $n >> $OUT/ab.txt 2>&1
    done
  done
done
cat $OUT/ab.txt
