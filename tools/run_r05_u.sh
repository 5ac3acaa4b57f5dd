# round 5: the persistent kernel with and without the compiler's packed fp32 math (variant noslp2: lrf_bcd_persist built with
# -fno-slp-vectorize), alternating on one box
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_u
mkdir -p $OUT
rm -f $OUT/ab.txt
cd $GRAFT_REPO_ROOT
for r in 7,3,3 16,8,8 26,13,13 12,12,12; do
  for rep in 1 2; do
    for l in liblrf_hip.so liblrf_hip_noslp2.so; do
      python tools/dev_lib_persist.py $l $r 256 >> $OUT/ab.txt 2>&1
    done
  done
done
cat $OUT/ab.txt
