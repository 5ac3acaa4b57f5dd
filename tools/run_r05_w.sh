# round 5: 64-image calls at ranks 17..32 with lrf_encode8 built without (shipped) and with (variant slp) SLP vectorisation
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_w
mkdir -p $OUT
rm -f $OUT/ab.txt
cd $GRAFT_REPO_ROOT
for c in 20,10,10:64 26,13,13:64 13,13,13:64 10,10,10:64 16,8,8:64; do
  rk=${c%%:*}; n=${c##*:}
  for rep in 1 2; do
    for l in liblrf_hip.so liblrf_hip_slp.so; do
      python tools/dev_lib_rank.py $l $rk $n >> $OUT/ab.txt 2>&1
    done
  done
done
cat $OUT/ab.txt
