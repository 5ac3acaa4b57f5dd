# round 5: final check of the unit split (k_planes16_gram in lrf_planes_gram.hip without SLP, lrf_encode8.hip with it)
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_x
mkdir -p $OUT
rm -f $OUT/t.txt
cd $GRAFT_REPO_ROOT
for c in 7,3,3:256 20,10,10:64 26,13,13:64 16,8,8:64 7,3,3:24; do
  rk=${c%%:*}; n=${c##*:}
  python tools/dev_lib_rank.py liblrf_hip.so $rk $n >> $OUT/t.txt 2>&1
done
cat $OUT/t.txt
bash tools/run_r05_e.sh
