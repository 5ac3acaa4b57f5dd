# round 5: the persistent kernel's threshold by rank family: step times at 48..192 images with the default threshold (3584
# blocks = 150 images) and with LRF_PERSIST=1 (1024 blocks)
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_o
mkdir -p $OUT
rm -f $OUT/thr.txt
cd $GRAFT_REPO_ROOT
for r in 7,3,3 16,8,8 12,12,12 26,13,13 20,20,20; do
  for n in 48 64 96 128; do
    python tools/dev_lib_rank.py liblrf_hip.so $r $n >> $OUT/thr.txt 2>&1
    LRF_PERSIST=1 python tools/dev_lib_rank.py liblrf_hip.so $r $n | sed 's/^/PERSIST=1 /' >> $OUT/thr.txt 2>&1
  done
done
cat $OUT/thr.txt
