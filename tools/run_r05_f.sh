# round 5: the order in which the initialisation kernels of a call's rank families are enqueued (LRF_INIT_ORDER, dev build)
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_f
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for r in 16,8,8 17,8,8 20,10,10 26,13,13 32,16,16; do
  for o in 1 2 0; do
    echo "order $o" >> $OUT/order.txt
    LRF_INIT_ORDER=$o python tools/dev_lib_rank.py liblrf_hip_dev.so $r 256 >> $OUT/order.txt 2>&1
  done
done
cat $OUT/order.txt
