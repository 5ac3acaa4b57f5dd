# round 5: k_init<ZR, 8, DENSE>: the no-spill form for calls with no more matrices than CUs
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_y
mkdir -p $OUT
rm -f $OUT/t.txt
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_parity.py tests/test_configs_at_size.py tests/test_qmf_class.py -x -q -m gpu > $OUT/t.log 2>&1 || { tail -30 $OUT/t.log; exit 1; }
tail -2 $OUT/t.log
for c in 16,8,8:64 20,10,10:64 26,13,13:64 12,12,12:64 16,8,8:24 26,13,13:8 16,8,8:96 7,3,3:256; do
  rk=${c%%:*}; n=${c##*:}
  python tools/dev_lib_rank.py liblrf_hip.so $rk $n >> $OUT/t.txt 2>&1
done
cat $OUT/t.txt
