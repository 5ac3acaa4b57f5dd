set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_a
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
K=2 LRF_PERSIST=1 python tools/dev_persist_mix_dbg.py 23,8,8 24,24,24 22,22,22 > $OUT/dbg.txt 2>&1 || true
grep "image 0" $OUT/dbg.txt | cut -c1-150
python -m pytest tests/test_configs_at_size.py -x -q -m gpu -k "persistent" > $OUT/t_persist.log 2>&1 || { tail -30 $OUT/t_persist.log; exit 1; }
python -m pytest tests/test_persist_error.py -x -q -m gpu > $OUT/t_err.log 2>&1 || { tail -30 $OUT/t_err.log; exit 1; }
LRF_SWEEP_BATCH=256 python3 tools/dev_rank_sweep.py > $OUT/rank_sweep256.txt 2>$OUT/rank_sweep256.err
cat $OUT/rank_sweep256.txt
python -m pytest tests -x -q -m gpu > $OUT/t_all.log 2>&1 || { tail -30 $OUT/t_all.log; exit 1; }
tail -3 $OUT/t_all.log
