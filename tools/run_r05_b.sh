set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_b
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_parity.py tests/test_configs_at_size.py tests/test_qmf_class.py -x -q -m gpu > $OUT/t_sel.log 2>&1 || { tail -30 $OUT/t_sel.log; exit 1; }
tail -2 $OUT/t_sel.log
LRF_SWEEP_BATCH=256 python3 tools/dev_rank_sweep.py > $OUT/rank_sweep256.txt 2>$OUT/rank_sweep256.err
cat $OUT/rank_sweep256.txt
cd /tmp
export TMPDIR=/tmp
export LRF_SWEEP_BATCH=256
for t in 16,8,8 26,13,13; do
  n=$(echo $t | tr , _)
  rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/tr_$n -o run -- python3 $GRAFT_REPO_ROOT/tools/dev_rank_sweep.py $t > $OUT/tr_$n.txt 2> $OUT/tr_$n.err
done
cd $GRAFT_REPO_ROOT
python tools/dev_trace_timeline.py $OUT/tr_16_8_8/run_kernel_trace.csv 4
python tools/dev_trace_timeline.py $OUT/tr_26_13_13/run_kernel_trace.csv 4
