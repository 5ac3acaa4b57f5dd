set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_c
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
{
for r in 16,8,8 10,5,5 26,13,13 20,10,10 7,3,3; do python tools/dev_lib_rank.py liblrf_hip.so $r; done
for r in 16,8,8 10,5,5; do python tools/dev_lib_rank.py liblrf_hip_w16m2.so $r; done
for r in 26,13,13 20,10,10; do python tools/dev_lib_rank.py liblrf_hip_w32f2.so $r; done
for r in 16,8,8 26,13,13; do LRF_NO_INIT_FORK=1 python tools/dev_lib_rank.py liblrf_hip_dev8.so $r; done
for r in 16,8,8 26,13,13; do python tools/dev_lib_rank.py liblrf_hip_dev8.so $r; done
} > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
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
