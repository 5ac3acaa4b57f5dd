set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_base
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.txt 2>&1
LRF_SWEEP_BATCH=256 python3 tools/dev_rank_sweep.py > $OUT/rank_sweep256.txt 2>$OUT/rank_sweep256.err
cd /tmp
export LRF_SWEEP_BATCH=256
for t in 16,8,8 26,13,13 10,5,5; do
  n=$(echo $t | tr , _)
  rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/tr_$n -o run -- python3 $GRAFT_REPO_ROOT/tools/dev_rank_sweep.py $t > $OUT/tr_$n.txt 2> $OUT/tr_$n.err
done
find $OUT -name "*.csv" | head -40
